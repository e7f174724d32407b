"""Round 6: C3's two 8 GB input columns over many placements in ONE process — the filter's time on each pair of buffers and
the two-stream read skeleton's rate on the same pair (gdv_device_stream_ceiling_on).  Twenty single 8 GB buffers; pairs of
neighbours (2k, 2k+1) and pairs ten allocations apart (k, k+10).   python tools/placement_c3_inputs.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GDV_NO_TIER0", "1")
import torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

rows = 1_000_000_000
flt = gandiva.make_filter(W.c3_schema(), W.c3_condition())
out = torch.empty(rows, dtype=torch.int32, device="cuda")
src = W.c3_device_batch(rows)
bufs = [torch.empty(rows * 8, dtype=torch.uint8, device="cuda") for _ in range(20)]


def ms(db, reps=10):
    for _ in range(3):
        flt.evaluate_device(db, "int32", out=out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        flt.evaluate_device(db, "int32", out=out)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def pair(i, j):
    cols = []
    for c, k in zip(src.columns, (i, j)):
        bufs[k].copy_(c.data)
        cols.append(gandiva.DeviceColumn(c.type, rows, None, bufs[k]))
    db = gandiva.DeviceBatch(src.schema, cols, rows)
    t = ms(db)
    ceil = bench.stream_ceiling_on([bufs[i], bufs[j]], [], rows)
    return t, ceil


print(f"where torch put the generator's columns: {ms(src):.3f} ms; skeleton {bench.stream_ceiling_on([c.data for c in src.columns], [], rows)}", flush=True)
for name, pairs in (("neighbours", [(2 * k, 2 * k + 1) for k in range(10)]), ("ten apart", [(k, k + 10) for k in range(10)])):
    for i, j in pairs:
        t, ceil = pair(i, j)
        print(f"{name} ({i:2d},{j:2d}) @ {bufs[i].data_ptr():#x} {bufs[j].data_ptr():#x}: filter {t:.3f} ms; read skeleton {ceil['GB/s']} GB/s "
              f"= {16e9 / ceil['GB/s'] / 1e6:.3f} ms (grid x{ceil['workgroups_per_cu']}, {ceil['subtiles_per_wave']} sub-tiles, nt {ceil['nontemporal']})", flush=True)
print(f"where torch put the generator's columns, again: {ms(src):.3f} ms")
