"""Round 6, verdict item 2: the slow placements are a property of the SET (tools/placement_map.py: ten buffers that are each
among the slowest alone make the fastest set).  Are they the regular spacing of the buffers?  ONE block per trial, C2's ten output
columns carved out of it at base + k * (2 GiB + delta): the set's write sweep and the projection kernel's time per delta,
over several fresh blocks (all kept alive).      python tools/placement_stagger.py [blocks]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gandiva_amd as gandiva  # noqa: E402
from gandiva_amd import _capi, workloads as W  # noqa: E402

lib = _capi.lib()
rows = 1 << 28
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
db = W.c2_device_batch(rows)
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
outs = proj.evaluate_device(db)
valid = [o.validity for o in outs]
del outs
torch.cuda.empty_cache()
KiB, MiB, GiB = 1 << 10, 1 << 20, 1 << 30
deltas = [0, 256, 4 * KiB, 64 * KiB, 256 * KiB, MiB, 2 * MiB, 3 * MiB, 5 * MiB + 4 * KiB, 8 * MiB, 17 * MiB, 33 * MiB + 64 * KiB, 64 * MiB, 127 * MiB + 4 * KiB]
span = 10 * (2 * GiB + max(deltas)) + 2 * MiB


def sweep(ptrs_w):
    ptrs = (C.c_void_p * len(ptrs_w))(*ptrs_w)
    g, wg, nt = C.c_double(), C.c_int(), C.c_int()
    rc = lib.gdv_device_stream_ceiling_on(ptrs, 0, len(ptrs_w), rows, C.byref(g), C.byref(wg), C.byref(nt))
    return g.value if rc == 0 else float("nan")


def kernel_ms(views):
    cols = [gandiva.DeviceColumn(t, rows, valid[e], v) for e, (v, t) in enumerate(zip(views, proj._out_types))]
    for _ in range(3):
        proj.evaluate_device(db, outputs=cols, sync=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        proj.evaluate_device(db, outputs=cols, sync=False)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 4


keep = []
table = np.zeros((blocks, len(deltas), 2))
for blk in range(blocks):
    if torch.cuda.mem_get_info()[0] < span + (4 << 30):
        blocks = blk
        break
    block = torch.empty(span, dtype=torch.uint8, device="cuda")
    keep.append(block)
    for j, d in enumerate(deltas):
        views = [block[k * (2 * GiB + d): k * (2 * GiB + d) + rows * 8] for k in range(10)]
        table[blk, j] = (kernel_ms(views), sweep([v.data_ptr() for v in views]))
    print(f"block {blk} @ {block.data_ptr():#x}: " + "  ".join(f"{d // KiB}K:{table[blk, j, 0]:.2f}ms/{table[blk, j, 1]:.0f}" for j, d in enumerate(deltas)), flush=True)
print("delta between consecutive columns (beyond 2 GiB) -> kernel ms over the blocks: min / median / max")
for j, d in enumerate(deltas):
    ms = table[:blocks, j, 0]
    print(f"  {d:>11d} B: {ms.min():.3f} / {np.median(ms):.3f} / {ms.max():.3f}   (write sweep of the set {table[:blocks, j, 1].min():.0f} .. {table[:blocks, j, 1].max():.0f} GB/s)")
