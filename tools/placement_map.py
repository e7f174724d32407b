"""Round 6, verdict item 2 ("name the mechanism"): is a slow placement a property of the individual buffers (WHERE in HBM each
lies) or of the set (how the buffers lie relative to each other)?  Allocates N buffers of 2 GiB (most of the HBM), measures
every buffer ALONE (write-only and read-only streaming sweep), then runs C2's projection kernel with its ten outputs on
(a) the ten individually fastest buffers, (b) the ten slowest, (c) ten random ones, and sweeps 256 MiB pieces of one fast and one
slow buffer.      python tools/placement_map.py [buffers]"""
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
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 96
db = W.c2_device_batch(rows)
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
outs = proj.evaluate_device(db)
valid = [o.validity for o in outs]
del outs
torch.cuda.empty_cache()


def sweep(reads, writes, elems):
    ptrs = (C.c_void_p * (len(reads) + len(writes)))(*(reads + writes))
    g, wg, nt = C.c_double(), C.c_int(), C.c_int()
    rc = lib.gdv_device_stream_ceiling_on(ptrs, len(reads), len(writes), elems, C.byref(g), C.byref(wg), C.byref(nt))
    return g.value if rc == 0 else float("nan")


bufs = []
while len(bufs) < nbuf and torch.cuda.mem_get_info()[0] > (6 << 30):
    bufs.append(torch.empty(rows * 8, dtype=torch.uint8, device="cuda"))
print(f"{len(bufs)} buffers of 2 GiB allocated ({len(bufs) * 2} GiB); free now {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB")
wr = np.array([sweep([], [b.data_ptr()], rows) for b in bufs])
rd = np.array([sweep([b.data_ptr()], [], rows) for b in bufs])
for i, b in enumerate(bufs):
    print(f"buffer {i:3d} @ {b.data_ptr():#x}: alone, write {wr[i]:7.1f} GB/s, read {rd[i]:7.1f} GB/s")
print(f"single-buffer write: min {wr.min():.0f} median {np.median(wr):.0f} max {wr.max():.0f} GB/s; read: min {rd.min():.0f} median {np.median(rd):.0f} max {rd.max():.0f}")


def kernel_ms(idx):
    cols = [gandiva.DeviceColumn(pa_t, rows, valid[e], bufs[i]) for e, (i, pa_t) in enumerate(zip(idx, proj._out_types))]
    for _ in range(3):
        proj.evaluate_device(db, outputs=cols, sync=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        proj.evaluate_device(db, outputs=cols, sync=False)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 4, sweep([], [bufs[i].data_ptr() for i in idx], rows)


order = np.argsort(-wr)
rng = np.random.default_rng(0)
for label, idx in (("ten individually FASTEST buffers", order[:10]), ("ten individually SLOWEST buffers", order[-10:]),
                   ("ten random buffers", rng.choice(len(bufs), 10, replace=False)), ("ten consecutive (allocation order 0-9)", np.arange(10)),
                   ("ten consecutive (allocation order last 10)", np.arange(len(bufs) - 10, len(bufs))),
                   ("5 fastest + 5 slowest", np.concatenate([order[:5], order[-5:]]))):
    ms, set_w = kernel_ms([int(i) for i in idx])
    print(f"C2 outputs on the {label}: kernel {ms:.3f} ms; write sweep of the set {set_w:.0f} GB/s; members' own write rates {np.round(wr[idx]).astype(int).tolist()}")
piece = (256 << 20) // 8
for label, i in (("fastest", int(order[0])), ("slowest", int(order[-1]))):
    rates = [sweep([], [bufs[i].data_ptr() + k * piece * 8], piece) for k in range(8)]
    print(f"256 MiB pieces of the {label} buffer ({i}): write {np.round(rates).astype(int).tolist()} GB/s")
