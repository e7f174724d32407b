"""Round 6, verdict item 2: subsets of a pool of single 2 GiB buffers as C2's ten output columns — how good is the best of T random
subsets (by the write sweep), and does the sweep pick a good one?      python tools/placement_subsets.py [pool] [subsets]"""
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
npool = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nsub = int(sys.argv[2]) if len(sys.argv) > 2 else 32
db = W.c2_device_batch(rows)
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)
outs = proj.evaluate_device(db)
valid = [o.validity for o in outs]
del outs
torch.cuda.empty_cache()
bufs = [torch.empty(rows * 8, dtype=torch.uint8, device="cuda") for _ in range(npool)]


def sweep(idx):
    ptrs = (C.c_void_p * len(idx))(*[bufs[i].data_ptr() for i in idx])
    g, wg, nt = C.c_double(), C.c_int(), C.c_int()
    lib.gdv_device_stream_ceiling_on(ptrs, 0, len(idx), rows, C.byref(g), C.byref(wg), C.byref(nt))
    return g.value


def kernel_ms(idx):
    cols = [gandiva.DeviceColumn(t, rows, valid[e], bufs[i]) for e, (i, t) in enumerate(zip(idx, proj._out_types))]
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


rng = np.random.default_rng(1)
subsets = [list(range(10))] + [sorted(rng.choice(npool, 10, replace=False).tolist()) for _ in range(nsub - 1)]
# strided picks: members far apart in allocation order
subsets += [list(range(k, npool, npool // 10))[:10] for k in range(min(4, npool // 10))]
res = [(sweep(s), kernel_ms(s), s) for s in subsets]
for k, (g, ms, s) in enumerate(res):
    tag = "first ten allocations" if k == 0 else ("strided" if k >= nsub else "random")
    print(f"{tag:22s} {s}: write sweep {g:7.1f} GB/s, kernel {ms:.3f} ms")
g = np.array([r[0] for r in res]); ms = np.array([r[1] for r in res])
rank = lambda x: np.argsort(np.argsort(x))  # noqa: E731
print(f"{len(res)} subsets of {npool} buffers: kernel ms min {ms.min():.3f} / median {np.median(ms):.3f} / max {ms.max():.3f}; the sweep's pick runs at {ms[np.argmax(g)]:.3f} ms; "
      f"first ten allocations {ms[0]:.3f} ms; spearman {np.corrcoef(rank(ms), rank(-g))[0, 1]:.3f}")
for T in (4, 8, 16, 32):
    picks = [ms[:T][np.argmax(g[:T])]]
    print(f"  best of the first {T} subsets by the sweep: {picks[0]:.3f} ms")
