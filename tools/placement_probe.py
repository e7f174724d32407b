"""Round 6, verdict item 2: does a cheap streaming probe of a SET of freshly allocated output buffers predict what the C2 kernel
will run at on that placement?  For each of N placements of C2's ten 2 GiB output columns (inputs fixed): the projection kernel's
ms per step, the write-only skeleton sweep over the ten buffers, and the 4-read + 10-write skeleton.  Prints the table and the
rank agreement.      python tools/placement_probe.py [placements] [rows]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gandiva_amd as gandiva  # noqa: E402
from gandiva_amd import _capi, workloads as W  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 28
lib = _capi.lib()
db = W.c2_device_batch(rows)
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), None)


def kernel_ms(outs):
    for _ in range(3):
        proj.evaluate_device(db, outputs=outs, sync=False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(4):
        proj.evaluate_device(db, outputs=outs, sync=False)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 4


def sweep(reads, writes):
    ptrs = (C.c_void_p * (len(reads) + len(writes)))(*[t.data_ptr() for t in reads + writes])
    g, wg, nt = C.c_double(), C.c_int(), C.c_int()
    rc = lib.gdv_device_stream_ceiling_on(ptrs, len(reads), len(writes), rows, C.byref(g), C.byref(wg), C.byref(nt))
    return g.value if rc == 0 else float("nan")


keep = []          # every placement stays allocated: the driver cannot hand the same pages back
table = []
outs = proj.evaluate_device(db)
for t in range(trials):
    if t:
        outs = proj.evaluate_device(db)
    keep.append(outs)
    ms = kernel_ms(outs)
    w = sweep([], [o.data for o in outs])
    rw = sweep([c.data for c in db.columns], [o.data for o in outs])
    ms2 = kernel_ms(outs)
    table.append((ms, ms2, w, rw, [hex(o.data.data_ptr()) for o in outs[:2]]))
    print(f"placement {t}: kernel {ms:.3f} ms (again {ms2:.3f}); write-only sweep of the 10 buffers {w:7.1f} GB/s; 4 reads + 10 writes sweep {rw:7.1f} GB/s; "
          f"first outputs at {table[-1][4]}", flush=True)
    if torch.cuda.mem_get_info()[0] < 26 << 30:
        break
import numpy as np  # noqa: E402
ms = np.array([r[0] for r in table]); w = np.array([r[2] for r in table]); rw = np.array([r[3] for r in table])
rank = lambda x: np.argsort(np.argsort(x))  # noqa: E731
print(f"spearman(kernel ms, 1 / write-only GB/s) = {np.corrcoef(rank(ms), rank(1 / w))[0, 1]:.3f}; "
      f"spearman(kernel ms, 1 / read+write GB/s) = {np.corrcoef(rank(ms), rank(1 / rw))[0, 1]:.3f}")
print(f"kernel ms: min {ms.min():.3f} median {np.median(ms):.3f} max {ms.max():.3f}; the placement the write-only probe would pick: {ms[np.argmax(w)]:.3f} ms; "
      f"the one the read+write probe would pick: {ms[np.argmax(rw)]:.3f} ms")
