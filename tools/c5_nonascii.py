"""C5 (like '%spark%', substr(s,2,5), upper(s), 10^8 utf8 rows) when the column is NOT pure ASCII:
device ms per Evaluate for 0 %, 1 % and 30 % of the rows holding a two-byte character, the Projector's
path after each (0 optimistic / 1 exact wave variant / 2 scanner-shaped general kernel), and the return
to the optimistic kernels on an ASCII batch.  Round 3: one such byte sent the Projector to the general
kernel for good (1.86 ms per batch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
exprs = W.c5_expressions()
proj = gandiva.make_projector(W.c5_schema(), exprs, None)


def timed(db, outs, reps=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        outs = proj.evaluate_device(db, outputs=outs)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, outs


batches = {f: W.c5_device_batch(n, non_ascii_fraction=f) for f in (0.0, 0.01, 0.30)}
outs = proj.evaluate_device(batches[0.0])
for f in (0.0, 0.01, 0.30, 0.0):
    first, outs = timed(batches[f], outs, reps=1)     # the first batch after a switch pays the re-run
    hint_after_first = proj.path_hint
    ms, outs = timed(batches[f], outs)
    print(f"{100 * f:5.1f} % non-ASCII rows: first batch {first:6.3f} ms (path after it: {hint_after_first}), "
          f"steady {ms:6.3f} ms per Evaluate (path {proj.path_hint})", flush=True)
