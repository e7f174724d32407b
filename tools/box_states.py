"""Round 5, verdict item 6: "two box states 9-15 % apart" — is it where an allocation lands?
ONE process, C2 at 2^28 rows (torch-generated inputs: the values do not matter here):
  round r = 0..R-1:  allocate the 4 input columns + 4 bitmaps + 10 outputs + 10 bitmaps (round 0: torch's caching
                     allocator; later rounds: after freeing everything and emptying the cache, so the driver hands
                     out fresh physical pages), run 5 untimed + 12 timed Evaluates (HIP events, per launch), print
                     median / min / max and the address bits of every buffer.
  then the same with every buffer carved from ONE 2 MiB-aligned hipMalloc block (torch.empty of the total size,
  sub-views at 2 MiB-aligned offsets).
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 28
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
carve_rounds = 0 if "--no-carve" in sys.argv else 2
proj = gandiva.make_projector(W.c2_schema(), W.c2_expressions(), pa.default_memory_pool())


def make_inputs_torch():
    return W.c2_device_batch(n)


def time_round(db, outs, label):
    ts = []
    for i in range(17):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        outs = proj.evaluate_device(db, outputs=outs, sync=False)
        e.record()
        torch.cuda.synchronize()
        if i >= 5:
            ts.append(s.elapsed_time(e))
    ts = np.array(ts)
    addrs = [c.data.data_ptr() for c in db.columns] + [o.data.data_ptr() for o in outs]
    low = sorted({(a >> 21) & 0x3ff for a in addrs})
    al = min((a & -a) for a in addrs)
    print(f"{label}: median {np.median(ts):.3f} ms  min {ts.min():.3f}  max {ts.max():.3f}   "
          f"smallest buffer alignment {al >> 10} KiB; first input @ {addrs[0]:#x}, first output @ {addrs[4]:#x}", flush=True)
    return outs


print(f"# C2, {n} rows, one process; torch {torch.__version__}; {torch.cuda.get_device_name(0)}", flush=True)
for r in range(rounds):
    db = make_inputs_torch()
    outs = time_round(db, None, f"round {r} (torch caching allocator{', fresh after empty_cache' if r else ''})")
    # same buffers, timed a second time: does the state belong to the buffers or to the moment?
    time_round(db, outs, f"round {r} again, same buffers")
    del db, outs
    torch.cuda.synchronize()
    torch.cuda.empty_cache()

# one big block, carved by hand: buffer i starts at a 2 MiB boundary + i * stagger bytes.  stagger = 0: every stream
# shares its low 21 address bits with every other (what torch's allocator gives for large tensors); a stagger
# de-phases the 14 + 10 concurrently advancing streams in the bits a channel / bank hash may use.
def carve(total_views, stagger):
    slot = [((_s + stagger * len(total_views) + (1 << 21) - 1) & ~((1 << 21) - 1)) for _s in total_views]
    block = torch.empty(sum(slot) + (1 << 21), dtype=torch.uint8, device="cuda")
    off = (-block.data_ptr()) % (1 << 21)
    views = []
    for i, (sz, want) in enumerate(zip(slot, total_views)):
        a = off + i * stagger
        views.append(block[a:a + want])
        off += sz
    return block, views


staggers = [int(x) for x in os.environ.get("STAGGERS", "0,256,4352,69888").split(",")] if carve_rounds else []
for r in range(carve_rounds):
  for stagger in staggers:
    src = make_inputs_torch()
    need = []
    for c in src.columns:
        need += [c.data.numel(), c.validity.numel()]
    vb = (n + 63) // 64 * 8
    for _ in range(10):
        need += [n * 8, vb]
    block, views = carve(need, stagger)
    cols = []
    for k, c in enumerate(src.columns):
        views[2 * k].copy_(c.data)
        views[2 * k + 1].copy_(c.validity)
        cols.append(gandiva.DeviceColumn(c.type, n, views[2 * k + 1], views[2 * k]))
    del src
    torch.cuda.empty_cache()
    db = gandiva.DeviceBatch(W.c2_schema(), cols, n)
    outs = [gandiva.DeviceColumn(pa.float64(), n, views[8 + 2 * e + 1], views[8 + 2 * e]) for e in range(10)]
    time_round(db, outs, f"carved round {r}, stagger {stagger:6d} B")
    del db, outs, cols, views, block
    torch.cuda.empty_cache()
