"""C5 (like '%spark%', substr(s,2,5), upper(s), 10^8 utf8 rows) over columns that are NOT pure ASCII, by the share of rows that
hold a two-byte character: steady device ms per Evaluate on the EXACT pair of wave kernels, with the pre-pass sweeping all of a
wave tile's sub-tile spans ahead of its row loop (round 5, GDV_PREPASS_AHEAD=1) and one span per iteration (round 4, =0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
offsets, base, _ = W.c5_numpy(n)
lens = np.diff(offsets.astype(np.int64))
pad = lambda t: torch.cat([t, torch.zeros((-t.numel()) % 64 + 64, dtype=torch.uint8)])  # noqa: E731
off_t = pad(torch.from_numpy(offsets.view(np.uint8).copy())).cuda()


def batch(f):
    data = base.copy()
    rng2 = np.random.Generator(np.random.PCG64(1021))
    rows = np.flatnonzero(rng2.random(n) < f) if f >= 1e-6 else np.array([n // 2])
    at = offsets[rows].astype(np.int64) + (rng2.random(len(rows)) * (lens[rows] - 1)).astype(np.int64)
    data[at] = 0xC3
    data[at + 1] = 0xA9
    col = gandiva.DeviceColumn(pa.string(), n, None, pad(torch.from_numpy(data)).cuda(), off_t)
    return gandiva.DeviceBatch(W.c5_schema(), [col], n)


def timed(proj, db, outs, reps=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        outs = proj.evaluate_device(db, outputs=outs)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, outs


fractions = [0.0, 1e-8, 0.001, 0.01, 0.30]
batches = {f: batch(f) if f > 0 else W.c5_device_batch(n) for f in fractions}
for split in ("1", "0"):
    os.environ["GDV_PREPASS_AHEAD"] = split
    proj = gandiva.make_projector(W.c5_schema(), W.c5_expressions(), None)
    outs = proj.evaluate_device(batches[0.0])
    for f in fractions:
        _, outs = timed(proj, batches[f], outs, reps=2)      # the switch (and the re-run) happens here
        ms, outs = timed(proj, batches[f], outs)
        print(f"GDV_PREPASS_AHEAD={split}  {100 * f:9.6f} % non-ASCII rows: {ms:6.3f} ms per Evaluate (path {proj.path_hint})", flush=True)
