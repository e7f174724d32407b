"""Round 6: C3's two 8 GB input columns — where torch put them vs copied into a device-pool set (spread + probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gandiva_amd as gandiva
from gandiva_amd import workloads as W

rows = 1_000_000_000
# mimic the sub-line position: something else lived in HBM before
junk = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(40)]
del junk
torch.cuda.empty_cache()
flt = gandiva.make_filter(W.c3_schema(), W.c3_condition())
out = torch.empty(rows, dtype=torch.int32, device="cuda")


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


for trial in range(3):
    db = W.c3_device_batch(rows)
    t_plain = ms(db)
    pool = gandiva.DevicePool()
    ptrs, probe = pool.reserve_set(2, rows * 8, 4)
    cols = []
    for c, p in zip(db.columns, ptrs):
        t = pool._tensor(p, rows * 8)
        t.copy_(c.data)
        cols.append(gandiva.DeviceColumn(c.type, rows, None, t))
    db2 = gandiva.DeviceBatch(db.schema, cols, rows)
    del db
    torch.cuda.empty_cache()
    t_pool = ms(db2)
    print(f"trial {trial}: inputs where torch put them {t_plain:.3f} ms; in a pool set {t_pool:.3f} ms (probe {probe['rates_gbs']} kept {probe['kept']})", flush=True)
    del db2, cols
    pool.close()
    torch.cuda.empty_cache()
