"""The BASELINE.json configurations as concrete expressions + synthetic inputs.

Frozen here so bench.py, the parity tests and the build check all use the same trees and
the same seeded data (SURVEY.md §8d / BASELINE.md §4):

  C1  (a + b) * c            int32, no nulls                         n = 2^20
  C2  10 float64 expressions over a,b,c,d with 10 % nulls            n = 2^28   (headline)
  C3  filter a > 499 AND b < 250 over int64 U[0,1000)                n = 10^9
"""
import numpy as np
import pyarrow as pa

from . import gandiva as gdv

# ------------------------------------------------------------------------------- C1


def c1_schema():
    return pa.schema([pa.field(n, pa.int32()) for n in "abc"])


def c1_expressions(builder=None):
    b = builder or gdv.TreeExprBuilder()
    s = c1_schema()
    a, bb, c = (b.make_field(s.field(i)) for i in range(3))
    add = b.make_function("add", [a, bb], pa.int32())
    mul = b.make_function("multiply", [add, c], pa.int32())
    return [b.make_expression(mul, pa.field("r", pa.int32()))]


def c1_batch(n=1 << 20, seed=1):
    rng = np.random.Generator(np.random.PCG64(seed))
    cols = [pa.array(rng.integers(-(1 << 15), 1 << 15, n, dtype=np.int32)) for _ in range(3)]
    return pa.RecordBatch.from_arrays(cols, schema=c1_schema())


# ------------------------------------------------------------------------------- C2

C2_BYTES_PER_ROW = 4 * 8 + 4 / 8 + 10 * 8 + 10 / 8  # 113.75: each input read once, each output written once


def c2_schema():
    return pa.schema([pa.field(n, pa.float64()) for n in "abcd"])


def c2_expressions(builder=None):
    """e0=a+b, e1=a-b, e2=a*b, e3=c+d, e4=c*d, e5=(a+b)*c, e6=(a-b)*d, e7=a*b+c*d,
    e8=(a+b)*(c-d), e9=((a*b)*c)*d"""
    bld = builder or gdv.TreeExprBuilder()
    s = c2_schema()
    f64 = pa.float64()
    a, b, c, d = (bld.make_field(s.field(i)) for i in range(4))

    def fn(name, x, y):
        return bld.make_function(name, [x, y], f64)
    add, sub, mul = (lambda x, y: fn("add", x, y)), (lambda x, y: fn("subtract", x, y)), (lambda x, y: fn("multiply", x, y))
    roots = [
        add(a, b), sub(a, b), mul(a, b), add(c, d), mul(c, d),
        mul(add(a, b), c), mul(sub(a, b), d), add(mul(a, b), mul(c, d)),
        mul(add(a, b), sub(c, d)), mul(mul(mul(a, b), c), d),
    ]
    return [bld.make_expression(r, pa.field(f"e{i}", f64)) for i, r in enumerate(roots)]


def c2_columns_numpy(n, seed_offset=0):
    """values ~ N(0,1) (PCG64 seeds 42..45), validity ~ Bernoulli(0.9) (seeds 142..145)."""
    vals, masks = [], []
    for k in range(4):
        vals.append(np.random.Generator(np.random.PCG64(42 + k + seed_offset)).standard_normal(n))
        masks.append(np.random.Generator(np.random.PCG64(142 + k + seed_offset)).random(n) >= 0.10)
    return vals, masks


def c2_batch(n, seed_offset=0):
    vals, masks = c2_columns_numpy(n, seed_offset)
    cols = [pa.array(v, mask=~m) for v, m in zip(vals, masks)]
    return pa.RecordBatch.from_arrays(cols, schema=c2_schema())


def c2_device_batch(n, device="cuda", chunk=1 << 24):
    """C2 inputs generated directly in HBM (torch Philox RNG: same distributions as
    c2_batch, different stream — parity at full size is checked through properties, and
    bit-exactly against the oracle on c2_batch-sized prefixes copied back to the host)."""
    import torch
    g = torch.Generator(device=device)
    cols = []
    nbytes_valid = (n + 63) // 64 * 8
    for k in range(4):
        g.manual_seed(42 + k)
        data = torch.empty(n, dtype=torch.float64, device=device)
        valid = torch.zeros(nbytes_valid, dtype=torch.uint8, device=device)
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            data[lo:hi].normal_(generator=g)
            keep = torch.rand(hi - lo, generator=g, device=device) >= 0.10
            # pack LSB-first: bit i of byte j is row 8*j + i
            pad = (-(hi - lo)) % 8
            if pad:
                keep = torch.cat([keep, torch.zeros(pad, dtype=torch.bool, device=device)])
            w = (keep.view(-1, 8).to(torch.uint8) << torch.arange(8, device=device, dtype=torch.uint8)).sum(1, dtype=torch.uint8)
            valid[lo // 8: lo // 8 + w.numel()] = w
        cols.append(gdv.DeviceColumn(pa.float64(), n, valid, data.view(torch.uint8)))
    return gdv.DeviceBatch(c2_schema(), cols, n)


# ------------------------------------------------------------------------------- C3

C3_K1, C3_K2 = 499, 250
C3_BYTES_PER_ROW = 16 + 4 * 0.125  # 16.5: two int64 reads + 4-byte index for the ~12.5 % selected


def c3_schema():
    return pa.schema([pa.field("a", pa.int64()), pa.field("b", pa.int64())])


def c3_condition(builder=None):
    b = builder or gdv.TreeExprBuilder()
    s = c3_schema()
    a, bb = b.make_field(s.field(0)), b.make_field(s.field(1))
    k1, k2 = b.make_literal(C3_K1, pa.int64()), b.make_literal(C3_K2, pa.int64())
    gt = b.make_function("greater_than", [a, k1], pa.bool_())
    lt = b.make_function("less_than", [bb, k2], pa.bool_())
    return b.make_condition(b.make_and([gt, lt]))


def c3_batch(n, null_fraction=0.0):
    a = np.random.Generator(np.random.PCG64(7)).integers(0, 1000, n, dtype=np.int64)
    b = np.random.Generator(np.random.PCG64(8)).integers(0, 1000, n, dtype=np.int64)
    if null_fraction > 0:
        ma = np.random.Generator(np.random.PCG64(107)).random(n) < null_fraction
        mb = np.random.Generator(np.random.PCG64(108)).random(n) < null_fraction
        cols = [pa.array(a, mask=ma), pa.array(b, mask=mb)]
    else:
        cols = [pa.array(a), pa.array(b)]
    return pa.RecordBatch.from_arrays(cols, schema=c3_schema())


def c3_device_batch(n, device="cuda", chunk=1 << 26):
    import torch
    g = torch.Generator(device=device)
    cols = []
    for k, seed in enumerate((7, 8)):
        g.manual_seed(seed)
        data = torch.empty(n, dtype=torch.int64, device=device)
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            data[lo:hi].random_(0, 1000, generator=g)
        cols.append(gdv.DeviceColumn(pa.int64(), n, None, data.view(torch.uint8)))
    return gdv.DeviceBatch(c3_schema(), cols, n)
