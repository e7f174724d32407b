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


def c2_device_batch(n, device="cuda", chunk=1 << 24, seed_offset=0):
    """C2 inputs generated directly in HBM (torch Philox RNG: same distributions as
    c2_batch, different stream — parity at full size is checked through properties, and
    bit-exactly against the oracle on c2_batch-sized prefixes copied back to the host)."""
    import torch
    g = torch.Generator(device=device)
    cols = []
    nbytes_valid = (n + 63) // 64 * 8
    for k in range(4):
        g.manual_seed(42 + k + seed_offset)
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


def c2_device_batch_pcg64(n, device="cuda", chunk=1 << 22, seed_offset=0):
    """C2 inputs in HBM from BASELINE.md §4's frozen streams: the SAME rows c2_batch(n) holds
    (numpy PCG64, value seeds 42..45, mask seeds 142..145 — consecutive calls on one Generator
    continue its stream, so chunked generation equals one call), produced chunk by chunk on
    the host — one thread per stream, numpy releases the GIL inside the generators — and
    uploaded as they come.  ~10-20 s for 2^28 rows; nothing of it is inside a timed region."""
    import threading
    import torch
    nbytes_valid = (n + 63) // 64 * 8
    datas = [torch.empty(n, dtype=torch.float64, device=device) for _ in range(4)]
    valids = [torch.zeros(nbytes_valid, dtype=torch.uint8, device=device) for _ in range(4)]
    errors = []

    def values(k):
        try:
            rng = np.random.Generator(np.random.PCG64(42 + k + seed_offset))
            for lo in range(0, n, chunk):
                m = min(chunk, n - lo)
                datas[k][lo:lo + m].copy_(torch.from_numpy(rng.standard_normal(m)))
        except Exception as e:  # surfaced by the caller
            errors.append(e)

    def masks(k):
        try:
            rng = np.random.Generator(np.random.PCG64(142 + k + seed_offset))
            for lo in range(0, n, chunk):          # chunk is a multiple of 8: whole bitmap bytes
                m = min(chunk, n - lo)
                w = np.packbits(rng.random(m) >= 0.10, bitorder="little")
                valids[k][lo // 8: lo // 8 + w.size].copy_(torch.from_numpy(w))
        except Exception as e:
            errors.append(e)
    threads = [threading.Thread(target=f, args=(k,)) for k in range(4) for f in (values, masks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    if str(device).startswith("cuda"):
        torch.cuda.synchronize()
    cols = [gdv.DeviceColumn(pa.float64(), n, valids[k], datas[k].view(torch.uint8)) for k in range(4)]
    return gdv.DeviceBatch(c2_schema(), cols, n)


def c2_expected_window(dbatch, lo, m):
    """What C2's ten outputs must hold for rows [lo, lo+m) (lo a multiple of 8), recomputed by an
    independent engine on the device: torch float64 elementwise arithmetic (IEEE add / subtract /
    multiply, one rounding per operator, no fused multiply-add: torch evaluates every operator as
    its own kernel) and the bitwise AND of the input validity bytes.  Returns
    ([10 float64 tensors], [10 uint8 tensors of ceil(m/8) validity bytes])."""
    import torch
    a, b, c, d = (col.data.view(torch.float64)[lo:lo + m] for col in dbatch.columns)
    nb = (m + 7) // 8
    va, vb, vc, vd = (col.validity[lo // 8: lo // 8 + nb] for col in dbatch.columns)
    ab, cd, abcd = va & vb, vc & vd, va & vb & vc & vd
    s, df, p, q = a + b, a - b, a * b, c * d
    vals = [s, df, p, c + d, q, s * c, df * d, p + q, s * (c - d), (p * c) * d]
    valid = [ab, ab, ab, cd, cd, ab & vc, ab & vd, abcd, abcd, abcd]
    return vals, valid


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


def c3_sum_expression(builder=None):
    """[a + b]: what the filter -> project chain of tools/filter_project_chain.py (and the fused
    filter-project kernel) projects for the rows C3's condition selects."""
    b = builder or gdv.TreeExprBuilder()
    s = c3_schema()
    a, bb = b.make_field(s.field(0)), b.make_field(s.field(1))
    return [b.make_expression(b.make_function("add", [a, bb], pa.int64()), pa.field("s", pa.int64()))]


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


# ------------------------------------------------------------------------------- C4
# TPC-H lineitem Q1 projections: ep*(1-disc), ep*(1-disc)*(1+tax) in decimal128, and the
# day difference 1998-12-01 - l_shipdate (date32).

C4_BYTES_PER_ROW = 3 * 16 + 4 + 4 / 8 + 2 * 16 + 4 + 3 / 8  # 88.875: 52.5 read + 36.375 written
C4_DATE_1998_12_01 = 10561  # days since 1970-01-01


def c4_schema():
    d = pa.decimal128(15, 2)
    return pa.schema([pa.field("l_extendedprice", d), pa.field("l_discount", d),
                      pa.field("l_tax", d), pa.field("l_shipdate", pa.date32())])


def c4_expressions(builder=None):
    b = builder or gdv.TreeExprBuilder()
    s = c4_schema()
    ep, disc, tax, ship = (b.make_field(s.field(i)) for i in range(4))
    d152 = pa.decimal128(15, 2)
    one = b.make_literal(100, d152)                                   # 1.00
    one_minus = b.make_function("subtract", [one, disc], pa.decimal128(16, 2))
    disc_price = b.make_function("multiply", [ep, one_minus], pa.decimal128(32, 4))
    one_plus = b.make_function("add", [one, tax], pa.decimal128(16, 2))
    charge = b.make_function("multiply", [disc_price, one_plus], pa.decimal128(38, 6))
    cutoff = b.make_literal(C4_DATE_1998_12_01, pa.date32())
    days = b.make_function("datediff", [cutoff, ship], pa.int32())
    return [b.make_expression(disc_price, pa.field("disc_price", pa.decimal128(32, 4))),
            b.make_expression(charge, pa.field("charge", pa.decimal128(38, 6))),
            b.make_expression(days, pa.field("days", pa.int32()))]


def _decimal_array(unscaled, t, mask=None):
    """int64 numpy unscaled values -> decimal128 pyarrow array (sign-extended 16-byte slots)."""
    n = len(unscaled)
    raw = np.empty((n, 2), dtype=np.int64)
    raw[:, 0] = unscaled
    raw[:, 1] = unscaled >> 63
    validity = None
    if mask is not None:
        validity = pa.py_buffer(np.packbits(~mask, bitorder="little"))
    return pa.Array.from_buffers(t, n, [validity, pa.py_buffer(raw)])


def c4_batch(n, null_fraction=0.0):
    d = pa.decimal128(15, 2)
    ep = np.random.Generator(np.random.PCG64(11)).integers(90000, 10500000, n, dtype=np.int64)
    disc = np.random.Generator(np.random.PCG64(12)).integers(0, 11, n, dtype=np.int64)
    tax = np.random.Generator(np.random.PCG64(13)).integers(0, 9, n, dtype=np.int64)
    ship = np.random.Generator(np.random.PCG64(14)).integers(8036, 10562, n, dtype=np.int32)
    masks = [None] * 4
    if null_fraction > 0:
        masks = [np.random.Generator(np.random.PCG64(111 + k)).random(n) < null_fraction for k in range(4)]
    cols = [_decimal_array(ep, d, masks[0]), _decimal_array(disc, d, masks[1]),
            _decimal_array(tax, d, masks[2]),
            pa.array(ship, type=pa.int32(), mask=masks[3]).cast(pa.date32())]
    return pa.RecordBatch.from_arrays(cols, schema=c4_schema())


def c4_device_batch(n, device="cuda", chunk=1 << 25):
    import torch
    g = torch.Generator(device=device)
    cols = []
    for seed, lo, hi in ((11, 90000, 10500000), (12, 0, 11), (13, 0, 9)):
        g.manual_seed(seed)
        data = torch.zeros(n, 2, dtype=torch.int64, device=device)  # high words stay 0 (values >= 0)
        for a in range(0, n, chunk):
            z = min(n, a + chunk)
            data[a:z, 0].random_(lo, hi, generator=g)
        cols.append(gdv.DeviceColumn(pa.decimal128(15, 2), n, None, data.view(torch.uint8).reshape(-1)))
    g.manual_seed(14)
    ship = torch.empty(n, dtype=torch.int32, device=device)
    ship.random_(8036, 10562, generator=g)
    cols.append(gdv.DeviceColumn(pa.date32(), n, None, ship.view(torch.uint8)))
    return gdv.DeviceBatch(c4_schema(), cols, n)


# ------------------------------------------------------------------------------- C5
# utf8 like / substr / upper over a var-len column: lengths U[4,20], ASCII letters, 5 % of the
# rows contain "spark".

def c5_schema():
    return pa.schema([pa.field("s", pa.string())])


def c5_expressions(builder=None):
    b = builder or gdv.TreeExprBuilder()
    s = b.make_field(c5_schema().field(0))
    like = b.make_function("like", [s, b.make_literal("%spark%", pa.string())], pa.bool_())
    sub = b.make_function("substr", [s, b.make_literal(2, pa.int64()), b.make_literal(5, pa.int64())],
                          pa.string())
    up = b.make_function("upper", [s], pa.string())
    return [b.make_expression(like, pa.field("is_spark", pa.bool_())),
            b.make_expression(sub, pa.field("sub", pa.string())),
            b.make_expression(up, pa.field("up", pa.string()))]


def c5_numpy(n, seed=21, null_fraction=0.0, non_ascii_fraction=0.0):
    """(offsets int32[n+1], bytes uint8[total], null mask or None).  non_ascii_fraction > 0: that share
    of the rows gets one two-byte character (e-acute, C3 A9) over two of its bytes — the variant of C5
    that real utf8 columns look like (round 4: profiles/r04_c5_nonascii.txt)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = rng.integers(4, 21, n).astype(np.int64)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lens, out=offsets[1:])
    letters = np.frombuffer(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ", dtype=np.uint8)
    data = letters[rng.integers(0, len(letters), int(offsets[-1]))].copy()
    # plant "spark" in ~5 % of the rows that are long enough
    pick = np.flatnonzero((rng.random(n) < 0.05) & (lens >= 5))
    pos = offsets[pick] + (rng.random(len(pick)) * (lens[pick] - 4)).astype(np.int64)
    for k, ch in enumerate(b"spark"):
        data[pos + k] = ch
    mask = (rng.random(n) < null_fraction) if null_fraction > 0 else None
    if non_ascii_fraction > 0:
        rng2 = np.random.Generator(np.random.PCG64(seed + 1000))
        rows = np.flatnonzero(rng2.random(n) < non_ascii_fraction)
        at = offsets[rows] + (rng2.random(len(rows)) * (lens[rows] - 1)).astype(np.int64)   # lens >= 4: room for two bytes
        data[at] = 0xC3
        data[at + 1] = 0xA9
    return offsets.astype(np.int32), data, mask


def c5_batch(n, null_fraction=0.0, non_ascii_fraction=0.0):
    offsets, data, mask = c5_numpy(n, null_fraction=null_fraction, non_ascii_fraction=non_ascii_fraction)
    validity = None if mask is None else pa.py_buffer(np.packbits(~mask, bitorder="little"))
    arr = pa.Array.from_buffers(pa.string(), n, [validity, pa.py_buffer(offsets), pa.py_buffer(data)])
    return pa.RecordBatch.from_arrays([arr], schema=c5_schema())


def c5_device_batch(n, device="cuda", non_ascii_fraction=0.0):
    import torch
    offsets, data, _ = c5_numpy(n, non_ascii_fraction=non_ascii_fraction)
    pad = lambda t: torch.cat([t, torch.zeros((-t.numel()) % 64 + 64, dtype=torch.uint8)])
    off_t = pad(torch.from_numpy(offsets.view(np.uint8).copy())).to(device)
    dat_t = pad(torch.from_numpy(data)).to(device)
    col = gdv.DeviceColumn(pa.string(), n, None, dat_t, off_t)
    return gdv.DeviceBatch(c5_schema(), [col], n)


def c5_device_batch_philox(n, device="cuda", seed=21, chunk=1 << 27):
    """C5's column generated in HBM (torch Philox): BASELINE.md §4's distributions — lengths U[4,20], ASCII
    letters, "spark" planted in ~5 % of the rows — not its PCG64 stream (c5_numpy draws 1.2 * 10^9 letters on one
    host core).  Seconds for 10^8 rows; parity against the oracle runs on c5_batch-sized inputs."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lens = torch.randint(4, 21, (n,), generator=g, device=device, dtype=torch.int64)
    offsets = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=offsets[1:])
    total = int(offsets[-1])
    assert total < (1 << 31), "int32 offsets"
    letters = torch.tensor(list(b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"), dtype=torch.uint8, device=device)
    data = torch.zeros(total + (-total) % 64 + 64, dtype=torch.uint8, device=device)
    for lo in range(0, total, chunk):
        m = min(chunk, total - lo)
        data[lo:lo + m] = letters[torch.randint(0, 52, (m,), generator=g, device=device)]
    pick = torch.nonzero((torch.rand(n, generator=g, device=device) < 0.05) & (lens >= 5)).view(-1)
    pos = offsets[pick] + (torch.rand(pick.numel(), generator=g, device=device) * (lens[pick] - 4).to(torch.float32)).to(torch.int64)
    pos = torch.minimum(pos, offsets[pick] + lens[pick] - 5)
    for k, ch in enumerate(b"spark"):
        data[pos + k] = ch
    off32 = offsets.to(torch.int32)
    off_t = torch.zeros((n + 1) * 4 + (-(n + 1) * 4) % 64 + 64, dtype=torch.uint8, device=device)
    off_t[:(n + 1) * 4] = off32.view(torch.uint8)
    col = gdv.DeviceColumn(pa.string(), n, None, data, off_t)
    return gdv.DeviceBatch(c5_schema(), [col], n)
