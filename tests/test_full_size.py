"""BASELINE.json configs C3 / C4 / C5 at their FULL sizes, checked, not just timed.

The oracle cannot hold these batches, so each config is pinned three ways:
  * size-independent properties over every row, recomputed by an independent engine on the
    GPU (torch integer arithmetic / nonzero / cumsum) or vectorised numpy on the host;
  * a 10^5-row prefix and a 10^5-row window from the middle of the batch copied to the host
    and compared bit for bit with the oracle;
  * structural invariants of the Arrow result (ascending indices, closing offset = byte total).
Round 4: C2 — the headline configuration — at its own 2^28 rows, every output and every validity buffer.
"""
import os

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import workloads as W
from helpers import assert_bit_exact
from oracle import oracle

pytestmark = pytest.mark.gpu

FULL = os.environ.get("GDV_FULL_SIZE", "1") != "0"
WIN = 100_000


def _host_fixed(col, lo, m, t):
    """rows [lo, lo+m) of a device column without nulls -> host pyarrow array"""
    w = t.bit_width // 8
    raw = col.data[lo * w:(lo + m) * w].cpu().numpy()
    return pa.Array.from_buffers(t, m, [None, pa.py_buffer(raw)])


def _host_out_fixed(col, lo, m, t):
    """rows [lo, lo+m) of a device OUTPUT column (lo multiple of 8) -> host pyarrow array"""
    assert lo % 8 == 0
    w = t.bit_width // 8
    raw = col.data[lo * w:(lo + m) * w].cpu().numpy()
    valid = col.validity[lo // 8:lo // 8 + (m + 7) // 8].cpu().numpy()
    return pa.Array.from_buffers(t, m, [pa.py_buffer(valid), pa.py_buffer(raw)])


def test_c2_headline_projection_at_2_to_the_28_rows():
    """The configuration BASELINE.json's metric is quoted on, at the size it is quoted at, on the data
    bench.py times (BASELINE.md §4's PCG64 streams): ALL ten outputs bit for bit against torch float64
    arithmetic and ALL ten validity buffers against the bitwise AND of their inputs' bitmaps, over every
    row in slabs; the head of the batch and a window from its middle bit-exact against the oracle."""
    import torch
    n = (1 << 28) if FULL else (1 << 22)
    db = W.c2_device_batch_pcg64(n)
    exprs = W.c2_expressions()
    proj = gandiva.make_projector(W.c2_schema(), exprs, None)
    outs = proj.evaluate_device(db)
    torch.cuda.synchronize()
    slab = 1 << 24
    for lo in range(0, n, slab):
        m = min(slab, n - lo)
        vals, valid = W.c2_expected_window(db, lo, m)
        for e, (o, v, vb) in enumerate(zip(outs, vals, valid)):
            assert torch.equal(o.data.view(torch.int64)[lo:lo + m], v.view(torch.int64)), f"e{e}, rows [{lo}, {lo + m})"
            assert torch.equal(o.validity[lo // 8:(lo + m) // 8], vb[:m // 8]), f"validity of e{e}, rows [{lo}, {lo + m})"
        del vals, valid
    # about one row in ten is NULL in each input: the merged bitmaps are neither all-ones nor all-zeros
    pop = int(torch.count_nonzero(outs[9].validity[:n // 8] == 0xff))
    assert 0.030 * (n // 8) < pop < 0.039 * (n // 8)   # P(all 8 rows of a byte valid in all 4 inputs) = 0.9^32 = 0.0343
    for lo in (0, (n // 2) & ~63):
        m = min(WIN, n - lo)
        cols = []
        for c in db.columns:
            raw = c.data[lo * 8:(lo + m) * 8].cpu().numpy()
            vb = c.validity[lo // 8:lo // 8 + (m + 7) // 8].cpu().numpy()
            cols.append(pa.Array.from_buffers(pa.float64(), m, [pa.py_buffer(vb), pa.py_buffer(raw)]))
        hb = pa.RecordBatch.from_arrays(cols, schema=W.c2_schema())
        if lo == 0:   # the device batch IS the frozen stream: its head equals the host generator's
            assert hb.equals(W.c2_batch(m))
        for o, w, e in zip(outs, oracle.project(exprs, hb), exprs):
            assert_bit_exact(_host_out_fixed(o, lo, m, w.type), w, f"{e} rows [{lo}, {lo + m})")


def test_c3_filter_at_one_billion_rows():
    """10^9 int64 rows -> uint32 SelectionVector: same index list as torch.nonzero over the
    same predicate (exact, every element), ascending, count = population count."""
    import torch
    n = 10**9 if FULL else 10**7
    db = W.c3_device_batch(n)
    flt = gandiva.make_filter(W.c3_schema(), W.c3_condition())
    sel = flt.evaluate_device(db, "int32")
    torch.cuda.synchronize()
    a = db.columns[0].data.view(torch.int64)
    b = db.columns[1].data.view(torch.int64)
    count = 0
    got = sel.indices[:sel.num_slots].view(torch.int32)
    # torch.nonzero in slabs (a 10^9-element nonzero needs > 2^31 intermediate elements)
    slab = 1 << 28
    for lo in range(0, n, slab):
        hi = min(n, lo + slab)
        want = torch.nonzero((a[lo:hi] > W.C3_K1) & (b[lo:hi] < W.C3_K2)).view(-1) + lo
        g = got[count:count + want.numel()].to(torch.int64) & 0xffffffff
        assert torch.equal(g, want), f"indices differ in rows [{lo}, {hi})"
        count += want.numel()
    assert sel.num_slots == count
    # prefix vs the oracle
    m = WIN
    hb = pa.RecordBatch.from_arrays([_host_fixed(c, 0, m, pa.int64()) for c in db.columns], schema=W.c3_schema())
    want = oracle.filter_indices(W.c3_condition(), hb, "int32").to_numpy()
    assert np.array_equal(got[:len(want)].cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_c4_decimal_projection_at_750m_rows():
    """7.5*10^8 lineitem rows: every decimal128 product and every day difference against
    64-bit torch arithmetic (the products fit 64 bits for this data: the high words must be
    zero), plus a prefix and a mid-batch window bit-exact against the oracle."""
    import torch
    n = 750_000_000 if FULL else 5_000_000
    db = W.c4_device_batch(n)
    exprs = W.c4_expressions()
    proj = gandiva.make_projector(W.c4_schema(), exprs, None)
    outs = proj.evaluate_device(db)
    torch.cuda.synchronize()
    ep, disc, tax = (db.columns[k].data.view(torch.int64).view(-1, 2) for k in range(3))
    ship = db.columns[3].data.view(torch.int32)
    dp = outs[0].data.view(torch.int64)[:2 * n].view(-1, 2)
    ch = outs[1].data.view(torch.int64)[:2 * n].view(-1, 2)
    days = outs[2].data.view(torch.int32)[:n]
    slab = 1 << 27
    for lo in range(0, n, slab):
        hi = min(n, lo + slab)
        want_dp = ep[lo:hi, 0] * (100 - disc[lo:hi, 0])
        assert torch.equal(dp[lo:hi, 0], want_dp) and not dp[lo:hi, 1].any()
        assert torch.equal(ch[lo:hi, 0], want_dp * (100 + tax[lo:hi, 0])) and not ch[lo:hi, 1].any()
        assert torch.equal(days[lo:hi], W.C4_DATE_1998_12_01 - ship[lo:hi])
    nb = n // 8
    for o in outs:  # no input nulls -> every validity bit of the whole bytes is set
        assert bool((o.validity[:nb] == 0xff).all())
    types = [pa.decimal128(15, 2)] * 3 + [pa.date32()]
    for lo in (0, (n // 2) & ~63):
        m = min(WIN, n - lo)
        hb = pa.RecordBatch.from_arrays([_host_fixed(c, lo, m, t) for c, t in zip(db.columns, types)],
                                        schema=W.c4_schema())
        want = oracle.project(exprs, hb)
        for o, w, e in zip(outs, want, exprs):
            assert_bit_exact(_host_out_fixed(o, lo, m, w.type), w, f"{e} rows [{lo}, {lo + m})")


def test_c5_strings_at_100m_rows():
    """10^8 utf8 rows (1.2 GB): like '%spark%' against a vectorised numpy search over the
    whole byte buffer, upper() bytes and offsets against torch, substr offsets against a
    torch cumsum of min(5, len - 1), closing offsets = byte totals, and a prefix + a mid-batch
    window bit-exact against the oracle."""
    import torch
    n = 100_000_000 if FULL else 2_000_000
    offsets, data, _ = W.c5_numpy(n)
    db = W.c5_device_batch(n)
    exprs = W.c5_expressions()
    proj = gandiva.make_projector(W.c5_schema(), exprs, None)
    outs = proj.evaluate_device(db)
    torch.cuda.synchronize()
    like, sub, up = outs
    total = int(offsets[-1])
    in_off = db.columns[0].offsets.view(torch.int32)[:n + 1]
    in_dat = db.columns[0].data[:total]

    # upper: same offsets, bytes = ASCII upper of the input, closing offset = byte total
    up_off = up.offsets.view(torch.int32)[:n + 1]
    assert torch.equal(up_off, in_off)
    assert up.data_used == total
    lower = (in_dat >= 97) & (in_dat <= 122)
    assert torch.equal(up.data[:total], torch.where(lower, in_dat - 32, in_dat))
    del lower

    # substr(s, 2, 5): lengths min(5, len - 1); offsets are their exclusive prefix sums
    lens = (in_off[1:] - in_off[:-1]).to(torch.int64)
    sub_len = torch.clamp(lens - 1, min=0, max=5)
    want_off = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(sub_len, 0, out=want_off[1:])
    sub_off = sub.offsets.view(torch.int32)[:n + 1]
    assert torch.equal(sub_off.to(torch.int64), want_off)
    assert sub.data_used == int(want_off[-1])
    # the first byte of every substr output row is the second byte of the input row
    first = sub.data[sub_off[:-1].to(torch.int64)]
    assert torch.equal(first, in_dat[in_off[:-1].to(torch.int64) + 1])
    del first, want_off, sub_len, lens

    # like '%spark%': rows that contain the five bytes, found on the host over the flat buffer
    hit = np.ones(total - 4, dtype=bool)
    for k, ch in enumerate(b"spark"):
        hit &= data[k:total - 4 + k] == ch
    pos = np.flatnonzero(hit)
    rows = np.searchsorted(offsets, pos, side="right") - 1
    inside = pos + 5 <= offsets[rows + 1]
    want_rows = np.unique(rows[inside])
    bits = np.unpackbits(like.data[:(n + 7) // 8].cpu().numpy(), bitorder="little")[:n]
    assert np.array_equal(np.flatnonzero(bits), want_rows)
    assert bool((like.validity[:n // 8] == 0xff).all())

    for lo in (0, (n // 2) & ~63):
        m = min(WIN, n - lo)
        o = offsets[lo:lo + m + 1].astype(np.int64)
        arr = pa.Array.from_buffers(pa.string(), m, [None, pa.py_buffer((o - o[0]).astype(np.int32)),
                                                     pa.py_buffer(data[o[0]:o[-1]].copy())])
        hb = pa.RecordBatch.from_arrays([arr], schema=W.c5_schema())
        want = oracle.project(exprs, hb)
        wl = np.unpackbits(np.frombuffer(want[0].buffers()[1], dtype=np.uint8), bitorder="little")[:m]
        assert np.array_equal(bits[lo:lo + m], wl)
        for out, w in ((sub, want[1]), (up, want[2])):
            oo = out.offsets.view(torch.int32)[lo:lo + m + 1].cpu().numpy().astype(np.int64)
            got = pa.Array.from_buffers(pa.string(), m, [None, pa.py_buffer((oo - oo[0]).astype(np.int32)),
                                                         pa.py_buffer(out.data[oo[0]:oo[-1]].cpu().numpy())])
            assert_bit_exact(got, w, f"rows [{lo}, {lo + m})")
