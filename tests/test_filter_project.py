"""Fused filter -> project (round 4, gdv_filter_project_*): ONE kernel does what the reference's callers
chain — Filter::Evaluate -> SelectionVector -> Projector::Evaluate(batch, selection_vector)
(pyarrow/tests/test_gandiva.py:329-373).  The oracle states the chain: filter_indices, take_rows, project;
the fused results must equal it bit for bit — values, validity bits, bool outputs, the emitted selection
vector in all three index widths — at every tile boundary, null density and selectivity, through host
buffers, HBM-resident buffers and the asynchronous entry; functions that can raise run on selected rows only.
"""
import ctypes as C

import numpy as np
import pyarrow as pa
import pytest

import gandiva_amd as gandiva
from gandiva_amd import _capi, gandiva as gg, workloads as W
from oracle import oracle
from helpers import assert_bit_exact, random_array

I32, I64, F64, BOOL = pa.int32(), pa.int64(), pa.float64(), pa.bool_()


def _batch(rng, n, null_fraction):
    types = [I64, I64, F64, I32, BOOL]
    cols = [random_array(rng, t, n, null_fraction) for t in types]
    # small value ranges so that predicates select a tunable share of the rows and divisors hit zero
    a = pa.array(rng.integers(0, 1000, n), I64, mask=~np.asarray(cols[0].is_valid()) if null_fraction else None)
    b = pa.array(rng.integers(0, 8, n), I64, mask=~np.asarray(cols[1].is_valid()) if null_fraction else None)
    return pa.RecordBatch.from_arrays([a, b, cols[2], cols[3], cols[4]], names=["a", "b", "x", "k", "f"])


def _plan(schema, threshold):
    """condition a > threshold (selectivity (999 - threshold) / 1000) and five expressions: int64 / float64
    arithmetic, a bool output, if/else with value-dependent validity, and a division that raises on b = 0."""
    bld = gandiva.TreeExprBuilder()
    a, b, x, k, f = (bld.make_field(schema.field(i)) for i in range(5))
    cond = bld.make_condition(bld.make_function("greater_than", [a, bld.make_literal(threshold, I64)], BOOL))
    nz = bld.make_function("not_equal", [b, bld.make_literal(0, I64)], BOOL)
    exprs = [
        bld.make_expression(bld.make_function("add", [a, b], I64), pa.field("s", I64)),
        bld.make_expression(bld.make_function("multiply", [x, x], F64), pa.field("xx", F64)),
        bld.make_expression(bld.make_or([bld.make_function("less_than", [a, b], BOOL), f]), pa.field("lt_or_f", BOOL)),
        bld.make_expression(bld.make_if(nz, bld.make_function("divide", [a, b], I64), bld.make_literal(-1, I64), I64),
                            pa.field("q", I64)),
        bld.make_expression(bld.make_function("add", [k, bld.make_literal(7, I32)], I32), pa.field("k7", I32)),
    ]
    return cond, exprs


def _chain(cond, exprs, batch, dtype):
    sel = oracle.filter_indices(cond, batch, dtype or "int32")
    return sel, oracle.project(exprs, oracle.take_rows(batch, sel.to_numpy()))


# ------------------------------------------------------------------------------------------ CPU: planning

def test_fused_plans_compile_for_gfx950_and_refuse_what_they_do_not_take():
    lib = _capi.lib()
    rng = np.random.default_rng(0)
    batch = _batch(rng, 100, 0.1)
    cond, exprs = _plan(batch.schema, 500)
    sh = gg._make_schema(batch.schema)
    try:
        arr = (C.c_void_p * len(exprs))(*[e._h for e in exprs])
        for mode in (0, 1, 2, 3):
            assert lib.gdv_precompile_filter_project(sh, cond._h, arr, len(exprs), mode) == 0, _capi.last_error()
        assert lib.gdv_precompile_filter_project(sh, cond._h, arr, 0, 0) != 0
        assert lib.gdv_precompile_filter_project(sh, None, arr, len(exprs), 0) != 0
    finally:
        lib.gdv_schema_free(sh)
    # var-len outputs / columns are not a fused shape: CodeGenError (40), the callers chain
    bld = gandiva.TreeExprBuilder()
    ss = pa.schema([pa.field("s", pa.string()), pa.field("a", I64)])
    s, a = bld.make_field(ss.field(0)), bld.make_field(ss.field(1))
    up = bld.make_expression(bld.make_function("upper", [s], pa.string()), pa.field("u", pa.string()))
    c2 = bld.make_condition(bld.make_function("greater_than", [a, bld.make_literal(1, I64)], BOOL))
    sh = gg._make_schema(ss)
    try:
        arr = (C.c_void_p * 1)(up._h)
        assert lib.gdv_precompile_filter_project(sh, c2._h, arr, 1, 2) == 40
        ln = bld.make_expression(bld.make_function("octet_length", [s], I32), pa.field("l", I32))
        arr = (C.c_void_p * 1)(ln._h)
        assert lib.gdv_precompile_filter_project(sh, c2._h, arr, 1, 2) == 40
    finally:
        lib.gdv_schema_free(sh)


def test_without_a_device_make_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    batch = _batch(np.random.default_rng(1), 10, 0.0)
    cond, exprs = _plan(batch.schema, 500)
    with pytest.raises(pa.lib.ArrowException, match="no HIP device"):
        gandiva.make_filter_project(batch.schema, cond, exprs, "int32")


# ------------------------------------------------------------------------------------------ GPU: parity

LENGTHS = [1, 63, 64, 65, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8193, 100_003]


@pytest.mark.gpu
@pytest.mark.parametrize("n", LENGTHS)
@pytest.mark.parametrize("nulls", [0.0, 0.15])
def test_fused_equals_the_chain_host_buffers(n, nulls):
    rng = np.random.default_rng(n * 7 + int(nulls * 100))
    batch = _batch(rng, n, nulls)
    cond, exprs = _plan(batch.schema, 870)
    fp = gandiva.make_filter_project(batch.schema, cond, exprs, "int32")
    assert fp.fused
    got, sel = fp.evaluate(batch)
    want_sel, want = _chain(cond, exprs, batch, "int32")
    assert sel.to_array().equals(want_sel)
    for e, (g, w) in enumerate(zip(got, want)):
        assert_bit_exact(g, w, f"expression {e}, {n} rows")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [None, "int16", "int32", "int64"])
@pytest.mark.parametrize("threshold", [-1, 499, 998, 5000])   # every row / half / 0.1 % / none
def test_fused_equals_the_chain_hbm_resident_all_index_widths_and_selectivities(dtype, threshold):
    import torch
    n = 50_001 if dtype == "int16" else 300_007
    rng = np.random.default_rng(threshold + 11)
    batch = _batch(rng, n, 0.1)
    cond, exprs = _plan(batch.schema, threshold)
    fp = gandiva.make_filter_project(batch.schema, cond, exprs, dtype)
    assert fp.fused
    db = gandiva.DeviceBatch.from_arrow(batch)
    outs, sel = fp.evaluate_device(db)
    torch.cuda.synchronize()
    want_sel, want = _chain(cond, exprs, batch, dtype)
    if dtype is None:
        assert sel is None
    else:
        assert sel.num_slots == len(want_sel) and sel.to_array().equals(want_sel)
    for e, (o, w) in enumerate(zip(outs, want)):
        assert o.num_rows == len(want_sel)
        assert_bit_exact(o.to_arrow(), w, f"expression {e}")
    # again into the same buffers, asynchronously: the count stays on the device until asked for
    # again into the same buffers (this plan can raise, so the call waits for the error word)
    outs2, sel2 = fp.evaluate_device(db, outputs=outs, indices=None if sel is None else sel.indices, sync=False)
    torch.cuda.synchronize()
    for e, (o, w) in enumerate(zip(outs2, want)):
        assert_bit_exact(o.to_arrow(), w, f"expression {e}, second call")
    # a plan that cannot raise, asynchronously: nothing waits, the count stays on the device until asked for
    fp3 = gandiva.make_filter_project(batch.schema, cond, exprs[:3], dtype)
    outs3, sel3 = fp3.evaluate_device(db, sync=False)
    assert outs3[0].count_tensor is not None and outs3[0].length == n      # sized for the capacity, count pending
    if sel3 is not None:
        assert sel3.pending
    torch.cuda.synchronize()
    for e, (o, w) in enumerate(zip(outs3, want[:3])):
        assert_bit_exact(o.to_arrow(), w, f"expression {e}, asynchronous")
        assert o.num_rows == len(want_sel)


@pytest.mark.gpu
def test_functions_that_raise_run_on_selected_rows_only():
    """a / b with b = 0 in rows the condition rejects must not raise (the chain's projector never sees
    them); in a selected row it must."""
    n = 20_000
    a = np.arange(n, dtype=np.int64)
    b = np.where(a % 2 == 0, 0, 3).astype(np.int64)        # every even row would divide by zero
    batch = pa.RecordBatch.from_arrays([pa.array(a), pa.array(b)], names=["a", "b"])
    bld = gandiva.TreeExprBuilder()
    fa, fb = bld.make_field(batch.schema.field(0)), bld.make_field(batch.schema.field(1))
    div = bld.make_expression(bld.make_function("divide", [fa, fb], I64), pa.field("q", I64))
    odd = bld.make_condition(bld.make_function("equal", [bld.make_function("mod", [fa, bld.make_literal(2, I64)], I64),
                                                         bld.make_literal(1, I64)], BOOL))
    fp = gandiva.make_filter_project(batch.schema, odd, [div], "int32")
    assert fp.fused
    got, sel = fp.evaluate(batch)
    want_sel, want = _chain(odd, [div], batch, "int32")
    assert sel.to_array().equals(want_sel)
    assert_bit_exact(got[0], want[0])
    anyrow = bld.make_condition(bld.make_function("greater_than", [fa, bld.make_literal(-1, I64)], BOOL))
    with pytest.raises(pa.lib.ArrowException, match="divide by zero"):
        gandiva.make_filter_project(batch.schema, anyrow, [div], "int32").evaluate(batch)


@pytest.mark.gpu
def test_plans_that_are_not_fused_take_the_chain_behind_the_same_interface():
    n = 10_007
    rng = np.random.default_rng(5)
    s = pa.array(["".join(rng.choice(list("abcXYZ"), rng.integers(0, 9))) for _ in range(n)], pa.string())
    a = pa.array(rng.integers(0, 100, n), I64)
    batch = pa.RecordBatch.from_arrays([s, a], names=["s", "a"])
    bld = gandiva.TreeExprBuilder()
    fs, fa = bld.make_field(batch.schema.field(0)), bld.make_field(batch.schema.field(1))
    cond = bld.make_condition(bld.make_function("greater_than", [fa, bld.make_literal(60, I64)], BOOL))
    exprs = [bld.make_expression(bld.make_function("upper", [fs], pa.string()), pa.field("u", pa.string())),
             bld.make_expression(bld.make_function("add", [fa, fa], I64), pa.field("aa", I64))]
    fp = gandiva.make_filter_project(batch.schema, cond, exprs, "int32")
    assert not fp.fused
    got, sel = fp.evaluate(batch)
    want_sel, want = _chain(cond, exprs, batch, "int32")
    assert sel.to_array().equals(want_sel)
    for g, w in zip(got, want):
        assert_bit_exact(g, w)


@pytest.mark.gpu
def test_c_abi_argument_checks():
    lib = _capi.lib()
    batch = _batch(np.random.default_rng(2), 1000, 0.0)
    cond, exprs = _plan(batch.schema, 500)
    fp = gandiva.make_filter_project(batch.schema, cond, exprs, "int16")
    big = _batch(np.random.default_rng(3), 70_000, 0.0)
    with pytest.raises(pa.lib.ArrowException, match="uint16 selection vector cannot address"):
        fp.evaluate(big)
    # an empty batch: zero rows out, nothing launched
    empty = batch.slice(0, 0)
    got, sel = fp.evaluate(empty)
    assert sel.num_slots == 0 and all(len(g) == 0 for g in got)
    assert lib.gdv_filter_project_num_outputs(fp._h) == len(exprs)
    assert "gdv_fp_lookback" in fp.llvm_ir and "@expr_0" in fp.llvm_ir


@pytest.mark.gpu
def test_fused_filter_project_at_c3_scale_against_torch():
    """2 * 10^8 int64 rows (the tools/filter_project_chain.py case at a size the suite can afford):
    every selected row's a + b and the whole selection vector against torch.nonzero / gather."""
    import torch
    n = 200_000_000
    db = W.c3_device_batch(n)
    bld = gandiva.TreeExprBuilder()
    a, c = (bld.make_field(W.c3_schema().field(i)) for i in range(2))
    expr = bld.make_expression(bld.make_function("add", [a, c], I64), pa.field("s", I64))
    fp = gandiva.make_filter_project(W.c3_schema(), W.c3_condition(), [expr], "int32")
    outs, sel = fp.evaluate_device(db)
    torch.cuda.synchronize()
    ta, tb = (col.data.view(torch.int64) for col in db.columns)
    want_idx = torch.nonzero((ta > W.C3_K1) & (tb < W.C3_K2)).view(-1)
    k = want_idx.numel()
    assert sel.num_slots == k
    assert torch.equal(sel.indices[:k].to(torch.int64) & 0xffffffff, want_idx)
    assert torch.equal(outs[0].data[:8 * k].view(torch.int64), (ta + tb)[want_idx])
    assert bool((outs[0].validity[:k // 8] == 0xff).all())
    if k % 8:
        assert int(outs[0].validity[k // 8]) == (1 << (k % 8)) - 1


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_trees_through_the_fused_operator_match_the_oracle_chain(seed):
    """The random numeric trees of tests/test_fuzz_trees.py (every fixed-width type, if/else, three-valued
    AND / OR, IN lists, casts, bool outputs) as condition + projections of ONE fused kernel, in all three
    index widths and without a selection vector, against the oracle's filter -> take -> project."""
    import test_fuzz_trees as F
    exprs, cond = F._expressions(seed)
    n = [1, 63, 64, 65, 1000, 8191, 8193, 60001][seed % 8]
    batch = F._batch(seed, n)
    dtype = [None, "int16", "int32", "int64"][seed % 4]
    fp = gandiva.make_filter_project(batch.schema, cond, exprs, dtype)
    assert fp.fused, "numeric trees are a fused shape"
    got, sel = fp.evaluate(batch)
    want_sel = oracle.filter_indices(cond, batch, dtype or "int32")
    if dtype is not None:
        assert sel.to_array().equals(want_sel), f"seed {seed}: {cond}"
    want = oracle.project(exprs, batch)
    for g, w, e in zip(got, want, exprs):
        assert_bit_exact(g, oracle.take_rows(w, want_sel.to_numpy()), f"seed {seed}: {e}")


# ------------------------------------------------------------------------------------------ round 5: the windowed shape

def test_the_plan_carries_both_shapes_and_the_direct_one_takes_the_windowed_argument_block(monkeypatch, tmp_path):
    """PlanFilterProject: the windowed kernel (GDV_FP_CAP rows of LDS window per wave tile) is the plan, the direct
    round-4 kernel its `exact` variant; rows too wide for the window, or a plan without any windowed output, keep
    the direct shape alone.  (CPU: both are planned and compiled for gfx950.)"""
    from test_planner_cpu import _precompile_fp
    batch = _batch(np.random.default_rng(0), 100, 0.1)
    cond, exprs = _plan(batch.schema, 500)
    files = _precompile_fp(monkeypatch, tmp_path / "a", batch.schema, cond, exprs, 2)
    texts = [open(tmp_path / "a" / f).read() for f in files]
    win = [t for t in texts if "GDV_FP_CAP" in t]
    assert len(win) == 1 and len(texts) == 2, files
    assert "#define GDV_FP_CAP 256" in win[0]            # 9984 bytes / (8 + 8 + 8 + 4 + 4 bytes per row) -> 256 rows
    assert "#define GDV_FP_K 3" in win[0] and "for (int kb = 0; kb < GDV_FP_K; kb++)" in win[0]
    assert "gdv_bits_flush_local(" in win[0] and "gdv_one<" in win[0] and "win0[slot]" in win[0]
    direct = [t for t in texts if "GDV_FP_CAP" not in t][0]
    assert "gdv_bits_flush(" in direct and "out0[opos]" in direct
    # bool outputs only + no selection vector: nothing to window
    files = _precompile_fp(monkeypatch, tmp_path / "b", batch.schema, cond, exprs[2:3], 0)
    assert len(files) == 1 and "GDV_FP_CAP" not in open(tmp_path / "b" / files[0]).read()
    # GDV_FP_WINDOW=0: the direct shape alone
    monkeypatch.setenv("GDV_FP_WINDOW", "0")
    files = _precompile_fp(monkeypatch, tmp_path / "c", batch.schema, cond, exprs, 2)
    assert len(files) == 1 and "GDV_FP_CAP" not in open(tmp_path / "c" / files[0]).read()


@pytest.mark.gpu
def test_the_kernel_follows_the_selectivity_of_recent_batches():
    """First batch: the windowed kernel whatever it selects (rows beyond the window take its re-read path — still
    bit-exact).  A synchronous evaluation that selected more than the window holds moves the next batches to the
    direct kernel; a sparse batch brings them back."""
    n = 90_001
    rng = np.random.default_rng(77)
    batch = _batch(rng, n, 0.1)
    bld = gandiva.TreeExprBuilder()
    results = {}
    for thr in (-1, 930):            # every valid row / ~7 %
        cond, exprs = _plan(batch.schema, thr)
        results[thr] = (cond, exprs, *_chain(cond, exprs, batch, "int32"))
    # one FilterProject per threshold would never change shape: build ONE whose threshold is a column
    thr_col = pa.array(np.full(n, 0, np.int64))
    sch = batch.schema.append(pa.field("t", I64))
    fa, ft = bld.make_field(sch.field(0)), bld.make_field(sch.field(5))
    cond = bld.make_condition(bld.make_function("greater_than", [fa, ft], BOOL))
    _, exprs = _plan(sch, 0)
    fp = gandiva.make_filter_project(sch, cond, exprs, "int32")
    assert fp.fused and fp.kernel_shape == 0
    shapes = []
    for thr in (-1, -1, 930, 930, -1):
        b2 = pa.RecordBatch.from_arrays(batch.columns + [pa.array(np.full(n, thr, np.int64))], schema=sch)
        shapes.append(fp.kernel_shape)
        got, sel = fp.evaluate(b2)
        want_sel, want = _chain(cond, exprs, b2, "int32")
        assert sel.to_array().equals(want_sel), thr
        for e, (g, w) in enumerate(zip(got, want)):
            assert_bit_exact(g, w, f"threshold {thr}, expression {e}, kernel shape {shapes[-1]}")
    assert shapes == [0, 1, 1, 0, 0], shapes


@pytest.mark.gpu
def test_a_stalled_launch_is_re_run_on_the_chain():
    """GDV_FP_FORCE_STALL=1 (read once per process: a subprocess) makes the engine treat every fused launch as one
    whose look-back gave up; evaluate then answers through Filter + selection-mode Projector — same results, no
    ExecutionError (round 4 returned "device scan stalled")."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, pyarrow as pa
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import gandiva_amd as gandiva
        from oracle import oracle
        import test_filter_project as T
        from helpers import assert_bit_exact
        batch = T._batch(np.random.default_rng(5), 33_333, 0.1)
        cond, exprs = T._plan(batch.schema, 700)
        for dtype in ("int32", None):
            fp = gandiva.make_filter_project(batch.schema, cond, exprs[:3] + exprs[4:], dtype)
            assert fp.fused
            got, sel = fp.evaluate(batch)
            want_sel, want = T._chain(cond, exprs[:3] + exprs[4:], batch, dtype)
            if dtype: assert sel.to_array().equals(want_sel)
            for g, w in zip(got, want): assert_bit_exact(g, w)
            dgot, dsel = fp.evaluate_device(gandiva.DeviceBatch.from_arrow(batch))
            for g, w in zip(dgot, want): assert_bit_exact(g.to_arrow(), w)
        print("chain ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GDV_FP_FORCE_STALL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "chain ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("rounds", [1, 2, 3])
@pytest.mark.parametrize("n", [1535, 1536, 1537, 12287, 12288, 12289, 12288 * 3 + 1536 * 5 + 77, 300_007])
def test_wave_tiles_of_several_rounds(monkeypatch, rounds, n):
    """The windowed kernel's wave tile is GDV_FP_K rounds of GDV_U sub-tiles (one look-back per K x 8192 rows at the
    C3 shape; here U = 8: 512 x K rows per wave, 4096 x K per workgroup): every round / wave-tile / workgroup-tile
    boundary, sparse (everything fits the window), medium and dense (every wave tile overflows into the re-read
    path) selections; values, validity, bool outputs and the selection vector against the oracle's chain."""
    monkeypatch.setenv("GDV_FP_K", str(rounds))
    rng = np.random.default_rng(n + rounds)
    batch = _batch(rng, n, 0.1)
    for thr, dtype in ((870, "int32"), (300, None), (-1, "int64")):
        cond, exprs = _plan(batch.schema, thr)
        fp = gandiva.make_filter_project(batch.schema, cond, exprs, dtype)
        fp.set_tuning("kernel", 0)            # stay on the windowed kernel whatever the batch selects
        assert fp.kernel_shape == 0 and f"#define GDV_FP_K {rounds}" in fp.llvm_ir
        for rep in range(2):
            got, sel = fp.evaluate(batch)
            want_sel, want = _chain(cond, exprs, batch, dtype)
            if dtype is not None:
                assert sel.to_array().equals(want_sel), (thr, rep)
            for e, (g, w) in enumerate(zip(got, want)):
                assert_bit_exact(g, w, f"threshold {thr}, expression {e}, {rounds} rounds, run {rep}")


@pytest.mark.gpu
def test_asynchronous_evaluations_move_the_kernel_shape_too():
    """An asynchronous evaluation never sees its count on the host; the operator learns it one call late through a
    pinned word of its own (the kernel that publishes the count writes it there as well)."""
    import torch
    n = 120_000
    rng = np.random.default_rng(3)
    batch = _batch(rng, n, 0.0)
    cond, exprs = _plan(batch.schema, -1)          # every row selected
    fp = gandiva.make_filter_project(batch.schema, cond, exprs[:3], "int32")   # (a plan that cannot raise: truly asynchronous)
    db = gandiva.DeviceBatch.from_arrow(batch)
    assert fp.kernel_shape == 0
    outs, sel = fp.evaluate_device(db, sync=False)
    torch.cuda.synchronize()
    assert fp.kernel_shape == 0                    # nothing has read the pinned word yet
    outs, sel = fp.evaluate_device(db, outputs=outs, indices=sel.indices, sync=False)   # reads what the first call left
    torch.cuda.synchronize()
    assert fp.kernel_shape == 1
    want_sel, want = _chain(cond, exprs[:3], batch, "int32")
    assert sel.num_slots == len(want_sel)
    for o, w in zip(outs, want):
        assert_bit_exact(o.to_arrow(), w, "asynchronous, direct kernel")


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [0, 1])
def test_tiles_come_from_a_ticket_not_from_the_dispatch_order(monkeypatch, shape):
    """Round 6 (verdict item 5): a workgroup's tile is the ticket it draws when it starts, so the look-back only ever waits
    for workgroups that are already running.  GDV_FP_EXPERIMENT=2 makes the workgroups with LOW block indices arrive
    late (each group of 256 consecutive block indices draws its tickets in roughly reversed order): tile != blockIdx
    for most workgroups, and the results are still the chain's, bit for bit — windowed and direct kernel, several
    hundred workgroup tiles, and no launch fell back to the chain (GDV_TRACE-free check: the kernel shape stays pinned)."""
    monkeypatch.setenv("GDV_FP_EXPERIMENT", "2")
    n = 4096 * 3 * 700 + 1234          # ~700 workgroup tiles of the windowed shape (U = 8, K = 3), more of the direct one
    rng = np.random.default_rng(77 + shape)
    batch = _batch(rng, n, 0.05)
    cond, exprs = _plan(batch.schema, 700)
    exprs = exprs[:3]
    fp = gandiva.make_filter_project(batch.schema, cond, exprs, "int32")
    assert "wg_ticket" in fp.llvm_ir and "__builtin_amdgcn_s_sleep(32)" in fp.llvm_ir
    fp.set_tuning("kernel", shape)
    db = gandiva.DeviceBatch.from_arrow(batch)
    want_sel, want = _chain(cond, exprs, batch, "int32")
    for rep in range(2):
        outs, sel = fp.evaluate_device(db)
        assert sel.num_slots == len(want_sel)
        assert sel.to_array().equals(want_sel)
        for e, (o, w) in enumerate(zip(outs, want)):
            assert_bit_exact(o.to_arrow(), w, f"late low blocks, kernel shape {shape}, expression {e}, run {rep}")
