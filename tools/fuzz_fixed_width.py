"""Round 6 campaign over the fixed-width tree generator of tests/test_fuzz_trees.py, on seeds and batch lengths the suite
does not use — aimed at what this round changed underneath every fixed-width plan: the 16-sub-tile wave tile of the
projection kernels (batch lengths around multiples of 1024 and 4096), the tier-0 interpreter (run the script once as it is
— every plan's FIRST Evaluate is then interpreted while hipRTC compiles behind it, and the plan is evaluated again once the
specialised kernel has arrived — and once under GDV_NO_TIER0=1), the fused filter -> project kernel with its ticket, and the
one-call host-sharded evaluation over three device contexts.  Every result bit for bit against the oracle.

  python tools/fuzz_fixed_width.py [first_seed] [count]
  python tools/fuzz_fixed_width.py core [first_seed] [count]     trees drawn from the tier-0 CORE only (add / subtract / multiply,
      the six comparisons, not, isnull / isnotnull, the numeric casts, if, AND / OR over int32 / int64 / float32 / float64 / bool):
      every plan has a post-fix program, its first Evaluate is interpreted, the later ones run the specialised kernel
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyarrow as pa
import gandiva_amd as gandiva
from gandiva_amd import _capi, shard
from oracle import oracle
from helpers import assert_bit_exact
import test_fuzz_trees as F

lib = _capi.lib()
CORE = len(sys.argv) > 1 and sys.argv[1] == "core"
if CORE:
    del sys.argv[1]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
SIZES = [1023, 1024, 1025, 4095, 4097, 16383, 16385, 65535, 65537, 131073, 262143, 500009]
gandiva.set_virtual_devices(3)
plans = failures = interpreted = 0
t0 = time.time()


def wait_specialised(run, seconds=60):
    """run() until no tier-0 launch is counted for it (the background compiler has delivered)"""
    if os.environ.get("GDV_FORCE_TIER0"):
        return run()   # (always interpreted: nothing to wait for)
    t = time.time()
    while time.time() - t < seconds:
        before = lib.gdv_tier0_launches()
        out = run()
        if lib.gdv_tier0_launches() == before:
            return out
        time.sleep(0.05)
    raise RuntimeError("the specialised kernel did not arrive")


I32, I64, F32, F64, BOOL = pa.int32(), pa.int64(), pa.float32(), pa.float64(), pa.bool_()
NUM = [I32, I64, F32, F64]
CASTS = {I64: [("castBIGINT", I32), ("castBIGINT", F32), ("castBIGINT", F64)], I32: [("castINT", I64), ("castINT", F32), ("castINT", F64)],
         F32: [("castFLOAT4", I32), ("castFLOAT4", I64), ("castFLOAT4", F64)], F64: [("castFLOAT8", I32), ("castFLOAT8", I64), ("castFLOAT8", F32)]}
CORE_SCHEMA = pa.schema([pa.field(f"{n}{i}", t) for n, t in (("i", I32), ("l", I64), ("f", F32), ("d", F64)) for i in range(2)] + [pa.field("b0", BOOL)])


class CoreGen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(31_000 + seed)
        self.b = gandiva.TreeExprBuilder()
        self.fields = {t: [self.b.make_field(f) for f in CORE_SCHEMA if f.type == t] for t in NUM + [BOOL]}

    def pick(self, xs):
        return xs[int(self.rng.integers(0, len(xs)))]

    def lit(self, t):
        r = self.rng
        if t in (I32, I64):
            return self.b.make_literal(int(r.integers(-1000, 1000)), t)
        return self.b.make_literal(float(self.pick([0.0, -1.5, 2.25, 1e6, 1e-3, 3.0])), t)

    def gen(self, t, depth):
        b, r = self.b, self.rng
        if depth <= 0 or r.random() < 0.15:
            return self.pick(self.fields[t]) if (t == BOOL or r.random() < 0.8) else self.lit(t)
        roll = r.random()
        if t == BOOL:
            if roll < 0.4:
                u = self.pick(NUM)
                return b.make_function(self.pick(["equal", "not_equal", "less_than", "less_than_or_equal_to", "greater_than", "greater_than_or_equal_to"]),
                                       [self.gen(u, depth - 1), self.gen(u, depth - 1)], BOOL)
            if roll < 0.5:
                return b.make_function("not", [self.gen(BOOL, depth - 1)], BOOL)
            if roll < 0.65:
                return b.make_function(self.pick(["isnull", "isnotnull"]), [self.gen(self.pick(NUM + [BOOL]), depth - 1)], BOOL)
            if roll < 0.85:
                kids = [self.gen(BOOL, depth - 1) for _ in range(int(r.integers(2, 4)))]
                return b.make_and(kids) if r.random() < 0.5 else b.make_or(kids)
            return b.make_if(self.gen(BOOL, depth - 1), self.gen(BOOL, depth - 1), self.gen(BOOL, depth - 1), BOOL)
        if roll < 0.55:
            return b.make_function(self.pick(["add", "subtract", "multiply"]), [self.gen(t, depth - 1), self.gen(t, depth - 1)], t)
        if roll < 0.8:
            name, src = self.pick(CASTS[t])
            return b.make_function(name, [self.gen(src, depth - 1)], t)
        return b.make_if(self.gen(BOOL, depth - 1), self.gen(t, depth - 1), self.gen(t, depth - 1), t)


def core_case(seed, n):
    g = CoreGen(seed)
    types = [g.pick(NUM) if k % 3 else BOOL for k in range(5)]
    exprs = [g.b.make_expression(g.gen(t, 4), pa.field(f"o{k}", t)) for k, t in enumerate(types)]
    cond = g.b.make_condition(g.gen(BOOL, 3))
    rng = np.random.default_rng(41_000 + seed)
    cols = []
    for f in CORE_SCHEMA:
        mask = rng.random(n) < (0.0 if f.name.endswith("1") else 0.15)
        if f.type in (I32, I64):
            v = rng.integers(-2000, 2000, n).astype(f.type.to_pandas_dtype())
            v[rng.random(n) < 0.02] = np.iinfo(v.dtype).max   # products and sums that wrap, casts that saturate
        elif f.type == BOOL:
            v = rng.random(n) < 0.5
        else:
            v = (rng.standard_normal(n) * 1e3).astype(f.type.to_pandas_dtype())
            v[rng.random(n) < 0.02] = np.inf
            v[rng.random(n) < 0.02] = np.nan
            v[rng.random(n) < 0.02] = 3e18
        cols.append(pa.array(v, type=f.type, mask=mask))
    return exprs, cond, pa.RecordBatch.from_arrays(cols, schema=CORE_SCHEMA)


for seed in range(first, first + count):
    n = SIZES[seed % len(SIZES)]
    tag = f"{'core ' if CORE else ''}seed {seed} n {n}"
    try:
        if CORE:
            exprs, cond, batch = core_case(seed, n)
        else:
            exprs, cond = F._expressions(seed)
            batch = F._batch(seed, n)
        want = oracle.project(exprs, batch)
        want_sel = oracle.filter_indices(cond, batch, "int32")
        # -- projector: the first Evaluate (tier 0 where the plan is inside its core), then the specialised kernel
        proj = gandiva.make_projector(batch.schema, exprs, None)
        before = lib.gdv_tier0_launches()
        got = proj.evaluate(batch)
        interpreted += lib.gdv_tier0_launches() > before
        for g, w, e in zip(got, want, exprs):
            assert_bit_exact(g, w, f"{tag} first Evaluate: {e}")
        got = wait_specialised(lambda: proj.evaluate(batch))
        for g, w, e in zip(got, want, exprs):
            assert_bit_exact(g, w, f"{tag}: {e}")
        # -- filter, both tiers
        flt = gandiva.make_filter(batch.schema, cond)
        sel = flt.evaluate(batch, pa.default_memory_pool(), "int32").to_array()
        assert sel.equals(want_sel), f"{tag} first Evaluate: {cond}"
        sv = wait_specialised(lambda: flt.evaluate(batch, pa.default_memory_pool(), "int32"))
        assert sv.to_array().equals(want_sel), f"{tag}: {cond}"
        # -- selection-mode projector, fused filter -> project
        if len(want_sel):
            got_sel = gandiva.make_projector(batch.schema, exprs, None, "UINT32").evaluate(batch, sv)
            for g, w, e in zip(got_sel, want, exprs):
                assert_bit_exact(g, oracle.take_rows(w, want_sel), f"{tag} (UINT32 selection): {e}")
        fp = gandiva.make_filter_project(batch.schema, cond, exprs, "int32")
        arrays, fsel = fp.evaluate(batch)
        assert fsel.to_array().equals(want_sel), f"{tag} filter-project indices (fused {fp.fused})"
        for g, w, e in zip(arrays, want, exprs):
            assert_bit_exact(g, oracle.take_rows(w, want_sel), f"{tag} filter-project (fused {fp.fused}): {e}")
        # -- one call, three device contexts, host buffers
        got = shard.evaluate_projector_host_sharded(proj, batch, [0, 1, 2])
        for g, w, e in zip(got, want, exprs):
            assert_bit_exact(g, w, f"{tag} host-sharded: {e}")
        ssv = shard.evaluate_filter_host_sharded(flt, batch, [0, 1, 2], "int32")
        assert ssv.to_array().equals(want_sel), f"{tag} host-sharded filter"
        plans += 1
    except Exception as e:   # noqa: BLE001
        failures += 1
        print(f"FAILED {tag}: {type(e).__name__}: {str(e)[:500]}", flush=True)
print(f"{'tier-0 core' if CORE else 'fixed-width'} campaign ({'GDV_NO_TIER0' if os.environ.get('GDV_NO_TIER0') else 'GDV_FORCE_TIER0' if os.environ.get('GDV_FORCE_TIER0') else 'tier 0 on'}): seeds {first}..{first + count - 1}, "
      f"{plans} seeds x (projector both tiers, filter both tiers, UINT32 selection projector, fused filter-project, host-sharded x3) "
      f"bit-exact vs the oracle, {failures} failures; first Evaluate interpreted for {interpreted} projectors; {time.time() - t0:.0f} s")
