"""Offline fuzz beyond the suite's own seeds (round 5): the string and materialised-value tree generators of
tests/test_fuzz_trees.py over seeds the suite does not use — row mode and under UINT32 selection vectors (the wave-shaped
selection-mode kernels) — plus a generator that mixes in the round-5 functions (initcap, the digests, the regexp subset),
every result bit for bit against the oracle.

  python tools/fuzz_offline.py [first_seed] [count]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyarrow as pa
import gandiva_amd as gandiva
from oracle import oracle
from helpers import assert_bit_exact
import test_fuzz_trees as F

STR, I32, I64, BOOL = pa.string(), pa.int32(), pa.int64(), pa.bool_()


class Round5Gen(F.TailTreeGen):
    def string(self, depth):
        b, r = self.b, self.rng
        if depth > 0 and r.random() < 0.3:
            roll = r.random()
            if roll < 0.35:
                return b.make_function("initcap", [self.string(depth - 1)], STR)
            if roll < 0.6:
                return b.make_function(self.pick(["hashMD5", "sha1", "hashSHA256"]), [self.string(depth - 1)], STR)
            if roll < 0.8:
                return b.make_function("regexp_replace", [self.string(depth - 1), b.make_literal(self.pick(["spark", "ar", "é"]), STR),
                                                          b.make_literal(self.pick(["", "X", "flink"]), STR)], STR)
            return b.make_function("sha256", [self.f["k"]], STR)
        return super().string(depth)

    def boolean(self, depth):
        b, r = self.b, self.rng
        if depth > 0 and r.random() < 0.15:
            return b.make_function(self.pick(["regexp_like", "regexp_matches"]),
                                   [super().string(depth - 1), b.make_literal(self.pick(["spark", "^sp", "rk$", "^spark$", "a"]), STR)], BOOL)
        return super().boolean(depth)


def round5_expressions(seed):
    g = Round5Gen(F._string_batch(0, 1).schema, 7000 + seed)
    exprs = [g.b.make_expression(g.string(3), pa.field("s0", STR)), g.b.make_expression(g.boolean(3), pa.field("b0", BOOL)),
             g.b.make_expression(g.string(2), pa.field("s1", STR)), g.b.make_expression(g.integer(3), pa.field("i0", I32))]
    return exprs, g.b.make_condition(g.boolean(3))


class LateGen(Round5Gen):
    """late round 5: general regular expressions, to_date, castVARCHAR(float), replace / lpad / rpad with per-row arguments"""
    REGEX = [r"sp.rk", r"\d+", r"^[a-z]+$", r"(ab|sp)+", r"[^a-z ]", r"a.*k$", r"é+", r"\w+\s\w+", r"^.{0,3}$", r"(?:ar|XY){1,2}", r"_%?"]

    def string(self, depth):
        b, r = self.b, self.rng
        if depth > 0 and r.random() < 0.35:
            roll = r.random()
            if roll < 0.25:
                return b.make_function("replace", [self.string(depth - 1), super().string(0), self.pick([b.make_literal("-", STR), super().string(0)])], STR)
            if roll < 0.5:
                return b.make_function(self.pick(["lpad", "rpad"]), [super().string(depth - 1), b.make_function("castINT", [self.f["k"]], I32), super().string(0)], STR)
            if roll < 0.75:
                x = b.make_function("divide", [b.make_function("castFLOAT8", [self.f["k"]], pa.float64()), b.make_literal(self.pick([3.0, 7.0, 1e-9, 1e9]), pa.float64())], pa.float64())
                return b.make_function("castVARCHAR", [x, b.make_literal(int(r.integers(0, 26)), I64)], STR)
            return b.make_function("castVARCHAR", [b.make_function("castFLOAT4", [self.f["k"]], pa.float32()), b.make_literal(30, I64)], STR)
        return super().string(depth)

    def boolean(self, depth):
        b, r = self.b, self.rng
        if depth > 0 and r.random() < 0.3:
            roll = r.random()
            if roll < 0.7:
                return b.make_function(self.pick(["regexp_like", "regexp_matches"]), [Round5Gen.string(self, depth - 1) if r.random() < 0.5 else self.f["s"],
                                                                                    b.make_literal(self.pick(self.REGEX), STR)], BOOL)
            d = b.make_function("to_date", [self.f["t"], b.make_literal(self.pick(["YYYY-MM-DD", "DD MON YY", "YYYY/MM/DD HH24"]), STR), b.make_literal(1, I32)], pa.date64())
            return b.make_function("isnull", [d], BOOL)
        return super().boolean(depth)


def late_expressions(seed):
    g = LateGen(F._string_batch(0, 1).schema, 9000 + seed)
    exprs = [g.b.make_expression(g.string(3), pa.field("s0", STR)), g.b.make_expression(g.boolean(3), pa.field("b0", BOOL)),
             g.b.make_expression(g.string(2), pa.field("s1", STR)), g.b.make_expression(g.boolean(2), pa.field("b1", BOOL))]
    return exprs, g.b.make_condition(g.boolean(3))


def check(tag, exprs, cond, batch):
    want = oracle.project(exprs, batch)
    got = gandiva.make_projector(batch.schema, exprs, None).evaluate(batch)
    for g, w, e in zip(got, want, exprs):
        g.validate(full=True)
        assert_bit_exact(g, w, f"{tag}: {e}")
    sv = gandiva.make_filter(batch.schema, cond).evaluate(batch, pa.default_memory_pool(), "int32")
    sel = sv.to_array()
    assert sel.equals(oracle.filter_indices(cond, batch, "int32")), f"{tag}: {cond}"
    if len(sel):
        got_sel = gandiva.make_projector(batch.schema, exprs, None, "UINT32").evaluate(batch, sv)
        for g, w, e in zip(got_sel, want, exprs):
            assert_bit_exact(g, oracle.take_rows(w, sel), f"{tag} (UINT32 selection): {e}")
    return len(sel)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sizes = [1, 63, 64, 65, 257, 511, 513, 1000, 4097, 12011, 30011]
plans = failures = 0
for seed in range(first, first + count):
    n = sizes[seed % len(sizes)]
    batch = F._string_batch(seed, n)
    makers = (("string", F._string_expressions), ("tail", F._tail_expressions), ("round5", round5_expressions), ("late", late_expressions))
    for name, maker in [m for m in makers if not os.environ.get("FUZZ_ONLY") or m[0] == os.environ["FUZZ_ONLY"]]:
        try:
            exprs, cond = maker(seed)
            check(f"{name} seed {seed} n {n}", exprs, cond, batch)
            plans += 1
        except Exception as e:   # noqa: BLE001
            failures += 1
            print(f"FAILED {name} seed {seed} n {n}: {type(e).__name__}: {str(e)[:400]}", flush=True)
print(f"offline fuzz: seeds {first}..{first + count - 1}: {plans} plans (projector + filter + UINT32 selection projector) bit-exact vs the oracle, {failures} failures")
