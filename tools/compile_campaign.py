"""Planner + hipRTC over seeds the suite does not use (CPU only: gfx950 is cross-compiled): every random tree generator of the tests and of
tools/fuzz_offline.py — numeric, string, materialised-value, round-5 functions, the late additions — is planned and compiled as a projector (row
mode and UINT32 selection mode) and as a filter.  A CodeGenError for a shape the backend documents as refused is not a failure; anything else is.
      python tools/compile_campaign.py <first_seed> <last_seed>"""
import ctypes as C, importlib.util, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GANDIVA_AMD_CACHE_DIR", tempfile.mkdtemp())
from gandiva_amd import _capi, gandiva as gg
import test_fuzz_trees as F
sys.argv, argv = [sys.argv[0], "0", "0"], sys.argv
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_offline.py"))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
lo, hi = int(argv[1]), int(argv[2])
lib = _capi.lib()
makers = [("numeric", lambda s: F._expressions(3000 + s), lambda s: F._batch(s, 8).schema),
          ("string", F._string_expressions, lambda s: F._string_batch(s, 8).schema),
          ("tail", F._tail_expressions, lambda s: F._string_batch(s, 8).schema),
          ("round5", fz.round5_expressions, lambda s: F._string_batch(s, 8).schema),
          ("late", fz.late_expressions, lambda s: F._string_batch(s, 8).schema)]
plans = refused = bad = 0
for seed in range(lo, hi):
    for name, maker, schema_of in makers:
        try:
            exprs, cond = maker(seed)
            sh = gg._make_schema(schema_of(seed))
            arr = (C.c_void_p * len(exprs))(*[e._h for e in exprs])
            for what, rc in (("projector", lib.gdv_precompile_projector(sh, arr, len(exprs), 0)), ("selection projector", lib.gdv_precompile_projector(sh, arr, len(exprs), 2)),
                             ("filter", lib.gdv_precompile_filter(sh, cond._h))):
                plans += 1
                if rc == 0:
                    continue
                msg = _capi.last_error()
                if "not supported yet" in msg:
                    refused += 1
                else:
                    bad += 1
                    print("FAIL", name, seed, what, rc, msg[:300].replace("\n", " "), flush=True)
            if name == "numeric":   # ... and as the fused filter -> project operator (both kernel shapes), UINT32 indices
                plans += 1
                rc = lib.gdv_precompile_filter_project(sh, cond._h, arr, len(exprs), 2)
                if rc != 0 and ("not supported yet" in _capi.last_error() or "chain" in _capi.last_error()):
                    refused += 1
                elif rc != 0:
                    bad += 1
                    print("FAIL", name, seed, "filter-project", rc, _capi.last_error()[:300].replace("\n", " "), flush=True)
            lib.gdv_schema_free(sh)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", name, seed, type(e).__name__, str(e)[:300].replace("\n", " "), flush=True)
print(f"seeds {lo}..{hi - 1}: {plans} plans compiled for gfx950, {refused} refused as documented, {bad} failures")
