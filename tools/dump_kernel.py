"""Dev tool: print the generated HIP source of a workload's fused kernel and (optionally)
cross-compile it for gfx950 and show resource usage + ISA histogram.  No GPU needed.

  python tools/dump_kernel.py c2 [--isa]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gandiva_amd as g  # noqa: E402
from gandiva_amd import _capi, gandiva as gg, workloads as W  # noqa: E402


def plan_source(which):
    """Returns the generated source by building the plan through the precompile entry and
    reading the kernel back from a Projector/Filter is impossible without a GPU, so the
    library is asked for the IR via a throw-away cache dir."""
    lib = _capi.lib()
    d = tempfile.mkdtemp()
    os.environ["GANDIVA_AMD_CACHE_DIR"] = d
    if which == "fp":   # the fused filter -> project kernel (C3's condition, a + b, uint32 selection vector)
        sh = gg._make_schema(W.c3_schema())
        cond, ex = W.c3_condition(), W.c3_sum_expression()
        arr = (C.c_void_p * len(ex))(*[e._h for e in ex])
        rc = lib.gdv_precompile_filter_project(sh, cond._h, arr, len(ex), int(os.environ.get("FP_MODE", "2")))
    elif which == "c3":
        sh = gg._make_schema(W.c3_schema())
        cond = W.c3_condition()
        rc = lib.gdv_precompile_filter(sh, cond._h)
    else:
        schema, exprs = {"c1": (W.c1_schema, W.c1_expressions), "c2": (W.c2_schema, W.c2_expressions),
                         "c4": (W.c4_schema, W.c4_expressions), "c5": (W.c5_schema, W.c5_expressions)}[which]
        ex = exprs()
        sh = gg._make_schema(schema())
        arr = (C.c_void_p * len(ex))(*[e._h for e in ex])
        rc = lib.gdv_precompile_projector(sh, arr, len(ex), 0)
    if rc:
        raise SystemExit(_capi.last_error())
    return d


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "c2"
    os.environ["GDV_DUMP_SOURCE"] = "1"
    d = plan_source(which)
    for f in os.listdir(d):
        p = os.path.join(d, f)
        if f.endswith(".hip"):
            print(open(p).read())
        if f.endswith(".hsaco") and "--isa" in sys.argv:
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", p], capture_output=True, text=True).stdout
            open("/tmp/gdv_last.s", "w").write(out)
            import collections
            hist = collections.Counter(l.split()[0] for l in out.splitlines() if l.startswith("\t") and l.split())
            for k, v in hist.most_common(40):
                print(f"{v:6d} {k}")
            meta = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", p], capture_output=True, text=True).stdout
            for l in meta.splitlines():
                if any(s in l for s in ("vgpr_count", "sgpr_count", "spill", "lds_size", "scratch", ".name:")):
                    print(l.strip())
