"""Builds the reference lineage's OWN Python binding — pyarrow/gandiva.pyx, compiled from
where it lies inside the installed pyarrow, never copied into this repo — against the
gandiva:: C++ API of gandiva_amd/cxx (libgandiva.so).  Output:
gandiva_amd/_pyarrow/gandiva.<abi>.so (git-ignored), loadable as `pyarrow.gandiva` through
gandiva_amd.pyarrow_gandiva.load().  With it, pyarrow/tests/test_gandiva.py runs UNMODIFIED
against the HIP backend (tests/test_pyarrow_gandiva.py)."""
import os
import subprocess
import sys
import sysconfig

import numpy as np
import pyarrow as pa

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT_DIR = os.path.join(PKG, "_pyarrow")


def build(force=False):
    """pyarrow's gandiva.pyx -> _pyarrow/gandiva.<abi>.so (returned), and this directory's host_pool.pyx (a
    pyarrow.MemoryPool over gandiva::HostMemoryPool) -> _pyarrow/host_pool.<abi>.so."""
    pa_dir = os.path.dirname(pa.__file__)
    out = _build_one(os.path.join(pa_dir, "gandiva.pyx"), "gandiva", force)
    _build_one(os.path.join(HERE, "host_pool.pyx"), "host_pool", force)
    return out


def _build_one(pyx, name, force):
    pa_dir = os.path.dirname(pa.__file__)
    site = os.path.dirname(pa_dir)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(OUT_DIR, name + ext)
    lib = os.path.join(PKG, "libgandiva.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(lib), os.path.getmtime(pyx)):
        return out
    os.makedirs(OUT_DIR, exist_ok=True)
    cpp = os.path.join(OUT_DIR, name + ".cpp")
    subprocess.check_call([sys.executable, "-m", "cython", "--cplus", "-3", "-I", site, pyx, "-o", cpp])
    arrow_so = sorted(f for f in os.listdir(pa_dir) if f.startswith("libarrow.so."))[0]
    cmd = ["g++", "-std=c++20", "-O1", "-g0", "-fPIC", "-shared", "-w", cpp, "-o", out,
           "-I", os.path.join(HERE, "include"), "-I", pa.get_include(), "-I", np.get_include(),
           "-I", sysconfig.get_paths()["include"],
           "-L", PKG, "-lgandiva", "-lgandiva_amd", "-L", pa_dir, "-l:libarrow_python.so", "-l:" + arrow_so,
           "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + pa_dir]
    subprocess.check_call(cmd)
    os.remove(cpp)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
