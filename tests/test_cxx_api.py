"""The gandiva:: C++ API (gandiva_amd/cxx: headers named as the reference's, libgandiva.so)
exercised by a C++ program the way a C++ caller of the reference would use it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "gandiva_amd", "cxx")
BIN = os.path.join(CXX, "tests", "test_gandiva_cxx")


def _build():
    subprocess.check_call(["make", "-C", CXX, "all", "test_cxx"], stdout=subprocess.DEVNULL)


def test_cxx_api_host_only():
    _build()
    out = subprocess.run([BIN, "--host-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK (host-only)" in out.stdout


@pytest.mark.gpu
def test_cxx_api_reference_kats_on_gpu():
    _build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("OK")


@pytest.mark.gpu
def test_a_cxx_process_that_ends_while_its_kernels_are_still_compiling_exits_cleanly(tmp_path):
    """Round 6: with an EMPTY code-object cache the program above finishes on tier 0 within a second of its first Make — and
    then crashed or hung in exit(): the compiler's function-local statics, registered during the compilation in flight, were
    torn down under the worker thread.  The main thread's thread_local guard (gdv_runtime.cc) joins the worker first."""
    _build()
    for attempt in range(3):
        cache = tmp_path / f"cache{attempt}"
        cache.mkdir()
        env = dict(os.environ, GANDIVA_AMD_CACHE_DIR=str(cache))
        env.pop("GDV_NO_TIER0", None)
        out = subprocess.run([BIN], capture_output=True, text=True, timeout=120, env=env)
        assert out.returncode == 0, f"attempt {attempt}: rc {out.returncode}\n" + out.stdout[-2000:] + out.stderr[-2000:]
        assert out.stdout.strip().endswith("OK")
