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
