"""GPU: the C ABI used from plain C -- include/cloops_hip.h compiled as C99 by gcc, linked against
libcloops_hip.so, no Python or C++ on the caller's side (tests/c/abi_smoke.c)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_consumer(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from cloops_amd import _lib
    so = _lib.lib_path() if hasattr(_lib, "lib_path") else os.path.join(ROOT, "cloops_amd", "libcloops_hip.so")
    exe = os.path.join(str(tmp_path), "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe,
                           "-L", os.path.dirname(so), "-lcloops_hip", "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi smoke ok" in out.stdout
