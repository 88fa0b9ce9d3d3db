"""CPU: the C-ABI library builds, loads without a GPU, and exports every symbol that
include/cloops_hip.h declares; no compute is attempted."""
import ctypes
import os
import re

import numpy as np
import pytest

from cloops_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    with open(os.path.join(ROOT, "include", "cloops_hip.h")) as fh:
        src = fh.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cl_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported():
    so = build.build()
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert getattr(lib, n) is not None, n
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_plumbing_without_gpu():
    lib = _lib.load()
    assert lib.cl_version() >= 100
    if lib.cl_device_count() == 0:
        # no CPU fallback: creating a chromosome must fail loudly
        from cloops_amd import api
        with pytest.raises(_lib.CloopsHipError) as ei:
            api.Chromosome(np.arange(4), np.arange(4) + 5)
        assert ei.value.code == _lib.CL_ERR_NODEVICE
        from cloops_amd.cDBSCAN2 import cDBSCAN
        with pytest.raises(_lib.CloopsHipError):
            cDBSCAN(np.array([[0, 1, 2], [1, 2, 3]]), 5, 2)


def test_missing_library_is_an_import_error(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libcloops_hip.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_header_is_plain_c(tmp_path):
    """include/cloops_hip.h is C (not C++): the plain-C consumer of tests/c compiles against it with -Werror"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-c",
                           os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", os.path.join(str(tmp_path), "a.o")])
