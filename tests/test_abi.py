"""The C-ABI library loads on a CPU-only host and exports every symbol that
include/tao_amodal_hip.h declares (no compute calls without a GPU)."""
import os
import re

import numpy as np

from tao_amodal_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    text = open(os.path.join(ROOT, "include", "tao_amodal_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(taoamd_[a-z_0-9]+)\s*\(", text))
    assert declared, "no declarations found"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_ingest_header_symbols_are_exported():
    import ctypes
    text = open(os.path.join(ROOT, "include", "tao_amodal_ingest.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(taoamd_[a-z_0-9]+)\s*\(", text))
    assert len(declared) == 20, declared
    lib = ctypes.CDLL(os.path.join(ROOT, "tao_amodal_amd", "libtao_amodal_ingest.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_thresholds_are_numpy_linspace_bit_for_bit():
    lib = _lib.load()
    a, b = np.zeros(10), np.zeros(101)
    assert lib.taoamd_thresholds_host(a.ctypes.data, b.ctypes.data) == 0
    assert np.array_equal(a, np.linspace(0.5, 0.95, 10))
    assert np.array_equal(b, np.linspace(0.0, 1.0, 101))


def test_status_strings():
    lib = _lib.load()
    assert lib.taoamd_strerror(0) == b"ok"
    assert b"workspace" in lib.taoamd_strerror(4)
    assert lib.taoamd_version() >= 100
