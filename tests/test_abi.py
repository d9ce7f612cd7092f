"""The C-ABI library loads and exports every symbol include/slip_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from slip_lu_amd import _lib


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(slip_hip_\w+|SLIP_LU_\w+|SLIP_hip_\w+)\s*\(", text)))


def test_header_symbols_exported():
    if not os.path.exists(_lib.DEFAULT_SO):
        pytest.skip("HIP library not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(_lib.DEFAULT_SO)
    names = declared_functions("slip_hip.h")
    assert set(_lib.EXPORTS) <= set(names)
    for nm in names:
        assert hasattr(lib, nm), f"{nm} declared in include/slip_hip.h but not exported"


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: creating a factorisation without a HIP device is an error, not a silent path."""
    import numpy as np
    import slip_lu_amd as sl
    if not os.path.exists(_lib.DEFAULT_SO):
        pytest.skip("HIP library not built")
    lib = _lib.load()
    if lib.slip_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    Ap = np.array([0, 1, 2], dtype=np.int64)
    Ai = np.array([0, 1], dtype=np.int32)
    with pytest.raises(sl.SlipError) as e:
        sl.factorize(2, Ap, Ai, np.ones(2, np.int32), np.ones(2, np.uint64), np.arange(2, dtype=np.int32))
    assert e.value.code == -100


def test_missing_library_raises():
    with pytest.raises(OSError):
        _lib.load("/nonexistent/libslip_hip.so")
