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


def test_shared_arrays_cannot_be_dereferenced_plainly(tmp_path):
    """cross-worker data is a distinct pointer type (slip_shared<T>, ref_lu_pipe.h): `P.pinv[i]` -- a load the compiler may keep
    in a register, a store that never leaves the L1 -- does not compile; the accessor form does"""
    import subprocess
    csrc = os.path.join(ROOT, "slip_lu_amd", "csrc")
    emu = os.path.join(ROOT, "tests", "emu")
    head = '#include "hip_rt_emu.h"\n#include "fiber_emu.h"\n#include "ref_lu_pipe.h"\n'
    good = head + "int f(const SlipParams &P) { return slip_ld_i32(P.pinv.at(3)) + slip_ld_i32(P.Lready.at(1)) + (int) slip_ld_u32(P.pkg.at() + 7); }\n"
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-DSLIP_EMULATE", "-I", emu, "-I", csrc, "-x", "c++"]
    src = tmp_path / "good.cpp"; src.write_text(good)
    assert subprocess.run(cmd + [str(src)], capture_output=True).returncode == 0
    for expr in ("P.pinv[3]", "P.row_perm[0]", "P.piv[2].len", "P.Lready[1]", "*P.pkg", "P.jobs[0]", "P.invd[5]", "P.sw_row[0] + P.sw_pos[0]"):
        bad = head + "int f(const SlipParams &P) { return (int) (%s); }\n" % expr
        src = tmp_path / "bad.cpp"; src.write_text(bad)
        assert subprocess.run(cmd + [str(src)], capture_output=True).returncode != 0, expr
