"""The HIP kernel SOURCE run lane-by-lane on the CPU (tests/emu) against the reference's goldens.

Not the product (the product is the hipcc build, tested under -m gpu); this catches logic errors in
the wave-level limb arithmetic and the column loop without a GPU, on tiny inputs only.
"""
import os
import subprocess

import pytest

from conftest import ROOT, check_against_golden, load_case

EMU = os.path.join(ROOT, "tests", "emu", "libslip_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu.so"])
    return EMU


@pytest.mark.parametrize("name,waves", [("test_mat", 2), ("test_mat_p1", 1), ("test_mat_p2", 4), ("test_mat_p4tol", 2),
                                        ("test_mat_p5", 2), ("test_mat_tol01", 2), ("test_mat_noord", 16),
                                        ("gen_n40", 2)])
def test_emulated_kernel_matches_reference(emu_lib, name, waves):
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    res = sl.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"],
                       pivot=entry["pivot"], tol=entry["tol"], kmax=entry["kmax"], limb_cap=entry["cap"],
                       waves=waves, lib_path=emu_lib)
    check_against_golden(entry, fix, res)
