"""Drop-in acceptance: a program written against the reference's SLIP_LU.h and library whose
SLIP_LU_factorize is served by libslip_lu_hip.so must (a) pass the reference's own exact check
SLIP_check_solution (SLIP_check_solution.c:31-113, as Demo/SLIPLU.c:287 does) and (b) hand back
L, U, rhos, pinv identical to the reference's (same report line, same hash), with the reference's
destructors freeing everything."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_case

pytestmark = pytest.mark.gpu
BUILD = os.path.join(ROOT, "tests", "dropin", "_build")


def write_triplet(path, fix, n):
    """The reference's triplet text format (Demo/demos.c:245-331), 1-based."""
    Ap, Ai, Alen, Alimbs = fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"]
    off = np.concatenate([[0], np.cumsum(np.abs(Alen))])
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(Ai)}\n")
        for j in range(n):
            for p in range(int(Ap[j]), int(Ap[j + 1])):
                v = 0
                for t in range(abs(int(Alen[p]))):
                    v |= int(Alimbs[off[p] + t]) << (64 * t)
                if Alen[p] < 0:
                    v = -v
                f.write(f"{int(Ai[p]) + 1} {j + 1} {v}\n")


@pytest.mark.parametrize("name,pivot", [("test_mat", 3), ("10teams", 3), ("prob159", 3), ("10teams", 0), ("gen_n300", 5)])
def test_dropin_factorize_matches_reference(tmp_path, name, pivot):
    hip, ref = os.path.join(BUILD, "dropin_hip"), os.path.join(BUILD, "dropin_ref")
    if not (os.path.exists(hip) and os.path.exists(ref)):
        pytest.skip("drop-in binaries not built (need the reference's headers: make -C tests/dropin)")
    entry, fix = load_case(name)
    trip = str(tmp_path / "A.txt")
    write_triplet(trip, fix, entry["n"])
    out_ref = subprocess.run([ref, trip, str(pivot)], capture_output=True, text=True, timeout=600)
    out_hip = subprocess.run([hip, trip, str(pivot)], capture_output=True, text=True, timeout=600)
    assert out_ref.returncode == 0, out_ref.stdout + out_ref.stderr
    assert out_hip.returncode == 0, out_hip.stdout + out_hip.stderr
    assert out_hip.stdout.startswith("check=0 ")
    assert out_hip.stdout == out_ref.stdout
