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


def binaries():
    """The two builds of tests/dropin/dropin_driver.c.  They are compiled by __graft_entry__.build() in the
    container that has the reference's headers and travel to the GPU box with the snapshot: a missing binary
    is a broken build, not a reason to skip."""
    hip, ref = os.path.join(BUILD, "dropin_hip"), os.path.join(BUILD, "dropin_ref")
    assert os.path.exists(hip) and os.path.exists(ref), \
        "drop-in binaries missing: run __graft_entry__.build() where /root/reference exists (make -C tests/dropin)"
    return hip, ref


def run_both(args):
    hip, ref = binaries()
    out_ref = subprocess.run([ref] + args, capture_output=True, text=True, timeout=900)
    out_hip = subprocess.run([hip] + args, capture_output=True, text=True, timeout=900)
    assert out_ref.returncode == 0, out_ref.stdout + out_ref.stderr
    assert out_hip.returncode == 0, out_hip.stdout + out_hip.stderr
    return out_hip.stdout, out_ref.stdout


@pytest.mark.parametrize("name,pivot,nrhs", [("test_mat", 3, 1), ("10teams", 3, 1), ("prob159", 3, 1), ("10teams", 0, 1),
                                             ("gen_n300", 5, 1), ("test_mat", 3, 2), ("gen_n300", 3, 3)])
def test_dropin_factorize_matches_reference(tmp_path, name, pivot, nrhs):
    """SLIP_LU_factorize + SLIP_LU_solve through libslip_lu_hip.so == the reference (L, U, rhos, pinv, x hashes),
    the reference's own SLIP_check_solution accepts the result (and rejects a corrupted b), and the
    reference's destructors free what the shim allocated."""
    entry, fix = load_case(name)
    trip = str(tmp_path / "A.txt")
    write_triplet(trip, fix, entry["n"])
    got, want = run_both([trip, str(pivot), str(nrhs)])
    assert got.startswith("check=0 check_corrupt=-4 ")
    assert got == want


def test_dropin_error_paths_match_reference():
    """NULL arguments -> SLIP_INCORRECT_INPUT, a singular matrix -> SLIP_SINGULAR under three pivot schemes,
    then SLIP_delete_sparse on whatever came back (Tcov/cov_test.c:342-345,466-472,674-678)."""
    got, want = run_both(["--errors"])
    assert "singular=-2" in got and "null_all factorize=-3 solve=-3" in got
    assert got == want


def test_dropin_on_rl5934_with_conversion_times(tmp_path, record_property):
    """an LP basis of the reference's benchmark set through the whole drop-in path: the triplet file is written by the
    product's own writer (slip_hip_write_triplet) and read by the REFERENCE's SLIP_tripread inside both binaries; the shim
    reports what its one-pass conversions cost next to the device time (SLIP_HIP_SHIM_TIMING)"""
    import re
    import slip_lu_amd as sl
    entry, fix = load_case("rl5934")
    trip = str(tmp_path / "A.txt")
    sl.write_triplet(trip, entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"])
    assert all(np.array_equal(np.asarray(a).astype(np.uint64), np.asarray(b).astype(np.uint64))
               for a, b in zip(sl.read_triplet(trip)[1:], (fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"])))
    hip, ref = binaries()
    args = [trip, "3", "1"]
    out_ref = subprocess.run([ref] + args, capture_output=True, text=True, timeout=900)
    out_hip = subprocess.run([hip] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, SLIP_HIP_SHIM_TIMING="1"))
    assert out_ref.returncode == 0 and out_hip.returncode == 0, out_hip.stdout + out_hip.stderr
    assert out_hip.stdout.startswith("check=0 check_corrupt=-4 ") and out_hip.stdout == out_ref.stdout
    m = re.search(r"slip_lu_hip timing: A to slabs ([0-9.]+) ms, upload\+factorise ([0-9.]+) ms \(kernel ([0-9.]+) ms\), download ([0-9.]+) ms, "
                  r"L/U/rhos to mpz ([0-9.]+) ms", out_hip.stderr)
    assert m, out_hip.stderr
    print("\nrl5934 drop-in:", m.group(0))
    for key, val in zip(("a_to_slabs_ms", "upload_factorise_ms", "kernel_ms", "download_ms", "to_mpz_ms"), m.groups()):
        record_property(key, float(val))
