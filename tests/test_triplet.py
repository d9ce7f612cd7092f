"""Triplet files <-> limb slabs: the host-side reader/writer of the slab ABI against what the reference's
SLIP_tripread + SLIP_build_sparse_trip_mpz hold (SLIP_LU/Demo/demos.c:245-331, Source/slip_trip_to_mat.c:23-69) -- the A
arrays of the golden fixtures were dumped from the reference after it had read these very files.  No GPU involved."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_case

REF_MATS = "/root/reference/SLIP_LU/ExampleMats"


def same(a, b):
    return all(np.array_equal(np.asarray(x).astype(np.int64) if i != 3 else np.asarray(x).astype(np.uint64),
                              np.asarray(y).astype(np.int64) if i != 3 else np.asarray(y).astype(np.uint64)) for i, (x, y) in enumerate(zip(a, b)))


def test_reader_matches_reference_on_fixture():
    """tests/golden/test_mat_triplet.txt is the reference's ExampleMats/test_mat.txt (a data file of its test suite)"""
    import slip_lu_amd as sl
    entry, fix = load_case("test_mat")
    n, Ap, Ai, Alen, Al = sl.read_triplet(os.path.join(GOLDEN, "test_mat_triplet.txt"))
    assert n == entry["n"]
    assert same((Ap, Ai, Alen, Al), (fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"]))


@pytest.mark.parametrize("case,fname", [("10teams", "10teams_mat.txt"), ("prob159", "prob159_mat.txt"), ("NSR8K", "NSR8K_mat.txt")])
def test_reader_matches_reference_on_example_mats(case, fname):
    import slip_lu_amd as sl
    path = os.path.join(REF_MATS, fname)
    if not os.path.exists(path):
        pytest.skip("the reference's ExampleMats are not on this machine")
    entry, fix = load_case(case)
    n, Ap, Ai, Alen, Al = sl.read_triplet(path)
    assert n == entry["n"]
    assert same((Ap, Ai, Alen, Al), (fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"]))


def test_round_trip_of_long_values_zero_based_and_duplicates(tmp_path):
    """multi-limb values of both signs through write -> read; a 0-based file (first entry holds a 0); duplicates and the file
    order inside a column are kept (slip_trip_to_mat.c:47-64)"""
    import slip_lu_amd as sl
    rng = np.random.default_rng(7)
    n = 23
    cols = [sorted(rng.choice(n, size=rng.integers(1, 6), replace=False).tolist()) for _ in range(n)]
    Ap = np.zeros(n + 1, np.int64); Ai = []; vals = []
    for c in range(n):
        for r in cols[c]:
            Ai.append(r)
            nbits = int(rng.integers(1, 400))
            v = int.from_bytes(rng.bytes((nbits + 7) // 8), "little") >> ((-nbits) % 8)
            vals.append(-v if rng.integers(2) else v)
        Ap[c + 1] = len(Ai)
    Alen = []; Al = []
    for v in vals:
        m = abs(v); l = []
        while m:
            l.append(m & (2 ** 64 - 1)); m >>= 64
        Alen.append(-len(l) if v < 0 else len(l)); Al += l
    Ai = np.array(Ai, np.int32); Alen = np.array(Alen, np.int32); Al = np.array(Al, np.uint64)
    p = str(tmp_path / "t.txt")
    sl.write_triplet(p, n, Ap, Ai, Alen, Al)
    assert same(sl.read_triplet(p)[1:], (Ap, Ai, Alen, Al))
    # the decimal text is what Python reads as the same integers
    lines = open(p).read().split("\n")[1:-1]
    assert [int(x.split()[2]) for x in lines] == vals
    # 0-based file with a duplicate: entries (0,0)=5, (2,1)=-7, (2,1)=9, (1,2)=1
    q = str(tmp_path / "z.txt")
    open(q, "w").write("3 3 4\n0 0 5\n2 1 -7\n2 1 9\n1 2 1\n")
    n0, Ap0, Ai0, Alen0, Al0 = sl.read_triplet(q)
    assert n0 == 3 and Ap0.tolist() == [0, 1, 3, 4] and Ai0.tolist() == [0, 2, 2, 1]
    assert Alen0.tolist() == [1, -1, 1, 1] and Al0.tolist() == [5, 7, 9, 1]


@pytest.mark.parametrize("text", ["", "3 3\n", "3 3 2\n1 1 1\n", "3 3 1\n1 1 x\n", "3 3 1\n4 1 1\n", "3 4 1\n1 1 1\n", "3 3 1\n1 1 1.5\n"])
def test_malformed_input_is_rejected(tmp_path, text):
    """SLIP_tripread returns SLIP_INCORRECT_INPUT for a short or malformed file (demos.c:259-262, 283-288, 309-315)"""
    import slip_lu_amd as sl
    p = str(tmp_path / "bad.txt")
    open(p, "w").write(text)
    with pytest.raises(sl.SlipError) as e:
        sl.read_triplet(p)
    assert e.value.code == -3
    for bad in ("bad_mat1.txt", "bad_mat2.txt"):
        path = os.path.join(REF_MATS, bad)
        if os.path.exists(path):
            with pytest.raises(sl.SlipError):
                sl.read_triplet(path)
