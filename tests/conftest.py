import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import slabfile  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def golden_index():
    return {e["name"]: e for e in json.load(open(os.path.join(GOLDEN, "index.json")))}


def load_case(name):
    """(index entry, dict with Ap/Ai/Alen/Alimbs/q + whatever reference output the fixture keeps)."""
    import oracle_lib
    entry = golden_index()[name]
    fix = slabfile.load(os.path.join(GOLDEN, name + ".slab.gz"))
    if entry["input"].startswith("gen:"):
        n, d, b, seed = entry["input"][4:].split(",")
        Ap, Ai, Ax = oracle_lib.matgen(int(n), float(d), int(b), int(seed))
        fix["Ap"], fix["Ai"] = Ap, Ai
        fix["Alen"] = np.sign(Ax).astype(np.int32)
        fix["Alimbs"] = np.abs(Ax).astype(np.uint64)
    return entry, fix


def check_against_golden(entry, fix, res, counters=True):
    """res: canonical factor dict of an implementation; compares with the reference's record (counters=False: the factors
    only -- a run that continued from a given prefix has not done the prefix's work)."""
    assert res["K"] == entry["K"], (res["K"], entry["K"])
    assert np.array_equal(res["pinv"], fix["pinv"]), "pinv differs"
    if entry["K"] > 0:
        assert slabfile.factor_digest(res) == entry["digest"], "factor digest differs from the reference"
    if entry["full"]:
        for k in slabfile.FACTOR_KEYS:
            assert np.array_equal(np.asarray(res[k]).astype(np.int64), np.asarray(fix[k]).astype(np.int64)), k
    if not counters:
        return
    c = entry["counters"]
    got = res["counters"]
    assert (int(got[0]), int(got[1]), int(got[2]), int(got[3]), int(got[4]), int(got[5])) == \
        (c["N_upd"], c["B_read"], c["B_write"], c["N_src"], c["L_streamed"], c["maxlimbs"]), (got, c)


@pytest.fixture(scope="session")
def hip_lib_path():
    from slip_lu_amd import _lib
    return _lib.DEFAULT_SO


def solve_inputs(case):
    """(n, Ap, Ai, Alen, Alimbs, q, fixture) of one entry of tests/golden/solve_index.json"""
    import numpy as np
    import slabfile
    import oracle_lib
    fix = slabfile.load(os.path.join(GOLDEN, case["name"] + ".slab.gz"))
    if case["input"].startswith("gen:"):
        a, d, b, seed = case["input"][4:].split(",")
        Ap, Ai, Ax = oracle_lib.matgen(int(a), float(d), int(b), int(seed))
        Alen, Alimbs = np.sign(Ax).astype(np.int32), np.abs(Ax).astype(np.uint64)
    else:
        Ap, Ai, Alen, Alimbs = fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"]
    return case["n"], Ap, Ai, Alen, Alimbs, fix["q"], fix


def check_solve(case, lib_path=None, nrhs=1, **kw):
    """device solve (slip_hip_factor_solve) == orc_solve numerators, and == the reference's rationals"""
    from fractions import Fraction
    import numpy as np
    import oracle_lib
    import slip_lu_amd as sl
    n, Ap, Ai, Alen, Alimbs, q, fix = solve_inputs(case)
    b = oracle_lib.solve_rhs(n)
    f = sl.Factorization(n, Ap, Ai, Alen, Alimbs, q, lib_path=lib_path, **kw)
    try:
        f.run(0)
        bs = np.concatenate([b if c % 2 == 0 else -3 * b + c for c in range(nrhs)])
        blen = np.sign(bs).astype(np.int32); blimbs = np.abs(bs[bs != 0]).astype(np.uint64)
        xlen, xlimbs = f.solve(blen, blimbs, nrhs=nrhs)
    finally:
        f.close()
    got = oracle_lib.bigints(xlen, xlimbs)
    want, det = oracle_lib.factorize_and_solve(n, Ap, Ai, Alen, Alimbs, q, b)
    assert got[:n] == want
    ref_num = oracle_lib.bigints(fix["xnumlen"], fix["xnumlimbs"])
    ref_den = oracle_lib.bigints(fix["xdenlen"], fix["xdenlimbs"])
    for i in range(n):
        assert Fraction(got[i], det) == Fraction(ref_num[i], ref_den[i]), i
    for c in range(1, nrhs):
        bc = b if c % 2 == 0 else -3 * b + c
        wc, _ = oracle_lib.factorize_and_solve(n, Ap, Ai, Alen, Alimbs, q, bc)
        assert got[c * n:(c + 1) * n] == wc, c
