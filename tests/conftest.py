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


def check_against_golden(entry, fix, res):
    """res: canonical factor dict of an implementation; compares with the reference's record."""
    assert res["K"] == entry["K"], (res["K"], entry["K"])
    assert np.array_equal(res["pinv"], fix["pinv"]), "pinv differs"
    if entry["K"] > 0:
        assert slabfile.factor_digest(res) == entry["digest"], "factor digest differs from the reference"
    if entry["full"]:
        for k in slabfile.FACTOR_KEYS:
            assert np.array_equal(np.asarray(res[k]).astype(np.int64), np.asarray(fix[k]).astype(np.int64)), k
    c = entry["counters"]
    got = res["counters"]
    assert (int(got[0]), int(got[1]), int(got[2]), int(got[3]), int(got[4]), int(got[5])) == \
        (c["N_upd"], c["B_read"], c["B_write"], c["N_src"], c["L_streamed"], c["maxlimbs"]), (got, c)


@pytest.fixture(scope="session")
def hip_lib_path():
    from slip_lu_amd import _lib
    return _lib.DEFAULT_SO
