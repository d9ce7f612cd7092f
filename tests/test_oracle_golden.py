"""The CPU restatement (oracle/ref_lu_oracle.c) against the compiled reference's outputs.

This pins the oracle: every fixture under tests/golden was produced by
oracle/_ref/ref_driver, i.e. by the unmodified reference (tests/golden/make_golden.py).
Bit-exact: integer work.
"""
import pytest

import oracle_lib
from conftest import check_against_golden, golden_index, load_case

FAST = [n for n, e in golden_index().items()
        if n not in ("NSR8K", "C5_n200k_c64", "C3_n50k_c32", "gen_n2000_pm1", "fome12", "rail4284")]


@pytest.mark.parametrize("name", FAST)
def test_oracle_matches_reference(name):
    entry, fix = load_case(name)
    res = oracle_lib.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"],
                               pivot=entry["pivot"], tol=entry["tol"], kmax=entry["kmax"], cap=entry["cap"])
    assert res["status"] == entry["status"]
    check_against_golden(entry, fix, res)


def test_survey_anchors():
    """N_upd / B_read / B_write of the reference replay."""
    idx = golden_index()
    assert (idx["test_mat"]["counters"]["N_upd"], idx["test_mat"]["counters"]["B_read"],
            idx["test_mat"]["counters"]["B_write"]) == (168, 4232, 1500)
    assert idx["10teams"]["counters"]["N_upd"] == 102983 and idx["10teams"]["counters"]["B_read"] == 1523196
    assert idx["prob159"]["counters"]["N_upd"] == 125353
    assert idx["NSR8K"]["counters"]["N_upd"] == 46684748
    assert idx["NSR8K"]["lnz"] + idx["NSR8K"]["unz"] - idx["NSR8K"]["K"] == 3239705


def test_singular_matrix():
    """A structurally singular input must end in SLIP_SINGULAR (-2), slip_get_smallest_pivot.c:93-96."""
    import numpy as np
    # 3x3 with an empty row 2: column 2's pattern has no eligible pivot
    Ap = np.array([0, 2, 4, 5], dtype=np.int64)
    Ai = np.array([0, 1, 0, 1, 0], dtype=np.int32)
    Ax = np.array([2, 3, 5, 7, 11], dtype=np.int64)
    res = oracle_lib.factorize(3, Ap, Ai, np.sign(Ax).astype(np.int32), np.abs(Ax).astype(np.uint64),
                               np.arange(3, dtype=np.int32))
    assert res["status"] == -2 and res["K"] == 2
