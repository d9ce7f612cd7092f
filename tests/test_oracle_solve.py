"""The CPU restatement of the REF triangular solves (orc_solve; SLIP_LU_solve.c:41-86) against the reference's
own rational solutions (tests/golden/solve_*.slab.gz, produced by oracle/_ref/ref_driver mode `solve`).
Pins the oracle of the next row of the scope table before the device path is built."""
import json
import os
from fractions import Fraction

import numpy as np
import pytest

import oracle_lib
import slabfile
from conftest import GOLDEN

CASES = json.load(open(os.path.join(GOLDEN, "solve_index.json")))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_solve_matches_reference(case):
    fix = slabfile.load(os.path.join(GOLDEN, case["name"] + ".slab.gz"))
    n = case["n"]
    if case["input"].startswith("gen:"):
        a, d, b, seed = case["input"][4:].split(",")
        Ap, Ai, Ax = oracle_lib.matgen(int(a), float(d), int(b), int(seed))
        Alen, Alimbs = np.sign(Ax).astype(np.int32), np.abs(Ax).astype(np.uint64)
    else:
        Ap, Ai, Alen, Alimbs = fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"]
    nums, det = oracle_lib.factorize_and_solve(n, Ap, Ai, Alen, Alimbs, fix["q"], oracle_lib.solve_rhs(n))
    ref_num = oracle_lib.bigints(fix["xnumlen"], fix["xnumlimbs"])
    ref_den = oracle_lib.bigints(fix["xdenlen"], fix["xdenlimbs"])
    for i in range(n):
        assert Fraction(nums[i], det) == Fraction(ref_num[i], ref_den[i]), i
