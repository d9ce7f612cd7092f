"""REF triangular solves on the GPU (slip_hip_factor_solve; SLIP_LU_solve.c:41-86) -- bit-exact against
the CPU restatement orc_solve, against the reference's own rational solutions (tests/golden/solve_*),
and through size-independent properties (A x = det b, linearity)."""
import json
import os

import numpy as np
import pytest

import oracle_lib
from conftest import GOLDEN, check_solve

pytestmark = pytest.mark.gpu

CASES = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "solve_index.json")))}


@pytest.mark.parametrize("name,nrhs,kw", [
    ("solve_test_mat", 1, {}), ("solve_gen_n40", 3, {}), ("solve_10teams", 2, {}),
    ("solve_gen_n40", 2, dict(workers=1)),                      # one worker takes the right-hand sides in turn
    ("solve_10teams", 16, {}),                                  # one worker per right-hand side
    ("solve_10teams", 5, dict(workers=2, waves=8)),
    ("solve_gen_n40", 1, dict(waves=1)), ("solve_10teams", 1, dict(waves=4, workers=3))])
def test_gpu_solve_matches_reference(name, nrhs, kw):
    check_solve(CASES[name], nrhs=nrhs, **kw)


def _slab(v):
    """python ints -> (signed limb counts, limbs)"""
    lens, limbs = [], []
    for x in v:
        a, l = abs(int(x)), 0
        while a:
            limbs.append(a & (2 ** 64 - 1)); a >>= 64; l += 1
        lens.append(-l if x < 0 else l)
    return np.array(lens, np.int32), np.array(limbs if limbs else [0], np.uint64)


def _dense_cols(n, Ap, Ai, Ax):
    return [[(int(Ai[p]), int(Ax[p])) for p in range(Ap[j], Ap[j + 1])] for j in range(n)]


@pytest.mark.parametrize("n,density,bits,seed,with_oracle", [(300, 0.02, 20, 11, True), (500, 0.008, 40, 12, False)])
def test_gpu_solve_residual_and_linearity(n, density, bits, seed, with_oracle):
    """A x = det b exactly for multi-limb right-hand sides (checked with python integers, no oracle needed),
    numerators linear in b, and equal to orc_solve on the first column (where the scalar oracle finishes
    in seconds: the second case reaches ~300 limbs, minutes of schoolbook arithmetic on the CPU)."""
    import slip_lu_amd as sl
    Ap, Ai, Ax = oracle_lib.matgen(n, density, bits, seed)
    Alen, Alimbs = sl.ints_to_slab(Ax)
    q = np.arange(n, dtype=np.int32)
    rng = np.random.default_rng(seed)
    b1 = [int(v) for v in rng.integers(-10 ** 6, 10 ** 6, n)]
    b2 = [int(rng.integers(-2 ** 62, 2 ** 62)) * int(rng.integers(1, 2 ** 62)) * (i % 3 != 0) for i in range(n)]   # 2 limbs, 1/3 zeros
    b3 = [x + y for x, y in zip(b1, b2)]
    blen, blimbs = _slab(b1 + b2 + b3)
    f = sl.Factorization(n, Ap, Ai, Alen, Alimbs, q)
    try:
        f.run(0)
        fac = f.download()
        xlen, xlimbs = f.solve(blen, blimbs, nrhs=3)
    finally:
        f.close()
    x = oracle_lib.bigints(xlen, xlimbs)
    det = oracle_lib.bigints(fac["rholen"], fac["rholimbs"])[-1]
    x1, x2, x3 = x[:n], x[n:2 * n], x[2 * n:]
    assert [a + b for a, b in zip(x1, x2)] == x3
    cols = _dense_cols(n, Ap, Ai, Ax)
    for xs, bs in ((x1, b1), (x2, b2)):
        r = [0] * n
        for p in range(n):                       # xs[p] belongs to column q[p] = p
            for i, a in cols[int(q[p])]:
                r[i] += a * xs[p]
        assert r == [det * v for v in bs]
    if not with_oracle:
        return
    want, _ = oracle_lib.factorize_and_solve(n, Ap, Ai, Alen, Alimbs, q, np.array(b1, dtype=np.int64))
    assert x1 == want


def test_gpu_solve_on_uploaded_factors():
    """slip_hip_factor_from_factors: the factors go to the host and back (what the SLIP_LU_solve drop-in does) and the
    solve on the uploaded copy equals the solve on the resident factorisation; such a handle refuses run/reset."""
    import slip_lu_amd as sl
    from conftest import solve_inputs
    n, Ap, Ai, Alen, Alimbs, q, _ = solve_inputs(CASES["solve_10teams"])
    b = oracle_lib.solve_rhs(n)
    blen, blimbs = np.sign(b).astype(np.int32), np.abs(b[b != 0]).astype(np.uint64)
    f = sl.Factorization(n, Ap, Ai, Alen, Alimbs, q)
    try:
        f.run(0)
        fac = f.download()
        x1 = f.solve(blen, blimbs)
    finally:
        f.close()
    g = sl.Factorization.from_factors(fac)
    try:
        x2 = g.solve(blen, blimbs)
        assert g.run(0, check=False) == -3                     # SLIP_INCORRECT_INPUT: no A behind this handle
        with pytest.raises(sl.api.SlipError):
            g.reset()
    finally:
        g.close()
    assert np.array_equal(x1[0], x2[0]) and np.array_equal(x1[1], x2[1])
    bad = dict(fac); bad["pinv"] = np.zeros(n, np.int32)
    with pytest.raises(sl.api.SlipError):
        sl.Factorization.from_factors(bad)


def test_gpu_solve_edge_cases():
    """all-zero right-hand sides, a single nonzero, and a factorisation that is not complete"""
    import slip_lu_amd as sl
    from conftest import solve_inputs
    n, Ap, Ai, Alen, Alimbs, q, _ = solve_inputs(CASES["solve_gen_n40"])
    e3 = np.zeros(n, np.int64); e3[3] = -7
    bs = np.concatenate([np.zeros(n, np.int64), e3, np.zeros(n, np.int64)])
    blen = np.sign(bs).astype(np.int32); blimbs = np.abs(bs[bs != 0]).astype(np.uint64)
    f = sl.Factorization(n, Ap, Ai, Alen, Alimbs, q)
    try:
        f.run(n // 2)                                            # columns [0, n/2) only
        with pytest.raises(sl.api.SlipError):
            f.solve(blen, blimbs, nrhs=3)
        f.run(0)
        xlen, xlimbs = f.solve(blen, blimbs, nrhs=3)
    finally:
        f.close()
    x = oracle_lib.bigints(xlen, xlimbs)
    assert x[:n] == [0] * n and x[2 * n:] == [0] * n
    want, _ = oracle_lib.factorize_and_solve(n, Ap, Ai, Alen, Alimbs, q, e3)
    assert x[n:2 * n] == want
