"""Every path of the column pipeline is HIT on hardware with the reference's digest (the reference's own
discipline is "statements not yet tested: 0", Tcov/cov_test.c:244-277).  slip_hip_info's path counters say which path a
run took; each case below requires its counter to be nonzero AND the factors to be the reference's, bit for bit."""
import threading

import numpy as np
import pytest

from conftest import check_against_golden, load_case

pytestmark = pytest.mark.gpu


def _factor(name, **kw):
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                         tol=entry["tol"], limb_cap=entry["cap"], **kw)
    return entry, fix, f


def _run_and_check(name, **kw):
    entry, fix, f = _factor(name, **kw)
    try:
        rc = f.run(entry["kmax"], check=False)
        res = f.download()
        res["status"] = rc
    finally:
        f.close()
    assert res["status"] == entry["status"]
    check_against_golden(entry, fix, res)
    return res["info"]


def test_gpu_helpers_run_on_another_workers_rows():
    """the farm: an update queue opened to waiting workers, items run by helpers on the owner's private rows (the L2
    write-back / invalidate hand-off of ref_lu_pipe.h)"""
    i = _run_and_check("C4_n100k_c64")
    assert i["farm_jobs"] > 0 and i["farm_items"] > 0, i


def test_gpu_short_chain_and_committer_commit():
    i = _run_and_check("C4_n100k_c64")
    assert i["short_commits"] > 0 and i["committer_commits"] > 0, i


def test_gpu_package_retraction_and_reexport():
    """a source that arrives after the export takes the package back; the column exports again later"""
    i = _run_and_check("prob159")
    assert i["retractions"] > 0 and i["reexports"] > 0, i


@pytest.mark.parametrize("name", ["NSR8K_w600", "10teams", "gen_n2000_pm1"])
def test_gpu_chain_engine_commits_full_packages(name):
    """short one-limb columns travel as FULL packages; the committer's chain engine applies the late sources in LDS, commits,
    hands the rows back (ref_lu_pipe_commit.h)"""
    i = _run_and_check(name)
    assert i["engine_commits"] > 0 and i["engine_sources"] > 0, i


def test_gpu_engine_off_equals_engine_on():
    """diagnostic flag 8 keeps every column on the candidates-only / worker paths: same bytes"""
    for flags in (8, 2, 1):
        i = _run_and_check("NSR8K_w600", debug_flags=flags)
        assert i["engine_commits"] == 0


def test_gpu_complete_path_beyond_the_lds_pattern():
    """patterns of more than 1024 rows (the late columns of NSR8K) take the complete path: row lists, positions and the
    sorted pattern in HBM"""
    entry, fix, f = _factor("NSR8K")
    try:
        f.run(0)
        res = f.download()
    finally:
        f.close()
    check_against_golden(entry, fix, res)
    widest = int((np.diff(res["Lp"]) + np.diff(res["Up"]) - 1).max())      # rows of the pattern: U part + L part (the pivot is in both)
    assert widest > 1024, widest


def test_gpu_stop_grow_relaunch():
    """a small capacity hint makes the kernel stop at a full slab, the host grow it and relaunch from the frontier"""
    i = _run_and_check("rl5934", lnz_hint=4000, unz_hint=4000)
    assert i["launches"] > 1, i


def test_gpu_two_handles_run_concurrently():
    """two factorisations on two streams at the same time (each launch has its own committer; a launch whose committer is
    not resident yet falls back to worker commits -- nothing waits for a workgroup that has not started)"""
    import ctypes as C
    import slip_lu_amd as sl
    import slabfile
    cases = [_factor("rl5934", workers=96), _factor("fome12", workers=96)]
    # two HIP streams from the runtime the library itself is linked against (no second runtime in this process)
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    streams = []
    for _ in range(2):
        h = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(h)) == 0 and h.value
        streams.append(h.value)
    errs = []

    def go(idx):
        try:
            entry, fix, f = cases[idx]
            for rep in range(3):
                f.reset()
                rc = f.run(entry["kmax"], stream=streams[idx], check=False)
                assert rc == 0
                assert slabfile.factor_digest(f.download()) == entry["digest"], (idx, rep)
        except Exception as e:                       # noqa: BLE001
            errs.append((idx, repr(e)))
    ts = [threading.Thread(target=go, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for _, _, f in cases:
        f.close()
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    for h in streams:
        hip.hipStreamDestroy(h)
    assert not errs, errs


@pytest.mark.parametrize("name,K", [("gen_n40", 15), ("10teams", 60), ("prob159", 400), ("NSR8K_w600", 300), ("gen_n2000_pm1", 1200)])
def test_gpu_continue_from_a_given_prefix(name, K):
    """slip_hip_factor_set_prefix on the device: K columns given, the rest factorised on top of them = the reference's factors"""
    import slip_lu_amd as sl
    entry, fix, a = _factor(name)
    _, _, b = _factor(name)
    try:
        a.run(K)
        d = a.download()
        assert d["K"] == K
        b.set_prefix(K, d, np.argsort(d["pinv"])[:K])
        assert b.info()["K"] == K
        rc = b.run(entry["kmax"], check=False)
        res = b.download(); res.update(b.info()); res["status"] = rc
        assert rc == entry["status"]
        check_against_golden(entry, fix, res, counters=False)
    finally:
        a.close(); b.close()


def test_gpu_farm_complete():
    """SURVEY 8(e) end to end on the device: the blocks' columns (factorised, rescaled, gathered -- world 1 here) become the
    prefix of the whole matrix's handle, which factorises the separator columns: equal to the whole-matrix oracle"""
    import test_subtree_farm as T
    from slip_lu_amd import parallel
    n, Ap, Ai, Ax, q, n0 = T.make_bordered((9, 14, 11, 7, 12), 6, 5, bits=12)
    whole = T.oracle_factor(n, Ap, Ai, Ax, q)
    got = parallel.farm_complete(None, n, Ap, Ai, Ax, q, n0)
    assert got["farm_prefix"] == n0 and got["K"] == n
    T.same_factors(got, whole)


def test_gpu_handles_reuse_pooled_buffers():
    """create -> run -> destroy cycles (what the drop-in SLIP_LU_factorize does per call) on buffers taken from the process-level
    pool: every cycle ends with the reference's factors (nothing relies on fresh device memory), and the pool can be given back"""
    from slip_lu_amd import _lib
    for name in ("gen_n2000_pm1", "10teams", "gen_n2000_pm1", "prob159", "10teams"):
        _run_and_check(name)
    _lib.load().slip_hip_pool_release()
    _run_and_check("10teams")
