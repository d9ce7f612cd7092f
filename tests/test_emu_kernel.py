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


@pytest.fixture(scope="module")
def emu_farm_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu_farm.so"])
    return os.path.join(ROOT, "tests", "emu", "libslip_emu_farm.so")


def set_seed(emu_lib, seed):
    """the interleaving of the emulated workgroups is a seeded pseudo-random schedule (tests/emu/fiber_emu.h)"""
    import ctypes
    lib = ctypes.CDLL(emu_lib)
    lib.slip_emu_set_seed.argtypes = [ctypes.c_ulonglong]
    lib.slip_emu_set_seed(seed)


@pytest.mark.parametrize("name,waves,workers", [("test_mat", 2, 1), ("test_mat", 1, 3), ("test_mat_p1", 1, 4), ("test_mat_p2", 4, 2),
                                                ("test_mat_p4tol", 2, 3), ("test_mat_p5", 2, 5), ("test_mat_tol01", 2, 2),
                                                ("test_mat_noord", 16, 1), ("test_mat_amd", 1, 10), ("gen_n40", 1, 6)])
def test_emulated_kernel_matches_reference(emu_lib, name, waves, workers):
    """the column-worker pipeline on `workers` concurrently emulated workgroups, several interleavings"""
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    for seed in (1, 2, 3):
        set_seed(emu_lib, seed)
        res = sl.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"],
                           pivot=entry["pivot"], tol=entry["tol"], kmax=entry["kmax"], limb_cap=entry["cap"],
                           waves=waves, workers=workers, lib_path=emu_lib)
        check_against_golden(entry, fix, res)


@pytest.mark.parametrize("name,waves,workers", [("test_mat", 2, 3), ("gen_n40", 2, 6), ("test_mat_p4tol", 1, 5)])
def test_emulated_workers_help_with_update_queues(emu_farm_lib, name, waves, workers):
    """a build in which every update queue of two or more items is opened to the waiting workers: items of one worker's
    column run on other workgroups, on the owner's private rows (the farm protocol of ref_lu_pipe.h)"""
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    for seed in (1, 2, 3):
        set_seed(emu_farm_lib, seed)
        res = sl.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"],
                           pivot=entry["pivot"], tol=entry["tol"], kmax=entry["kmax"], limb_cap=entry["cap"],
                           waves=waves, workers=workers, lib_path=emu_farm_lib)
        check_against_golden(entry, fix, res)


def test_emulated_kernel_under_sanitizers():
    """the kernel source (committer, packages, helpers included) under AddressSanitizer + UBSan on the CPU: a subprocess with
    libasan preloaded runs small cases against the goldens; any report aborts it (GPU sanitizers are not available)"""
    import sys
    emu_dir = os.path.join(ROOT, "tests", "emu")
    subprocess.check_call(["make", "-s", "-C", emu_dir, "-f", "sanitize.mk", "libslip_emu_san.so"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from conftest import load_case, check_against_golden\n"
        "import slip_lu_amd as sl\n"
        "for name, w, wv in [('test_mat', 3, 2), ('test_mat_p5', 4, 2), ('test_mat_p4tol', 5, 1)]:\n"
        "    entry, fix = load_case(name)\n"
        "    res = sl.factorize(entry['n'], fix['Ap'], fix['Ai'], fix['Alen'], fix['Alimbs'], fix['q'], pivot=entry['pivot'], tol=entry['tol'],\n"
        "                       kmax=entry['kmax'], limb_cap=entry['cap'], waves=wv, workers=w, lib_path=%r)\n"
        "    check_against_golden(entry, fix, res)\n"
        "print('sanitized run ok')\n") % (ROOT, os.path.join(ROOT, "tests"), os.path.join(emu_dir, "libslip_emu_san.so"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "sanitized run ok" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr, out.stderr[-4000:]


@pytest.mark.parametrize("name,waves,workers,nrhs", [("solve_test_mat", 2, 1, 1), ("solve_gen_n40", 2, 2, 2)])
def test_emulated_solve_matches_reference(emu_lib, name, waves, workers, nrhs):
    """forward / back substitution of the kernel source (slip_solve_rhs) against orc_solve and the reference's x;
    the right-hand sides are spread over the workers"""
    import json
    from conftest import GOLDEN, check_solve
    case = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "solve_index.json")))}[name]
    check_solve(case, lib_path=emu_lib, nrhs=nrhs, waves=waves, workers=workers)


@pytest.mark.parametrize("name,waves,workers,nrhs", [("solve_gen_n40", 2, 5, 1), ("solve_test_mat", 1, 4, 2)])
def test_emulated_solve_with_helpers(emu_farm_lib, name, waves, workers, nrhs):
    """fewer right-hand sides than workgroups: the others help with the update queues of the substitutions (forward: kind 1,
    backward: kind 5; the build that opens every queue of two or more items), same numerators"""
    import json
    from conftest import GOLDEN, check_solve
    case = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "solve_index.json")))}[name]
    check_solve(case, lib_path=emu_farm_lib, nrhs=nrhs, waves=waves, workers=workers)


def test_emulated_solve_zero_and_unit_rhs(emu_lib):
    import json
    import numpy as np
    import oracle_lib
    import slip_lu_amd as sl
    from conftest import GOLDEN, solve_inputs
    case = {c["name"]: c for c in json.load(open(os.path.join(GOLDEN, "solve_index.json")))}["solve_test_mat"]
    n, Ap, Ai, Alen, Alimbs, q, _ = solve_inputs(case)
    e = np.zeros(n, np.int64); e[2] = 5
    bs = np.concatenate([np.zeros(n, np.int64), e])
    f = sl.Factorization(n, Ap, Ai, Alen, Alimbs, q, waves=2, lib_path=emu_lib)
    try:
        f.run(0)
        xlen, xlimbs = f.solve(np.sign(bs).astype(np.int32), np.abs(bs[bs != 0]).astype(np.uint64), nrhs=2)
        fac = f.download()
    finally:
        f.close()
    x = oracle_lib.bigints(xlen, xlimbs)
    want, _ = oracle_lib.factorize_and_solve(n, Ap, Ai, Alen, Alimbs, q, e)
    assert x[:n] == [0] * n and x[n:] == want
    # the same through a handle built from the downloaded factors (what the SLIP_LU_solve drop-in does)
    g = sl.Factorization.from_factors(fac, waves=2, lib_path=emu_lib)
    try:
        x2 = g.solve(np.sign(bs).astype(np.int32), np.abs(bs[bs != 0]).astype(np.uint64), nrhs=2)
    finally:
        g.close()
    assert oracle_lib.bigints(*x2) == x


def check_rescale(lib_path, name, seed, **kw):
    """slip_hip_factor_rescale (subtree farm): every stored value times its column's / its row's pivot
    position's scale, against python integers"""
    import numpy as np
    import oracle_lib
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    n = entry["n"]
    f = sl.Factorization(n, fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], lib_path=lib_path, **kw)
    try:
        f.run(0)
        before = f.download()
        rng = np.random.default_rng(seed)
        scales = [int(rng.integers(1, 2 ** 62)) ** int(rng.integers(1, 5)) * (1 if rng.random() < 0.5 else -1) for _ in range(n)]
        f.rescale(scales)
        after = f.download()
    finally:
        f.close()
    pinv = before["pinv"]
    L0 = oracle_lib.bigints(before["Llen"], before["Llimbs"]); L1 = oracle_lib.bigints(after["Llen"], after["Llimbs"])
    U0 = oracle_lib.bigints(before["Ulen"], before["Ulimbs"]); U1 = oracle_lib.bigints(after["Ulen"], after["Ulimbs"])
    r0 = oracle_lib.bigints(before["rholen"], before["rholimbs"]); r1 = oracle_lib.bigints(after["rholen"], after["rholimbs"])
    assert np.array_equal(before["Li"], after["Li"]) and np.array_equal(before["Ui"], after["Ui"])
    for k in range(n):
        for p in range(int(before["Lp"][k]), int(before["Lp"][k + 1])):
            assert L1[p] == L0[p] * scales[k], ("L", k, p)
        for p in range(int(before["Up"][k]), int(before["Up"][k + 1])):
            assert U1[p] == U0[p] * scales[int(pinv[int(before["Ui"][p])])], ("U", k, p)
        assert r1[k] == r0[k] * scales[k], ("rho", k)


def test_emulated_rescale(emu_lib):
    check_rescale(emu_lib, "gen_n40", 5, waves=2, workers=2)


@pytest.mark.parametrize("name,waves,workers,seed", [("10teams", 2, 6, 1), ("NSR8K_w600", 2, 6, 3),
                                                     ("gen_n40_pm1", 2, 5, 1), ("gen_n40", 2, 6, 2)])
def test_emulated_chain_engine(emu_lib, name, waves, workers, seed):
    """FULL packages through the committer's chain engine (late sources applied in LDS, rows handed back) on the CPU
    emulation: the reference's factors and algorithmic counters, and the engine really ran"""
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    set_seed(emu_lib, seed)
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                         limb_cap=entry["cap"], waves=waves, workers=workers, lib_path=emu_lib)
    try:
        f.run(entry["kmax"])
        res = f.download()
    finally:
        f.close()
    check_against_golden(entry, fix, res)
    assert res["info"]["engine_commits"] > 0, res["info"]
    if name != "gen_n40":
        assert res["info"]["engine_sources"] > 0, res["info"]


def test_emulated_engine_off_and_candidates_only(emu_lib):
    """diagnostic flag 8: no full packages -- the candidates-only packages and the worker commits alone"""
    import slip_lu_amd as sl
    for name, flags in (("gen_n40", 8), ("gen_n40_pm1", 8), ("test_mat_p4tol", 8), ("gen_n40", 2)):
        entry, fix = load_case(name)
        res = sl.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                           kmax=entry["kmax"], limb_cap=entry["cap"], waves=2, workers=5, lib_path=emu_lib, debug_flags=flags)
        check_against_golden(entry, fix, res)
        assert res["info"]["engine_commits"] == 0


def test_emulated_hensel_division(emu_lib):
    """wr_div_hensel (the exact division that needs no 2-adic inverse; wave_bigint_reg.h) on the CPU emulation of the wave
    primitives, against Python integers: the same cases the device test runs (tests/test_gpu_parity.py)"""
    import ctypes as C
    import test_gpu_parity
    test_gpu_parity._hensel_cases(C.CDLL(emu_lib))


def _weak_run(lib_path, name, seed, workers, waves):
    """one emulated factorisation in the emulator's WEAK-STORE mode (tests/emu/fiber_emu.h: sc1 stores land late and out of
    order, nobody but the issuing wave sees them before): True if the factors are the reference's"""
    import ctypes
    import slip_lu_amd as sl
    lib = ctypes.CDLL(lib_path)
    lib.slip_emu_set_seed.argtypes = [ctypes.c_ulonglong]
    lib.slip_emu_set_weak(1)
    lib.slip_emu_set_seed(seed)
    entry, fix = load_case(name)
    try:
        res = sl.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                           kmax=entry["kmax"], limb_cap=entry["cap"], waves=waves, workers=workers, lib_path=lib_path, check=False)
        check_against_golden(entry, fix, res)
        return True
    except (AssertionError, sl.SlipError):
        return False
    finally:
        lib.slip_emu_set_weak(0)


@pytest.mark.parametrize("name,workers,waves", [("gen_n40", 5, 2), ("gen_n40_pm1", 24, 2), ("test_mat_p4tol", 9, 1), ("10teams", 9, 2)])
def test_emulated_pipeline_in_weak_store_mode(emu_lib, emu_farm_lib, name, workers, waves):
    """the pipeline's hand-offs under delayed, reordered write-through stores: every word another
    workgroup acts on is behind a drain / release of the data it announces -- committer packages and mailboxes, the
    frontier, Lready, job slots of the helpers (second build)"""
    for seed in ((3, 11) if name != "gen_n40" else (3,)):
        assert _weak_run(emu_lib, name, seed, workers, waves), (name, seed)
    if name != "10teams":
        assert _weak_run(emu_farm_lib, name, 5, workers, waves), name


def test_weak_store_mode_finds_the_permutation_race():
    """the race of round 3, put back by a test-only build switch (SLIP_EMU_BUG_PERM: the permutation swaps of a batch stored by
    the publishing waves in parallel -- successive columns write the same words of row_perm / pinv): the weak-store mode must
    find it (on hardware it showed as INTERNAL site 102 on 10teams once in a few runs), and the real build must pass the same
    seeds.  The two round-2 races are outside the model: the verdict word in a reused package slot is a protocol error in
    program order (the sequentially consistent mode's seeded schedules cover it, test_gpu_verdicts_survive_slot_reuse on
    hardware), and 'every wave searching while wave 0 rewrites the row' is an LDS race inside one workgroup -- LDS is plain
    memory here and the lanes of a workgroup run in one fixed interleaving."""
    bug = os.path.join(ROOT, "tests", "emu", "libslip_emu_bugperm.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu_bugperm.so", "libslip_emu.so"])
    seeds = (12, 19)
    found = [s for s in seeds if not _weak_run(bug, "10teams", s, 9, 2)]
    assert found, "the weak-store mode did not find the race that was put back"
    for s in seeds:
        assert _weak_run(EMU, "10teams", s, 9, 2), s
