"""Host-side hardening of the C ABI, on the CPU emulation build (tests/emu): the library must answer inconsistent device
state, short caller buffers and exhausted memory with an error code -- never with a write outside the caller's arrays or a
launch on freed buffers."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, check_against_golden, load_case

EMU = os.path.join(ROOT, "tests", "emu", "libslip_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu.so"])
    return EMU


def _factor(emu_lib, name="gen_n40", **kw):
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                         tol=entry["tol"], limb_cap=entry["cap"], lib_path=emu_lib, **kw)
    return entry, fix, f


def test_download_rejects_inconsistent_entry_records(emu_lib):
    """an entry record that disagrees with the device's limb counter (what a protocol race left behind in round 2) is a
    DEVICE_ERROR before anything is copied into the caller's arrays"""
    import slip_lu_amd as sl
    entry, fix, f = _factor(emu_lib, waves=2, workers=2)
    try:
        f.run(0)
        good = f.download()
        lib = f.lib
        lib.slip_emu_corrupt_entry.argtypes = [C.c_void_p, C.c_int, C.c_longlong, C.c_int]
        t = int(good["Lp"][3])
        keep = int(good["Llen"][t])
        assert lib.slip_emu_corrupt_entry(f.h, 0, t, 2 * abs(keep) * 2 + 40) == 0       # far longer than it is
        with pytest.raises(sl.SlipError) as ei:
            f.download()
        assert ei.value.code == -100
        assert lib.slip_emu_corrupt_entry(f.h, 0, t, 1 << 30) == 0                        # beyond the slab
        with pytest.raises(sl.SlipError) as ei:
            f.download()
        assert ei.value.code == -100
        # put it back (digits: the record counts 32-bit digits): the handle serves the factors again
        digits = 2 * abs(keep) - (1 if abs(keep) and int(good["Llimbs"][int(np.abs(good["Llen"][:t]).sum()) + abs(keep) - 1]) >> 32 == 0 else 0)
        assert lib.slip_emu_corrupt_entry(f.h, 0, t, digits if keep > 0 else -digits) == 0
        again = f.download()
        assert np.array_equal(again["Llimbs"], good["Llimbs"]) and np.array_equal(again["Llen"], good["Llen"])
    finally:
        f.close()


def test_download_rejects_short_limb_array(emu_lib):
    """the capacity the caller states for a limb array is honoured: one limb short is INCORRECT_INPUT, nothing written"""
    import slip_lu_amd as sl
    entry, fix, f = _factor(emu_lib, waves=1, workers=3)
    try:
        f.run(0)
        i = f.info()
        with pytest.raises(sl.SlipError) as ei:
            f.download(limb_capacity=i["l_limbs"] - 1)
        assert ei.value.code == -3
        check_against_golden(entry, fix, f.download())
        # a limb array without a capacity is refused outright
        buf = np.zeros(max(i["l_limbs"], 1), np.uint64)
        rc = f.lib.slip_hip_factor_download(f.h, None, None, None, buf.ctypes.data, None, None, None, None, None, None, None, None, None, None)
        assert rc == -3
    finally:
        f.close()


def test_rescale_wants_one_scale_per_column(emu_lib):
    import slip_lu_amd as sl
    entry, fix, f = _factor(emu_lib, waves=2, workers=2)
    try:
        f.run(0)
        with pytest.raises(ValueError):
            f.rescale([3] * (entry["n"] - 1))
        lens = np.ones(entry["n"] - 1, np.int32); limbs = np.full(entry["n"] - 1, 3, np.uint64)
        assert f.lib.slip_hip_factor_rescale(f.h, entry["n"] - 1, lens.ctypes.data, limbs.ctypes.data, None) == -3
        f.rescale([1] * entry["n"])
        check_against_golden(entry, fix, f.download())
    finally:
        f.close()


def test_grow_x_reapplies_the_memory_budget(emu_lib):
    """a GROW_X relaunch under a tight budget: the worker count is recomputed for the wider stride (it used to stay, and
    the O(workers * n) workspace doubled until the allocation failed), and the factors still equal the reference's"""
    import slip_lu_amd as sl
    lib = C.CDLL(emu_lib)
    lib.slip_emu_set_budget.argtypes = [C.c_longlong]
    entry, fix = load_case("gen_n300")
    n = entry["n"]
    try:
        # room for six workers at the initial stride (16 digits), so fewer at the strides the run grows to
        lib.slip_emu_set_budget(6 * (n * (16 + 4 * 16 + 16) + 4096))
        f = sl.Factorization(n, fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                             waves=1, lib_path=emu_lib)
        try:
            w0 = f.info()["workers"]
            assert w0 <= 6
            f.run(60)
            i = f.info()
            assert i["launches"] > 1 and i["xcap_digits"] > 16, i          # the stride grew
            assert i["workers"] < w0, (w0, i["workers"])                     # ... and the budget was applied again
            ref = sl.factorize(n, fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                               kmax=60, waves=1, workers=2, lib_path=emu_lib)
            got = f.download()
            for k in ("pinv", "Lp", "Li", "Llen", "Llimbs", "Up", "Ui", "Ulen", "Ulimbs", "rholen", "rholimbs"):
                assert np.array_equal(got[k], ref[k]), k
        finally:
            f.close()
    finally:
        lib.slip_emu_set_budget(0)


def test_create_rejects_dimensions_beyond_the_protocol_fields(emu_lib):
    """column numbers travel in 24-bit fields of the commit protocol: n >= 2^24 - 1 is refused at create"""
    import slip_lu_amd as sl
    lib = sl._lib.load(emu_lib)
    h = C.c_void_p()
    one = np.zeros(2, np.int64)
    rc = lib.slip_hip_factor_create(C.byref(h), (1 << 24) - 1, one.ctypes.data, one.ctypes.data, one.ctypes.data, one.ctypes.data, one.ctypes.data, None)
    assert rc == -3 and not h.value


@pytest.mark.parametrize("name,K", [("gen_n40", 15), ("test_mat", 10), ("test_mat", 0), ("test_mat_p4tol", 4), ("10teams", 60)])
def test_continue_from_a_given_prefix(emu_lib, name, K):
    """slip_hip_factor_set_prefix: the first K columns are given (here: taken from a run that stopped at K), the handle goes
    on from column K and ends with the reference's factors -- SLIP_LU_factorize.c:190-264 entered at k = K"""
    import slip_lu_amd as sl
    entry, fix = load_case(name)
    kw = dict(pivot=entry["pivot"], tol=entry["tol"], limb_cap=entry["cap"], waves=2, workers=4, lib_path=emu_lib)
    a = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], **kw)
    b = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], **kw)
    try:
        if K:
            a.run(K)
            d = a.download()
        else:
            d = dict(Lp=[0], Li=[], Llen=[], Llimbs=[], Up=[0], Ui=[], Ulen=[], Ulimbs=[], pinv=np.arange(entry["n"]))
        piv_row = np.argsort(d["pinv"])[:K]
        b.run(3)                                   # whatever the handle held before is dropped
        b.set_prefix(K, d, piv_row)
        assert b.info()["K"] == K
        b.run(entry["kmax"])
        res = b.download(); res.update(b.info())
        check_against_golden(entry, fix, res, counters=False)
        if K >= 2:
            # inconsistent input is refused: a pivot row named twice; a pivot row that is not in its column
            bad = piv_row.copy(); bad[1] = bad[0]
            with pytest.raises(sl.SlipError) as ei:
                b.set_prefix(K, d, bad)
            assert ei.value.code == -3
            bad = piv_row.copy(); bad[0] = next(r for r in range(entry["n"]) if r not in set(int(v) for v in d["Li"][d["Lp"][0]:d["Lp"][1]]) | {int(piv_row[0])}) if d["Lp"][1] < entry["n"] else bad[0]
            if bad[0] != piv_row[0]:
                with pytest.raises(sl.SlipError):
                    b.set_prefix(K, d, bad)
            # and the handle still works afterwards
            b.set_prefix(K, d, piv_row)
            b.run(entry["kmax"])
            check_against_golden(entry, fix, dict(b.download(), **b.info()), counters=False)
    finally:
        a.close(); b.close()
