"""Python face of the C ABI in include/slip_hip.h (tests, bench, smoke).

Mirrors the reference's expert sequence for the hot path
(SLIP_LU/Demo/SLIPLU.c:237-256): the column order q comes from the analysis
step (data to this module), `Factorization.run` is SLIP_LU_factorize.
All arithmetic happens in the HIP library; this file only marshals arrays.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

STATUS = {0: "SLIP_OK", -1: "SLIP_OUT_OF_MEMORY", -2: "SLIP_SINGULAR",
          -3: "SLIP_INCORRECT_INPUT", -100: "DEVICE_ERROR"}


class SlipError(RuntimeError):
    def __init__(self, code, where):
        super().__init__(f"{where}: {STATUS.get(code, code)}")
        self.code = code


def ints_to_slab(values):
    """int64 numpy array -> (signed limb counts, limbs): |v| < 2^63."""
    v = np.asarray(values, dtype=np.int64)
    lens = np.sign(v).astype(np.int32)
    limbs = np.abs(v[v != 0]).astype(np.uint64)
    return lens, limbs


def matgen(n, density, bits, seed, lib_path=None):
    """The benchmark's synthetic CSC (slip_matgen.h) -> Ap, Ai, Ax (int64 values)."""
    lib = _lib.load(lib_path)
    pAp, pAi, pAx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.slip_hip_matgen(n, density, bits, seed, C.byref(pAp), C.byref(pAi), C.byref(pAx))
    if rc:
        raise SlipError(rc, "slip_hip_matgen")
    Ap = np.ctypeslib.as_array(C.cast(pAp, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    nnz = int(Ap[n])
    Ai = np.ctypeslib.as_array(C.cast(pAi, C.POINTER(C.c_int32)), shape=(nnz,)).copy()
    Ax = np.ctypeslib.as_array(C.cast(pAx, C.POINTER(C.c_int64)), shape=(nnz,)).copy()
    for p in (pAp, pAi, pAx):
        lib.slip_hip_free(p)
    return Ap, Ai, Ax


def read_triplet(path, lib_path=None):
    """A triplet file (SLIP_tripread's format, SLIP_LU/Demo/demos.c:245-331) -> n, Ap, Ai, Alen, Alimbs: the CSC limb slabs
    SLIP_build_sparse_trip_mpz would hold (slip_trip_to_mat.c:23-69), ready for Factorization()."""
    lib = _lib.load(lib_path)
    n, nl = C.c_int32(), C.c_int64()
    pAp, pAi, pAlen, pAl = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    rc = lib.slip_hip_read_triplet(os.fsencode(path), C.byref(n), C.byref(pAp), C.byref(pAi), C.byref(pAlen), C.byref(pAl), C.byref(nl))
    if rc:
        raise SlipError(rc, "slip_hip_read_triplet")
    n = n.value
    Ap = np.ctypeslib.as_array(C.cast(pAp, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    nnz = int(Ap[n])
    Ai = np.ctypeslib.as_array(C.cast(pAi, C.POINTER(C.c_int32)), shape=(nnz,)).copy()
    Alen = np.ctypeslib.as_array(C.cast(pAlen, C.POINTER(C.c_int32)), shape=(nnz,)).copy()
    Al = np.ctypeslib.as_array(C.cast(pAl, C.POINTER(C.c_uint64)), shape=(max(nl.value, 1),)).copy()[:nl.value]
    for p in (pAp, pAi, pAlen, pAl):
        lib.slip_hip_free(p)
    return n, Ap, Ai, Alen, Al


def write_triplet(path, n, Ap, Ai, Alen, Alimbs, lib_path=None):
    """CSC limb slabs -> a triplet file (1-based, decimal) that SLIP_tripread and read_triplet accept."""
    lib = _lib.load(lib_path)
    Ap = np.ascontiguousarray(Ap, np.int64); Ai = np.ascontiguousarray(Ai, np.int32)
    Alen = np.ascontiguousarray(Alen, np.int32); Al = np.ascontiguousarray(Alimbs, np.uint64)
    if Al.size == 0:
        Al = np.zeros(1, np.uint64)
    rc = lib.slip_hip_write_triplet(os.fsencode(path), int(n), Ap.ctypes.data, Ai.ctypes.data, Alen.ctypes.data, Al.ctypes.data)
    if rc:
        raise SlipError(rc, "slip_hip_write_triplet")


class Factorization:
    """A resident REF LU factorisation on one GPU (handle of slip_hip_factor_*)."""

    def __init__(self, n, Ap, Ai, Alen, Alimbs, q, pivot=3, tol=1.0, limb_cap=0, waves=0,
                 lnz_hint=0, unz_hint=0, workers=0, lib_path=None, debug_flags=0):
        self.lib = _lib.load(lib_path)
        self.n = int(n)
        Ap = np.ascontiguousarray(Ap, dtype=np.int64)
        Ai = np.ascontiguousarray(Ai, dtype=np.int32)
        Alen = np.ascontiguousarray(Alen, dtype=np.int32)
        Alimbs = np.ascontiguousarray(Alimbs, dtype=np.uint64)
        if Alimbs.size == 0:
            Alimbs = np.zeros(1, dtype=np.uint64)
        q = np.ascontiguousarray(q, dtype=np.int32)
        opt = _lib.Options(pivot, tol, limb_cap, waves, lnz_hint, unz_hint, workers, debug_flags)
        self.h = C.c_void_p()
        rc = self.lib.slip_hip_factor_create(C.byref(self.h), self.n, Ap.ctypes.data, Ai.ctypes.data,
                                             Alen.ctypes.data, Alimbs.ctypes.data, q.ctypes.data,
                                             C.byref(opt))
        if rc:
            self.h = None
            raise SlipError(rc, "slip_hip_factor_create")

    @classmethod
    def from_factors(cls, fac, waves=0, workers=0, lib_path=None):
        """A solve-only handle around factors in the form `download()` returns (slip_hip_factor_from_factors)."""
        self = cls.__new__(cls)
        self.lib = _lib.load(lib_path)
        self.n = int(fac["n"])
        arrs = [np.ascontiguousarray(fac[k], dtype=t) for k, t in (
            ("Lp", np.int64), ("Li", np.int32), ("Llen", np.int32), ("Llimbs", np.uint64),
            ("Up", np.int64), ("Ui", np.int32), ("Ulen", np.int32), ("Ulimbs", np.uint64), ("pinv", np.int32))]
        arrs = [a if a.size else np.zeros(1, a.dtype) for a in arrs]
        opt = _lib.Options(3, 1.0, 0, waves, 0, 0, workers, 0)
        self.h = C.c_void_p()
        rc = self.lib.slip_hip_factor_from_factors(C.byref(self.h), self.n, *[a.ctypes.data for a in arrs], C.byref(opt))
        if rc:
            self.h = None
            raise SlipError(rc, "slip_hip_factor_from_factors")
        return self

    def set_prefix(self, K, fac, piv_row):
        """The first K columns are given (slip_hip_factor_set_prefix): `fac` holds Lp/Li/Llen/Llimbs/Up/Ui/Ulen/Ulimbs of
        those columns in the form `download()` returns, piv_row[k] the pivot row of column k; run() continues at column K."""
        K = int(K)
        arrs = [np.ascontiguousarray(fac[k], dtype=t) for k, t in (
            ("Lp", np.int64), ("Li", np.int32), ("Llen", np.int32), ("Llimbs", np.uint64),
            ("Up", np.int64), ("Ui", np.int32), ("Ulen", np.int32), ("Ulimbs", np.uint64))]
        arrs.append(np.ascontiguousarray(piv_row, dtype=np.int32))
        if K > 0 and (arrs[0].size < K + 1 or arrs[4].size < K + 1 or arrs[8].size < K):
            raise SlipError(-3, "set_prefix: arrays shorter than K columns")
        arrs = [a if a.size else np.zeros(1, a.dtype) for a in arrs]
        rc = self.lib.slip_hip_factor_set_prefix(self.h, K, *[a.ctypes.data for a in arrs])
        if rc:
            raise SlipError(rc, "slip_hip_factor_set_prefix")

    def reset(self):
        rc = self.lib.slip_hip_factor_reset(self.h)
        if rc:
            raise SlipError(rc, "slip_hip_factor_reset")

    def run(self, kmax=0, stream=None, check=True):
        rc = self.lib.slip_hip_factor_run(self.h, int(kmax), C.c_void_p(stream or 0))
        if rc and check:
            raise SlipError(rc, "slip_hip_factor_run")
        return rc

    def info(self):
        i = _lib.Info()
        self.lib.slip_hip_factor_info(self.h, C.byref(i))
        return {k: getattr(i, k) for k, _ in _lib.Info._fields_}

    def download(self, limb_capacity=None):
        """Factor arrays in the canonical form the tests compare (original row ids).  limb_capacity: the capacity to CLAIM for
        the L limb array (tests of the capacity check; the array itself is always sized from info())."""
        i = self.info()
        K = i["K"]
        out = dict(n=self.n, K=K, status=i["status"],
                   Lp=np.zeros(K + 1, np.int64), Up=np.zeros(K + 1, np.int64),
                   Li=np.zeros(i["lnz"], np.int32), Ui=np.zeros(i["unz"], np.int32),
                   Llen=np.zeros(i["lnz"], np.int32), Ulen=np.zeros(i["unz"], np.int32),
                   Llimbs=np.zeros(max(i["l_limbs"], 1), np.uint64), Ulimbs=np.zeros(max(i["u_limbs"], 1), np.uint64),
                   rholen=np.zeros(K, np.int32), pinv=np.zeros(self.n, np.int32))
        # pivots are entries of L: an upper bound of their limbs is l_limbs
        rho = np.zeros(max(i["l_limbs"], 1), np.uint64)
        cap = C.c_int64(rho.size)
        lcap, ucap = C.c_int64(limb_capacity if limb_capacity is not None else out["Llimbs"].size), C.c_int64(out["Ulimbs"].size)
        rc = self.lib.slip_hip_factor_download(
            self.h, out["Lp"].ctypes.data, out["Li"].ctypes.data, out["Llen"].ctypes.data, out["Llimbs"].ctypes.data, C.byref(lcap),
            out["Up"].ctypes.data, out["Ui"].ctypes.data, out["Ulen"].ctypes.data, out["Ulimbs"].ctypes.data, C.byref(ucap),
            out["rholen"].ctypes.data, rho.ctypes.data, C.byref(cap), out["pinv"].ctypes.data)
        if rc:
            raise SlipError(rc, "slip_hip_factor_download")
        out["Llimbs"] = out["Llimbs"][:lcap.value]
        out["Ulimbs"] = out["Ulimbs"][:ucap.value]
        out["rholimbs"] = rho[:cap.value].copy()
        out["counters"] = np.array([i["n_upd"], i["b_read"], i["b_write"], i["n_src"], i["l_streamed"],
                                    i["max_limbs"], K, i["limb_macs"]], dtype=np.int64)
        out["info"] = i
        return out

    def solve(self, blen, blimbs, nrhs=1, stream=None):
        """REF forward/back substitution on the resident factors (slip_hip_factor_solve): dense b in
        original row order as a limb slab -> (xlen, xlimbs) numerators over det by pivot position."""
        blen = np.ascontiguousarray(blen, dtype=np.int32)
        blimbs = np.ascontiguousarray(blimbs, dtype=np.uint64)
        if blen.size != self.n * nrhs:
            raise ValueError("blen must hold n*nrhs entries")
        if blimbs.size == 0:
            blimbs = np.zeros(1, dtype=np.uint64)
        pl, px, nl = C.c_void_p(), C.c_void_p(), C.c_int64()
        rc = self.lib.slip_hip_factor_solve(self.h, int(nrhs), blen.ctypes.data, blimbs.ctypes.data,
                                            C.byref(pl), C.byref(px), C.byref(nl), C.c_void_p(stream or 0))
        if rc:
            raise SlipError(rc, "slip_hip_factor_solve")
        xlen = np.ctypeslib.as_array(C.cast(pl, C.POINTER(C.c_int32)), shape=(self.n * nrhs,)).copy()
        xlimbs = (np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_uint64)), shape=(nl.value,)).copy()
                  if nl.value else np.zeros(0, np.uint64))
        self.lib.slip_hip_free(pl)
        self.lib.slip_hip_free(px)
        return xlen, xlimbs

    def pivots(self):
        """the pivot chain rho[0..K) only (signed limb counts, limbs): what the subtree farm exchanges"""
        i = self.info()
        K = i["K"]
        rholen = np.zeros(max(K, 1), np.int32)
        rho = np.zeros(max(i["l_limbs"], 1), np.uint64)
        cap = C.c_int64(rho.size)
        rc = self.lib.slip_hip_factor_download(self.h, None, None, None, None, None, None, None, None, None, None,
                                               rholen.ctypes.data, rho.ctypes.data, C.byref(cap), None)
        if rc:
            raise SlipError(rc, "slip_hip_factor_download")
        return rholen[:K], rho[:cap.value].copy()

    def rescale(self, scales, stream=None):
        """Subtree farm: multiply the committed columns by per-column big-integer scales on the device
        (slip_hip_factor_rescale): L(:,k), rho[k] by scales[k]; U entries by the scale of their row's pivot position."""
        lens, limbs = [], []
        for v in scales:
            a, l = abs(int(v)), 0
            while a:
                limbs.append(a & (2 ** 64 - 1)); a >>= 64; l += 1
            lens.append(-l if v < 0 else l)
        lens = np.array(lens, np.int32); limbs = np.array(limbs if limbs else [0], np.uint64)
        K = self.info()["K"]
        if len(lens) != K:
            raise ValueError(f"rescale needs one scale per committed column: {len(lens)} given, K = {K}")
        rc = self.lib.slip_hip_factor_rescale(self.h, int(len(lens)), lens.ctypes.data, limbs.ctypes.data, C.c_void_p(stream or 0))
        if rc:
            raise SlipError(rc, "slip_hip_factor_rescale")

    def solve_ms(self):
        return self.lib.slip_hip_factor_solve_ms(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.slip_hip_factor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def factorize(n, Ap, Ai, Alen, Alimbs, q, pivot=3, tol=1.0, kmax=0, limb_cap=0, waves=0, check=True,
              lib_path=None, workers=0, lnz_hint=0, unz_hint=0, debug_flags=0):
    """One-shot SLIP_LU_factorize on the GPU; returns the canonical factor dict."""
    f = Factorization(n, Ap, Ai, Alen, Alimbs, q, pivot=pivot, tol=tol, limb_cap=limb_cap, waves=waves,
                      lib_path=lib_path, workers=workers, lnz_hint=lnz_hint, unz_hint=unz_hint, debug_flags=debug_flags)
    try:
        rc = f.run(kmax, check=check)
        out = f.download()
        out["status"] = rc
        return out
    finally:
        f.close()
