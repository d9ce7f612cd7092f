"""ctypes binding of the CPU restatement (oracle/liboracle.so).  Test-side only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "liboracle.so")

NCOUNTERS = 8
COUNTER_NAMES = ("N_upd", "B_read", "B_write", "N_src", "L_streamed", "maxlimbs", "K_done", "limb_macs")


class _Result(C.Structure):
    _fields_ = [("n", C.c_int32), ("K", C.c_int32), ("status", C.c_int32),
                ("lnz", C.c_int64), ("unz", C.c_int64),
                ("Lp", C.POINTER(C.c_int64)), ("Up", C.POINTER(C.c_int64)),
                ("Li", C.POINTER(C.c_int32)), ("Ui", C.POINTER(C.c_int32)),
                ("Llen", C.POINTER(C.c_int32)), ("Ulen", C.POINTER(C.c_int32)),
                ("Lnl", C.c_int64), ("Unl", C.c_int64),
                ("Llimbs", C.POINTER(C.c_uint64)), ("Ulimbs", C.POINTER(C.c_uint64)),
                ("rholen", C.POINTER(C.c_int32)), ("rholimbs", C.POINTER(C.c_uint64)), ("rhonl", C.c_int64),
                ("pinv", C.POINTER(C.c_int32)),
                ("counters", C.c_int64 * NCOUNTERS), ("seconds", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "port"])
        _lib = C.CDLL(_SO)
        _lib.orc_factorize.restype = C.POINTER(_Result)
        _lib.orc_factorize.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_double, C.c_int32, C.c_int32]
        _lib.orc_free.argtypes = [C.POINTER(_Result)]
        _lib.orc_ipge.restype = C.c_int
        _lib.orc_matgen.argtypes = [C.c_int32, C.c_double, C.c_int32, C.c_uint64,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        _lib.orc_free_ptr.argtypes = [C.c_void_p]
    return _lib


def solve_rhs(n):
    """the deterministic right-hand side of the solve goldens (oracle/ref_driver.c, mode solve)"""
    return np.array([((i * 2654435761) % (1 << 32)) % 2001 - 1000 for i in range(n)], dtype=np.int64)


def factorize_and_solve(n, Ap, Ai, Alen, Alimbs, q, b, pivot=3, tol=1.0):
    """orc_factorize + orc_solve: returns (numerators as python ints in permuted order, det as python int)."""
    Ap = np.ascontiguousarray(Ap, dtype=np.int64); Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Alen = np.ascontiguousarray(Alen, dtype=np.int32); Alimbs = np.ascontiguousarray(Alimbs, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.int32)
    L = lib()
    r = L.orc_factorize(n, Ap.ctypes.data, Ai.ctypes.data, Alen.ctypes.data, Alimbs.ctypes.data, q.ctypes.data,
                        pivot, float(tol), 0, 0)
    assert r and r.contents.status == 0 and r.contents.K == n
    b = np.asarray(b, dtype=np.int64)
    blen = np.sign(b).astype(np.int32); blimbs = np.abs(b[b != 0]).astype(np.uint64)
    if blimbs.size == 0:
        blimbs = np.zeros(1, np.uint64)
    xl, xp, nl = C.POINTER(C.c_int32)(), C.POINTER(C.c_uint64)(), C.c_int64()
    L.orc_solve.argtypes = [C.POINTER(_Result), C.c_int32, C.c_void_p, C.c_void_p,
                            C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_int64)]
    rc = L.orc_solve(r, 1, blen.ctypes.data, blimbs.ctypes.data, C.byref(xl), C.byref(xp), C.byref(nl))
    assert rc == 0
    lens = _arr(xl, n, np.int32); limbs = _arr(xp, nl.value, np.uint64)
    R = r.contents
    rl = _arr(R.rholen, n, np.int32); rlimbs = _arr(R.rholimbs, R.rhonl, np.uint64)
    L.orc_free_ptr(C.cast(xl, C.c_void_p)); L.orc_free_ptr(C.cast(xp, C.c_void_p))
    L.orc_free(r)
    return bigints(lens, limbs), bigints(rl, rlimbs)[-1]


def bigints(lens, limbs):
    """(signed limb counts, limbs) -> list of python ints"""
    out, o = [], 0
    for l in lens:
        a = abs(int(l)); v = 0
        for t in range(a):
            v |= int(limbs[o + t]) << (64 * t)
        out.append(-v if l < 0 else v); o += a
    return out


def _arr(ptr, count, dtype):
    if count <= 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)


def factorize(n, Ap, Ai, Alen, Alimbs, q, pivot=3, tol=1.0, kmax=0, cap=0):
    """Run the CPU restatement; returns a dict with the canonical factor arrays."""
    Ap = np.ascontiguousarray(Ap, dtype=np.int64)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    Alen = np.ascontiguousarray(Alen, dtype=np.int32)
    Alimbs = np.ascontiguousarray(Alimbs, dtype=np.uint64)
    if Alimbs.size == 0:
        Alimbs = np.zeros(1, dtype=np.uint64)
    q = np.ascontiguousarray(q, dtype=np.int32)
    r = lib().orc_factorize(n, Ap.ctypes.data, Ai.ctypes.data, Alen.ctypes.data, Alimbs.ctypes.data,
                            q.ctypes.data, pivot, float(tol), kmax, cap)
    if not r:
        raise MemoryError("orc_factorize")
    R = r.contents
    K = R.K
    out = dict(n=n, K=K, status=R.status, seconds=R.seconds,
               counters=np.array(list(R.counters), dtype=np.int64))
    if R.status != -3:
        out.update(
            Lp=_arr(R.Lp, K + 1, np.int64), Up=_arr(R.Up, K + 1, np.int64),
            Li=_arr(R.Li, R.lnz, np.int32), Ui=_arr(R.Ui, R.unz, np.int32),
            Llen=_arr(R.Llen, R.lnz, np.int32), Ulen=_arr(R.Ulen, R.unz, np.int32),
            Llimbs=_arr(R.Llimbs, R.Lnl, np.uint64), Ulimbs=_arr(R.Ulimbs, R.Unl, np.uint64),
            rholen=_arr(R.rholen, K, np.int32), rholimbs=_arr(R.rholimbs, R.rhonl, np.uint64),
            pinv=_arr(R.pinv, n, np.int32))
    lib().orc_free(r)
    return out


def matgen(n, density, bits, seed):
    """slip_matgen.h through the oracle library -> Ap, Ai, Ax (int64 values)."""
    L = lib()
    pAp, pAi, pAx = C.c_void_p(), C.c_void_p(), C.c_void_p()
    if L.orc_matgen(n, density, bits, seed, C.byref(pAp), C.byref(pAi), C.byref(pAx)):
        raise MemoryError("orc_matgen")
    Ap = np.ctypeslib.as_array(C.cast(pAp, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    nnz = int(Ap[n])
    Ai = np.ctypeslib.as_array(C.cast(pAi, C.POINTER(C.c_int32)), shape=(nnz,)).copy()
    Ax = np.ctypeslib.as_array(C.cast(pAx, C.POINTER(C.c_int64)), shape=(nnz,)).copy()
    for p in (pAp, pAi, pAx):
        L.orc_free_ptr(p)
    return Ap, Ai, Ax
