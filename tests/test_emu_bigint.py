"""wave_bigint.h / wave_bigint_reg.h primitives, run lane-by-lane on the CPU emulator, vs Python ints."""
import ctypes as C
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import ROOT

SO = os.path.join(ROOT, "tests", "emu", "libemu_bigint.so")


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libemu_bigint.so"])
    return C.CDLL(SO)


def digs(v, n):
    return np.array([(v >> (32 * i)) & 0xFFFFFFFF for i in range(max(n, 1))], dtype=np.uint32)


def val(a, n):
    return sum(int(a[i]) << (32 * i) for i in range(n))


def P(a):
    return a.ctypes.data_as(C.c_void_p)


def rnd(r, nd, kind):
    if nd == 0:
        return 0
    v = r.getrandbits(32 * nd) if kind == 0 else (1 << (32 * nd)) - 1 - (r.getrandbits(8) if kind == 2 else 0)
    return v | (1 << (32 * nd - 1))


def test_memory_primitives(L):
    r = random.Random(1)
    for _ in range(60):
        la = r.choice([0, 1, 2, 3, 17, 64, 65, 129, 200]); lb = r.choice([1, 2, 31, 64, 66, 130, 190])
        W = r.choice([1, 2, 64, 65, 128, 130, 200, la + lb, max(1, la + lb - 1)])
        a, b = rnd(r, la, r.randrange(3)), rnd(r, lb, r.randrange(3))
        out = np.zeros(W + 1, dtype=np.uint32)
        L.emu_mul_lo(P(digs(a, la)), la, P(digs(b, lb)), lb, W, P(out))
        assert val(out, W) == (a * b) % (1 << (32 * W))
        for sub in (0, 1):
            L.emu_addsub(P(digs(a, la)), la, P(digs(b, lb)), lb, W, sub, P(out))
            aa, bb = a % (1 << (32 * min(la, W))) if la else 0, b % (1 << (32 * min(lb, W)))
            assert val(out, W) == ((aa - bb) if sub else (aa + bb)) % (1 << (32 * W))
        d = rnd(r, lb, 0) | 1; want = r.choice([1, 2, 7, 64, 65, 130])
        inv = np.zeros(want, dtype=np.uint32)
        L.emu_inv(P(digs(d, lb)), lb, 0, want, P(inv))
        assert (val(inv, want) * d) % (1 << (32 * want)) == 1


def test_register_primitives(L):
    r = random.Random(2)
    for _ in range(60):
        D = r.choice([1, 2, 3, 4]); cap = 64 * D
        la = min(r.choice([1, 2, 3, 17, 64, 65, 129, 200, 256]), cap); lb = min(r.choice([1, 2, 31, 64, 66, 130, 256]), cap)
        a, b = rnd(r, la, r.randrange(3)), rnd(r, lb, r.randrange(3)); M = 1 << (32 * cap)
        out = np.zeros(cap, dtype=np.uint32)
        L.emu_reg_mul(D, P(digs(a, la)), la, P(digs(b, lb)), lb, cap, P(out))
        assert val(out, cap) == (a * b) % M
        for sub in (0, 1):
            L.emu_reg_addsub(D, P(digs(a, la)), la, P(digs(b, lb)), lb, cap, sub, P(out))
            assert val(out, cap) == ((a - b) if sub else (a + b)) % M
        sh = min(r.choice([0, 1, 31, 32, 33, 95, 2000]), 32 * cap - 1)
        L.emu_reg_shr(D, P(digs(b, lb)), lb, sh, cap, P(out))
        assert val(out, cap) == (b >> sh) % M
        d = rnd(r, lb, 0) | 1; want = r.randrange(1, cap + 1)
        inv = np.zeros(cap, dtype=np.uint32)
        L.emu_reg_inv(D, P(digs(d, lb)), lb, 0, want, P(inv))
        assert (val(inv, want) * d) % (1 << (32 * want)) == 1


def test_register_digit_multiply(L):
    """wr_mul_digit: one-digit a times a long number (the rows of a column times the previous pivot)"""
    r = random.Random(3)
    for _ in range(80):
        D = r.choice([1, 2, 3, 4]); cap = 64 * D
        lb = min(r.choice([1, 2, 63, 64, 65, 127, 128, 191, 255]), cap - 1)
        b = rnd(r, lb, r.randrange(3))
        a = r.choice([1, 2, 0xFFFFFFFF, 0x80000000, r.getrandbits(32) | 1, r.getrandbits(16) | 1])
        out = np.zeros(cap, dtype=np.uint32)
        L.emu_reg_mul_digit(D, C.c_uint32(a), P(digs(b, lb)), lb, cap, P(out))
        assert val(out, cap) == a * b
        a2 = r.choice([1, 0xFFFFFFFF, r.getrandbits(32) | 1]); out2 = np.zeros(cap, dtype=np.uint32)
        L.emu_reg_mul_digit2(D, C.c_uint32(a), C.c_uint32(a2), P(digs(b, lb)), lb, cap, P(out), P(out2))
        assert val(out, cap) == a * b and val(out2, cap) == a2 * b
