#!/usr/bin/env python3
"""Cycles per wave-level big-integer primitive on the GPU (development aid)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slip_lu_amd import _lib
lib = C.CDLL(_lib.DEFAULT_SO)
names = {0: "mul_lo", 1: "sub", 2: "len", 3: "shr", 4: "mul+mul", 5: "wave_sync", 6: "reg_mul", 7: "hensel", 8: "newton+mul"}
for nw in (1, 8):
    for lds in (1,):
        for (op, la, lb, W) in [(0, 2, 128, 130), (0, 1, 64, 65), (0, 2, 2, 4), (0, 64, 64, 128), (0, 128, 128, 128),
                                (6, 2, 128, 130), (6, 2, 2, 4), (6, 64, 64, 128), (6, 128, 128, 128), (6, 43, 86, 86), (6, 128, 128, 192), (6, 16, 16, 32),
                                (7, 16, 8, 16), (8, 16, 8, 16), (7, 64, 64, 64), (8, 64, 64, 64), (7, 128, 64, 128), (8, 128, 64, 128), (7, 200, 128, 200), (8, 200, 128, 200)]:
            out = (C.c_ulonglong * nw)()
            rc = lib.slip_hip_wave_op_bench(op, la, lb, W, 200, nw, lds, out)
            print(f"waves={nw} out={'lds' if lds else 'global'} {names[op]:9s} la={la:3d} lb={lb:3d} W={W:3d}: "
                  f"{min(out)}..{max(out)} cycles", flush=True)
