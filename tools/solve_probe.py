#!/usr/bin/env python3
"""Kernel time of SLIP_LU_solve's substitutions on the device for complete-run goldens, one and sixteen right-hand sides.
usage: solve_probe.py case[,case...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slip_lu_amd as sl
from conftest import load_case
for name in sys.argv[1].split(","):
    e, fx = load_case(name)
    g = sl.Factorization(e["n"], fx["Ap"], fx["Ai"], fx["Alen"], fx["Alimbs"], fx["q"], limb_cap=e["cap"])
    g.run(e["kmax"])
    b = (np.arange(e["n"], dtype=np.int64) * 2654435761 % (1 << 32)) % 2001 - 1000
    bl, bx = np.sign(b).astype(np.int32), np.abs(b[b != 0]).astype(np.uint64)
    x1 = g.solve(bl, bx); x1 = g.solve(bl, bx)
    t1 = g.solve_ms()
    i = g.info()
    import ctypes as C
    w = np.zeros(8, np.int32)
    g.lib.slip_hip_factor_debug_words.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    phases = None
    if g.lib.slip_hip_factor_debug_words(g.h, 24 * e["n"] + 3072, 8, w.ctypes.data) == 0 and w[4]:
        d = [((int(w[q + 1]) - int(w[q])) & 0xFFFFFFFF) / 100.0 for q in range(4)]
        phases = dict(scatter_us=round(d[0], 1), forward_us=round(d[1], 1), times_det_us=round(d[2], 1), backward_us=round(d[3], 1))
    bl16 = np.tile(bl, 16); bx16 = np.tile(bx, 16)
    x16 = g.solve(bl16, bx16, nrhs=16)
    t16 = g.solve_ms()
    same = np.array_equal(x16[0][:e["n"]], x1[0]) and np.array_equal(x16[0][-e["n"]:], x1[0])
    print(json.dumps(dict(case=name, n=e["n"], solve_ms=round(t1, 3), solve16_ms=round(t16, 3), farm_jobs=i.get("farm_jobs"), farm_items=i.get("farm_items"), same=bool(same), phases=phases)), flush=True)
    g.close()
