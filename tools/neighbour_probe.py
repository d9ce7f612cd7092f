#!/usr/bin/env python3
"""Does the column worker on the CU next to the committer slow the committer down (shared instruction cache)?  Runs a case
with the workers on the committer's neighbouring CUs standing aside (debug flag bits 4-5 = distance) and prints the
committer's phase times (libslip_hip_cprof.so) and where the workgroups ran.  usage: neighbour_probe.py case [flags ...]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_case
import slip_lu_amd as sl
path = os.path.join(ROOT, "slip_lu_amd", "csrc", "libslip_hip_cprof.so")
name = sys.argv[1]
flagsets = [int(x) for x in sys.argv[2:]] or [0, 16, 32]
entry, fix = load_case(name)
for flags in flagsets:
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                         limb_cap=entry["cap"], lib_path=path, debug_flags=flags)
    best = None
    for rep in range(4):
        f.reset(); f.run(entry["kmax"], check=False)
        i = f.info()
        out = (C.c_ulonglong * 24)()
        f.lib.slip_hip_factor_phase_cycles(f.h, out)
        rec = (i["kernel_ms"], [out[q] / 100.0 for q in range(20)], i)
        if best is None or rec[0] < best[0]:
            best = rec
    ms, o, i = best
    cols = max(o[7] * 100, 1)
    print(f"{name} flags {flags}: K {i['K']} kernel_ms {ms:.3f} committer {i['committer_commits']} engine {i['engine_commits']}/{i['engine_sources']} "
          f"wait {o[0]:.0f} load {o[1]:.0f} serial {o[3]:.0f} publish {o[4]:.0f} verdict {o[5]:.0f} us; per column: serial {o[3] / cols:.2f} "
          f"[setup {o[10] / cols:.2f} rows {o[11] / cols:.2f} cap {o[12] / cols:.2f} choose {o[13] / cols:.2f} mul {o[15] / cols:.2f} "
          f"e1 {o[17] / cols:.2f} e3 {o[18] / cols:.2f} e4 {o[19] / cols:.2f} record {o[16] / cols:.2f}]")
    f.lib.slip_hip_factor_phase_cycles(f.h, out)
    c0, c1, c2 = int(out[20]), int(out[21]), int(out[22])
    if c0:
        print(f"    calibration in the committer: dependent LDS read {(c0 >> 32) / 256:.0f} cycles = {(c0 & 0xFFFFFFFF) * 10 / 256:.0f} ns; dependent VALU mul-add "
              f"{(c1 >> 32) / 1024:.1f} cycles = {(c1 & 0xFFFFFFFF) * 10 / 1024:.1f} ns; s_memrealtime stamp {c2 / 64:.0f} cycles; clock {(c1 >> 32) / max((c1 & 0xFFFFFFFF) * 10, 1):.2f} GHz")
    w = np.zeros(2048, np.int32)
    f.lib.slip_hip_factor_debug_words.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    if f.lib.slip_hip_factor_debug_words(f.h, 24 * entry["n"] + 2048, 2048, w.ctypes.data) == 0 and flags == flagsets[0]:
        w = w[w != 0] & 0xFFFF
        dec = [(int(v >> 8) & 0xF, int(v >> 5) & 0x7, int(v >> 4) & 1, int(v) & 0xF) for v in w]
        print("    committer (block 0) at xcc/se/sh/cu", dec[0], "; blocks per xcc:", np.bincount([d[0] for d in dec]).tolist(),
              "; distinct cu ids:", sorted(set(d[3] for d in dec)), "; se ids:", sorted(set(d[1] for d in dec)), "; sh:", sorted(set(d[2] for d in dec)))
        same = [(b, d) for b, d in enumerate(dec) if d[:3] == dec[0][:3]]
        print("    blocks in the committer's shader array:", same[:20])
    f.close()
