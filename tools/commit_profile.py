#!/usr/bin/env python3
"""The committer's phase times on the headline window (libslip_hip_cprof.so, -DSLIP_PROFILE_COMMIT) as a small JSON for
profiles/: bench.py's roofline.chain takes the committer's serial time per column from the newest one.
usage: commit_profile.py out.json [case]"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import slip_lu_amd as sl
path = os.path.join(ROOT, "slip_lu_amd", "csrc", "libslip_hip_cprof.so")
name = sys.argv[2] if len(sys.argv) > 2 else "C4_n100k_c64"
entry, fix = load_case(name)
f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                     limb_cap=entry["cap"], lib_path=path)
best = None
for rep in range(5):
    f.reset(); f.run(entry["kmax"], check=False)
    i = f.info()
    out = (C.c_ulonglong * 24)()
    f.lib.slip_hip_factor_phase_cycles(f.h, out)
    rec = (i["kernel_ms"], [int(out[q]) for q in range(24)], i)
    if best is None or rec[0] < best[0]:
        best = rec
f.close()
ms, o, i = best
cols = max(o[7], 1); us = lambda q: o[q] / 100.0
serial = sum(us(q) for q in (3, 10, 11, 12, 13, 15, 16, 17, 18, 19))
res = dict(case=name, kernel_ms=ms, columns=i["K"], committer_commits=i["committer_commits"], engine_commits=i["engine_commits"],
           batches=o[6], committed_by_committer=o[7], rejects=o[8],
           us_total=dict(waiting_for_packages=us(0), packages_into_lds=us(1), rho_after_resync=us(2), serial=serial, publish_and_drain=us(4), verdicts=us(5)),
           serial_us_per_column=serial / cols, load_us_per_column=us(1) / cols, publish_us_per_column=(us(4) + us(5)) / cols,
           note="times of thread 0 of the committer workgroup (s_memrealtime, 10 ns ticks); each stamp costs about 0.05 us, ten per column")
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res))
