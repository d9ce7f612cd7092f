#!/usr/bin/env python3
"""Where the committer workgroup spends its time (libslip_hip_cprof.so: hipcc ... -DSLIP_PROFILE_COMMIT).
usage: commit_phase_probe.py case[,case...] [workers] [waves]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import slip_lu_amd as sl
path = os.path.join(ROOT, "slip_lu_amd", "csrc", "libslip_hip_cprof.so")
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
names = ["waiting for packages", "packages into LDS", "rho after a resync", "serial part (wave 0)", "publish + drain", "verdicts + frontier"]
for name in sys.argv[1].split(","):
    entry, fix = load_case(name)
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                         tol=entry["tol"], limb_cap=entry["cap"], lib_path=path, workers=workers, waves=waves)
    f.run(entry["kmax"], check=False); f.reset(); f.run(entry["kmax"], check=False)
    i = f.info()
    out = (C.c_ulonglong * 24)()
    f.lib.slip_hip_factor_phase_cycles(f.h, out)
    print(f"{name}: K {i['K']} kernel_ms {i['kernel_ms']:.3f} by committer {i['committer_commits']} (engine {i['engine_commits']}, late sources {i['engine_sources']}); "
          f"batches {out[6]} columns {out[7]} rejects {out[8]} ready-at-poll {out[9]} retractions {i['retractions']} re-exports {i['reexports']}")
    for q, nm in enumerate(names):
        print(f"    {nm:28s} {out[q] / 100.0:10.1f} us total  {out[q] / 100.0 / max(out[6], 1):8.2f} us per batch  {out[q] / 100.0 / max(out[7], 1):8.2f} us per column")
    for q, nm in ((10, "c: setup + intermed2"), (11, "c0: rows vs pivots"), (12, "c0: capacity"), (13, "c0: choose + diag"), (15, "c0: rho multiply"),
                  (17, "c1: state + hash"), (18, "c1: late sources"), (19, "c1: finals + search + hand-back"), (16, "c: record + rings")):
        print(f"        {nm:34s} {out[q] / 100.0:10.1f} us total  {out[q] / 100.0 / max(out[7], 1):8.2f} us per column")
    f.close()
