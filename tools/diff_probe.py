#!/usr/bin/env python3
"""Development probe: HIP factors vs the CPU restatement, first differences per array.
usage: diff_probe.py case[,case...] [workers] [waves]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
import slip_lu_amd as sl
from conftest import load_case

workers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name in sys.argv[1].split(","):
    entry, fix = load_case(name)
    ref = oracle_lib.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                               kmax=entry["kmax"], cap=entry["cap"], tol=entry["tol"])
    got = sl.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                       kmax=entry["kmax"], limb_cap=entry["cap"], check=False, workers=workers, waves=waves)
    i = got["info"]
    print(name, "K", got["K"], ref["K"], "status", got["status"], "launches", i["launches"], "workers", i["workers"], "ms", round(i["kernel_ms"], 2))
    for k in ("pinv", "Lp", "Li", "Llen", "Up", "Ui", "Ulen", "rholen", "Llimbs", "Ulimbs", "rholimbs"):
        a, b = np.asarray(got[k]).astype(np.int64), np.asarray(ref[k]).astype(np.int64)
        if a.shape != b.shape:
            print("  ", k, "shape", a.shape, b.shape); continue
        d = np.nonzero(a != b)[0]
        if len(d):
            print("  ", k, "differs at", len(d), "of", len(a), "first", d[:8], "got", a[d[:4]], "want", b[d[:4]])
            if k in ("Li", "Llen", "Ui", "Ulen"):
                P = np.asarray(ref["Lp" if k[0] == "L" else "Up"])
                cols = np.searchsorted(P, d[:8], side="right") - 1
                print("      columns", cols, "col start", P[cols])
    print("   counters got", list(got["counters"]), "ref", list(ref["counters"]))
