#!/usr/bin/env python3
"""Development probe: repeat a golden case under several launch shapes and report how often the factors differ.
usage: race_probe.py case reps"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slip_lu_amd as sl
import slabfile
from conftest import load_case
name, reps = sys.argv[1], int(sys.argv[2])
libp = sys.argv[3] if len(sys.argv) > 3 else None
shapes = eval(sys.argv[4]) if len(sys.argv) > 4 else ((8, 0, 0), (4, 0, 0), (8, 0, 1), (8, 32, 0), (2, 0, 0))
entry, fix = load_case(name)
for waves, workers, flags in shapes:
    bad = []
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                         limb_cap=entry["cap"], waves=waves, workers=workers, debug_flags=flags, lib_path=libp)
    for rep in range(reps):
        f.reset(); rc = f.run(entry["kmax"], check=False)
        d = f.download()
        ok = rc == 0 and d["K"] == entry["K"] and np.array_equal(d["pinv"], fix["pinv"]) and slabfile.factor_digest(d) == entry["digest"]
        if not ok:
            bad.append((rep, rc, d["K"]))
    print(json.dumps(dict(lib=os.path.basename(libp or "default"), case=name, waves=waves, workers=f.info()["workers"], no_early=flags, reps=reps, bad=bad[:6], nbad=len(bad))), flush=True)
    f.close()
