#!/usr/bin/env python3
"""Development probe: rerun a case until the factors differ from the CPU restatement, then show where."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, slabfile
import slip_lu_amd as sl
from conftest import load_case
name, reps, workers = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
entry, fix = load_case(name)
ref = oracle_lib.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], kmax=entry["kmax"], cap=entry["cap"], tol=entry["tol"])
f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"], limb_cap=entry["cap"], workers=workers)
for rep in range(reps):
    f.reset(); rc = f.run(entry["kmax"], check=False)
    got = f.download()
    if rc == 0 and slabfile.factor_digest(got) == entry["digest"]:
        continue
    print("rep", rep, "rc", rc, "K", got["K"])
    inv = np.argsort(ref["pinv"])
    for k in ("pinv", "Lp", "Up", "Li", "Ui", "Llen", "Ulen", "rholen", "Llimbs", "Ulimbs", "rholimbs"):
        a, b = np.asarray(got[k]).astype(np.int64), np.asarray(ref[k]).astype(np.int64)
        if a.shape != b.shape:
            print("  ", k, "shape", a.shape, b.shape); continue
        d = np.nonzero(a != b)[0]
        if len(d):
            print("  ", k, "differs at", len(d), "first", d[:6], "got", a[d[:4]], "want", b[d[:4]])
            if k in ("Li", "Llen", "Ui", "Ulen"):
                P = np.asarray(ref["Lp" if k[0] == "L" else "Up"])
                cols = np.unique(np.searchsorted(P, d, side="right") - 1)
                print("      columns", cols[:10], "count", len(cols), "col sizes", [int(P[c + 1] - P[c]) for c in cols[:5]])
            if k == "pinv":
                print("      rows", d[:6], "ref pos", b[d[:6]], "got pos", a[d[:6]])
    break
else:
    print("no mismatch in", reps)
f.close()
