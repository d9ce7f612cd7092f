#!/usr/bin/env python3
"""Effect of the master/helper workgroup size (waves per workgroup) on golden cases (development aid)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slabfile
from conftest import load_case
import slip_lu_amd as sl
cases = sys.argv[1].split(","); waves = [int(x) for x in sys.argv[2].split(",")]
for name in cases:
    entry, fix = load_case(name)
    for W in waves:
        f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                             tol=entry["tol"], limb_cap=entry["cap"], waves=W)
        rc = f.run(entry["kmax"], check=False)
        ok = slabfile.factor_digest(f.download()) == entry["digest"] if f.info()["K"] > 0 else None
        f.reset(); f.run(entry["kmax"], check=False); i = f.info(); f.close()
        print(json.dumps(dict(case=name, waves=W, rc=rc, parity=ok, kernel_ms=round(i["kernel_ms"], 2))), flush=True)
