#!/usr/bin/env python3
"""Run golden cases with diagnostic flag sets, several repetitions each; print parity and the path counters.
usage: flags_probe.py case[,case] flags[,flags...] [reps] [workers]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slabfile
from conftest import load_case
import slip_lu_amd as sl
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 0
libp = os.environ.get("SLIP_PROBE_LIB")
for name in sys.argv[1].split(","):
    entry, fix = load_case(name)
    for flags in [int(x) for x in sys.argv[2].split(",")]:
        f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                             limb_cap=entry["cap"], debug_flags=flags, workers=workers, lib_path=libp)
        for rep in range(reps):
            f.reset()
            rc = f.run(entry["kmax"], check=False)
            i = f.info()
            ok = None
            if rc == 0 and i["K"] > 0:
                ok = slabfile.factor_digest(f.download()) == entry["digest"]
            print(f"{name} flags {flags} rep {rep}: rc {rc} K {i['K']} parity {ok} ms {i['kernel_ms']:.3f} launches {i['launches']} committer {i['committer_commits']} "
                  f"engine {i['engine_commits']}/{i['engine_sources']} short {i['short_commits']} retract {i['retractions']}/{i['reexports']} farm {i['farm_jobs']}", flush=True)
            if rc != 0:
                break
        f.close()
