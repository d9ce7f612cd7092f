#!/usr/bin/env python3
"""Development probe: one case, one launch shape, repeated; prints the status and the diagnostic counters of every run.
usage: w2_probe.py case waves workers reps [flags]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import slip_lu_amd as sl
name, waves, workers, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
entry, fix = load_case(name)
reuse = os.environ.get("SLIP_REUSE")
f = None
for rep in range(reps):
    if reuse and f is not None:
        f.reset(); rc = f.run(entry["kmax"], check=False); i = f.info()
        print(rep, "rc", rc, "K", i["K"], "launches", i["launches"], "short", i["short_commits"], "committer", i["committer_commits"], "farm", i["farm_jobs"], i["farm_items"], "ms", round(i["kernel_ms"], 2), flush=True)
        continue
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"], tol=entry["tol"],
                         limb_cap=entry["cap"], waves=waves, workers=workers, debug_flags=flags, **({"lib_path": os.environ["SLIP_PROBE_LIB"]} if os.environ.get("SLIP_PROBE_LIB") else {}))
    rc = f.run(entry["kmax"], check=False)
    i = f.info()
    print(rep, "rc", rc, "K", i["K"], "launches", i["launches"], "short", i["short_commits"], "committer", i["committer_commits"], "farm", i["farm_jobs"], i["farm_items"], "ms", round(i["kernel_ms"], 2), flush=True)
    if not reuse:
        f.close()
