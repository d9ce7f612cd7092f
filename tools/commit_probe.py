#!/usr/bin/env python3
"""How many columns the committer / the short chain took, and the kernel time, with and without the committer.
usage: commit_probe.py case[,case...] [workers] [waves]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import slip_lu_amd as sl
LIB = os.environ.get("SLIP_PROBE_LIB")
FLAGS = [int(x) for x in os.environ.get("SLIP_PROBE_FLAGS", "0,4,2").split(",")]
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name in sys.argv[1].split(","):
    entry, fix = load_case(name)
    for flags in FLAGS:
        f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                             tol=entry["tol"], limb_cap=entry["cap"], workers=workers, waves=waves, debug_flags=flags, **({"lib_path": LIB} if LIB else {}))
        f.run(entry["kmax"], check=False); i0 = f.info(); f.reset(); rc = f.run(entry["kmax"], check=False)
        i = f.info()
        print(f"{name}: first run launches {i0['launches']} xcap {i0['xcap_digits']} ms {i0['kernel_ms']:.2f}; second: launches {i['launches']} xcap {i['xcap_digits']}; flags {flags} rc {rc} K {i['K']} short_commits {i['short_commits']} by committer {i['committer_commits']} farm jobs {i['farm_jobs']} helped items {i['farm_items']} workers {i['workers']} kernel_ms {i['kernel_ms']:.3f} "
              f"= {1e3 * i['kernel_ms'] / max(i['K'], 1):.2f} us/col", flush=True)
        f.close()
