#!/usr/bin/env python3
"""Development probe: kernel time of golden cases for several worker counts / waves.
usage: worker_probe.py case[,case...] workers[,workers...] [waves[,waves...]]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slip_lu_amd as sl
from conftest import load_case

cases = sys.argv[1].split(",")
workers = [int(x) for x in sys.argv[2].split(",")]
waves = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for name in cases:
    entry, fix = load_case(name)
    for wv in waves:
        for W in workers:
            f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                                 tol=entry["tol"], limb_cap=entry["cap"], workers=W, waves=wv)
            f.run(entry["kmax"], check=False); f.reset(); rc = f.run(entry["kmax"], check=False)
            i = f.info()
            ok = (i["K"] == entry["K"] and i["b_read"] == entry["counters"]["B_read"])
            print(json.dumps(dict(case=name, workers=i["workers"], waves=i["waves"], rc=rc, ok=ok, kernel_ms=round(i["kernel_ms"], 3),
                                  launches=i["launches"], us_per_col=round(1e3 * i["kernel_ms"] / max(i["K"], 1), 2))), flush=True)
            f.close()
