#!/usr/bin/env python3
"""What a create -> run -> destroy cycle costs on the host (the drop-in SLIP_LU_factorize does one per call), with the process-level
pool of device buffers (default) and without (SLIP_HIP_POOL_MB=0).  usage: pool_probe.py case [cycles]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slip_lu_amd as sl
from conftest import load_case
name = sys.argv[1]; cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 6
e, fx = load_case(name)
ts = []
for c in range(cycles):
    t0 = time.perf_counter()
    f = sl.Factorization(e["n"], fx["Ap"], fx["Ai"], fx["Alen"], fx["Alimbs"], fx["q"], pivot=e["pivot"], tol=e["tol"], limb_cap=e["cap"])
    t1 = time.perf_counter()
    f.run(e["kmax"])
    t2 = time.perf_counter()
    kms = f.info()["kernel_ms"]
    f.close()
    t3 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), kms, 1e3 * (t3 - t2)))
print(name, "pool", os.environ.get("SLIP_HIP_POOL_MB", "default"), "create / run / kernel / destroy ms per cycle:")
for t in ts:
    print("   %.2f / %.2f / %.2f / %.2f" % t)
