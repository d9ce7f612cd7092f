#!/usr/bin/env python3
"""Run golden cases on the GPU, check the digest, print device timings (development aid)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import slabfile  # noqa: E402
from conftest import load_case  # noqa: E402
import slip_lu_amd as sl  # noqa: E402

names = sys.argv[1:]
rows = []
for name in names:
    waves = 0
    if "@" in name:
        name, w = name.split("@"); waves = int(w)
    entry, fix = load_case(name)
    t0 = time.time()
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"],
                         pivot=entry["pivot"], tol=entry["tol"], limb_cap=entry["cap"], waves=waves)
    t1 = time.time()
    rc = f.run(entry["kmax"], check=False)
    t2 = time.time()
    i = f.info()
    ok = None
    if i["K"] > 0:
        ok = slabfile.factor_digest(f.download()) == entry["digest"]
    f.reset(); f.run(entry["kmax"], check=False); i2 = f.info()
    f.close()
    nnz = i["lnz"] + i["unz"] - i["K"]
    row = dict(case=name, waves=waves, rc=rc, K=i["K"], nnz=nnz, parity=ok, kernel_ms=round(i["kernel_ms"], 3),
               kernel_ms_2nd=round(i2["kernel_ms"], 3), launches=i["launches"], run_wall_ms=round((t2 - t1) * 1e3, 1),
               create_ms=round((t1 - t0) * 1e3, 1), knnz_per_s=round(nnz / max(i2["kernel_ms"], 1e-9), 1),
               b_read=i["b_read"], b_write=i["b_write"], gbs=round((i["b_read"]) / max(i2["kernel_ms"], 1e-9) / 1e6, 3),
               ref_s=entry["ref_seconds"], speedup_vs_ref=round(entry["ref_seconds"] * 1e3 / max(i2["kernel_ms"], 1e-9), 2))
    rows.append(row)
    print(json.dumps(row), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "probe.jsonl"), "a") as fh:
    for r in rows:
        fh.write(json.dumps(r) + "\n")
