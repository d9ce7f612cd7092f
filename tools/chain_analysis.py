#!/usr/bin/env python3
"""Dependency structure of a golden case from the oracle's factors (CPU): per column the sources (pivot positions of
the rows of U(:,k) below the pivot), the longest dependency chain through the column DAG (column k depends on column j
when j is a source of k), and how the source distances are distributed.  The chain depth is the lower bound on the
number of sequential hops any schedule of the left-looking loop needs (bench.py: roofline.chain).
usage: chain_analysis.py case [case ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_case
import oracle_lib


def analyse(name):
    entry, fix = load_case(name)
    r = oracle_lib.factorize(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                             tol=entry["tol"], kmax=entry["kmax"], cap=entry["cap"])
    K = r["K"]; Up, Ui, pinv, Lp = r["Up"], r["Ui"], r["pinv"], r["Lp"]
    Ulen, Llen = r["Ulen"], r["Llen"]
    depth = np.zeros(K, np.int64); nsrc = np.zeros(K, np.int64); lastsrc = np.full(K, -1, np.int64)
    wchain = np.zeros(K, np.float64)       # chain weighted by the L entries streamed
    maxl = np.zeros(K, np.int64)
    for k in range(K):
        rows = Ui[Up[k]:Up[k + 1] - 1]                     # the pivot is last
        src = pinv[rows]
        lens = Ulen[Up[k]:Up[k + 1] - 1]
        src = src[lens != 0]                               # explicit zeros are not sources
        nsrc[k] = len(src)
        maxl[k] = np.abs(Llen[Lp[k]:Lp[k + 1]]).max() if Lp[k + 1] > Lp[k] else 0
        if len(src):
            lastsrc[k] = src.max()
            depth[k] = depth[src].max() + 1
            wchain[k] = (wchain[src] + (Lp[src + 1] - Lp[src])).max()
    dist = (np.arange(K) - lastsrc)[lastsrc >= 0]
    print(f"{name}: K {K} nnz {len(r['Li']) + len(r['Ui']) - K} columns with sources {int((nsrc > 0).sum())} source applications {int(nsrc.sum())} "
          f"chain depth {int(depth.max())} (columns on the longest chain + 1)")
    if len(dist):
        print(f"    distance to the latest source: median {int(np.median(dist))} p10 {int(np.percentile(dist, 10))} min {int(dist.min())}; "
              f"within 1: {int((dist <= 1).sum())} within 4: {int((dist <= 4).sum())} within 16: {int((dist <= 16).sum())} within 64: {int((dist <= 64).sum())} within 256: {int((dist <= 256).sum())}")
    lcol = Lp[1:K + 1] - Lp[:K]
    print(f"    rows per L column: mean {lcol.mean():.1f} max {int(lcol.max())}; max limbs per column: median {int(np.median(maxl))} max {int(maxl.max())}; "
          f"columns with every value one limb: {int((maxl <= 1).sum())}")
    return dict(K=K, depth=int(depth.max()), nsrc=nsrc, lastsrc=lastsrc, depthv=depth, lcol=lcol, maxl=maxl)


if __name__ == "__main__":
    for nm in sys.argv[1:]:
        analyse(nm)
