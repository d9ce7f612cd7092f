#!/usr/bin/env python3
"""Where a launch's wall time goes, from the per-column time line of the diagnostic build (SLIP_TIMELINE=file tools/phase_probe.py case):
the columns in commit order, the gap each one's commit leaves after its predecessor's, and what the column was doing.
usage: gap_report.py timeline.npy [top]"""
import numpy as np, sys
a = np.load(sys.argv[1]); top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
us = lambda x: x / 100.0
start, exp, com, ver, end, swe, fl = a.T[:7]
K = len(com)
gap = np.diff(com, prepend=com[0])
adopted = (fl & 2) != 0
print(f"K {K}: first commit at {us(com[0]):.0f} us, last commit {us(com[-1]):.0f} us, last end {us(end.max()):.0f} us; committed by the committer {int(adopted.sum())}")
edges = [0, 3, 6, 12, 25, 50, 100, 200, 400, 10 ** 9]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (us(gap) >= lo) & (us(gap) < hi)
    print(f"  gaps {lo:4d}..{hi if hi < 10 ** 9 else 'inf':>4} us: {int(m.sum()):4d} columns, {us(gap[m].sum()):8.0f} us in all, {int((m & adopted).sum())} by the committer")
print("largest gaps:")
for k in np.argsort(-gap)[:top]:
    print(f"  col {k:5d}: gap {us(gap[k]):6.1f} us; sweep end -> commit {us(com[k] - swe[k]):7.1f}, prev commit -> sweep end {us(swe[k] - com[k - 1]) if k else 0:7.1f}, "
          f"export -> commit {us(com[k] - exp[k]) if exp[k] else -1:7.1f}, commit -> end {us(end[k] - com[k]):7.1f}; prev: commit -> end {us(end[k - 1] - com[k - 1]) if k else 0:7.1f}; flags {int(fl[k]) & 0xFFFF:#x} {'C' if adopted[k] else 'w'}")
