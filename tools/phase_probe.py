#!/usr/bin/env python3
"""Per-phase cycle breakdown from the diagnostic build (libslip_hip_prof.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import slip_lu_amd as sl
from slip_lu_amd import _lib
path = os.path.join(ROOT, "slip_lu_amd", "csrc", "libslip_hip_prof.so")
names = ["scatter", "sweep", "pattern", "Lhist", "pivot", "offsets", "copy", "commit"]
for name in sys.argv[1:]:
    entry, fix = load_case(name)
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                         tol=entry["tol"], limb_cap=entry["cap"], lib_path=path)
    f.run(entry["kmax"], check=False); f.reset(); f.run(entry["kmax"], check=False)
    i = f.info()
    out = (C.c_ulonglong * 20)()
    f.lib.slip_hip_factor_phase_cycles(f.h, out)
    tot = (sum(out[:8]) + sum(out[8:11]) + sum(out[12:16])) or 1
    print(name, "K", i["K"], "kernel_ms %.2f" % i["kernel_ms"], "cycles/col %.0f" % (tot / max(i["K"], 1)),
          " ".join(f"{n}={100.0 * out[j] / tot:.1f}%" for j, n in enumerate(names)),
          "| Lhist parts (cycles/col): stage=%d lanes=%d waves=%d wave_items/col=%.1f" % (
              out[8] / max(i["K"], 1), out[9] / max(i["K"], 1), out[10] / max(i["K"], 1), out[11] / max(i["K"], 1)),
          "| pivot parts: table=%d keys=%d (rest in pivot) | copy loop: first pass=%d second pass=%d (barrier in copy)" % (
              out[12] / max(i["K"], 1), out[13] / max(i["K"], 1), out[14] / max(i["K"], 1), out[15] / max(i["K"], 1)))
    f.close()
