#!/usr/bin/env python3
"""Per-phase cycle breakdown of the column workers from the diagnostic build (libslip_hip_prof.so, `make prof`).
Thread 0 of every worker stamps its phases; the sums over all workers are divided by the columns committed.
usage: phase_probe.py case[,case...] [workers] [waves]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
import slip_lu_amd as sl
path = os.environ.get("SLIP_PROF_LIB") or os.path.join(ROOT, "slip_lu_amd", "csrc", "libslip_hip_prof.so")
SLOTS = {0: "scatter", 1: "sweep(work)", 16: "wait F", 17: "wait F2", 20: "  sweep after last F wait", 2: "position snapshot",
         21: "early: classify", 22: "early: reduce+candidates", 13: "early: search+diag (or full search)", 6: "stage1 publish",
         14: "pattern+rank", 8: "hist:stage rho", 9: "hist:classify", 11: "hist:mul (class A)", 10: "hist:drain (divisions)",
         3: "hist:rest", 12: "table+cap", 4: "diag rule", 5: "offsets", 7: "stage2 copy",
         18: "CRITICAL: F seen -> F stored", 23: "(early commits per column)", 15: "(early candidates computed per column)"}
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 0
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
for name in sys.argv[1].split(","):
    entry, fix = load_case(name)
    f = sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                         tol=entry["tol"], limb_cap=entry["cap"], lib_path=path, workers=workers, waves=waves, debug_flags=int(os.environ.get("SLIP_FLAGS", "0")))
    f.run(entry["kmax"], check=False); f.reset(); f.run(entry["kmax"], check=False)
    i = f.info()
    out = (C.c_ulonglong * 24)()
    f.lib.slip_hip_factor_phase_cycles(f.h, out)
    K = max(i["K"], 1)
    print(f"{name}: K {i['K']} workers {i['workers']} waves {i['waves']} kernel_ms {i['kernel_ms']:.2f} = {1e3 * i['kernel_ms'] / K:.1f} us/col; "
          f"columns stamped {out[19]}; cycles per committed column:")
    for slot, label in SLOTS.items():
        print(f"    {label:36s} {out[slot] / K:12.2f}")
    if hasattr(f.lib, "slip_hip_factor_column_trace"):
        import numpy as np
        tr = np.zeros(24 * i["K"], np.int32)
        f.lib.slip_hip_factor_column_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        if f.lib.slip_hip_factor_column_trace(f.h, tr.ctypes.data, i["K"]) == 0:
            seen = tr[8 * i["K"]:9 * i["K"]].astype(np.int64)
            flags = tr[17 * i["K"]:18 * i["K"]]
            tl = tr[18 * i["K"]:24 * i["K"]].reshape(-1, 6).astype(np.int64)
            if os.environ.get("SLIP_TIMELINE"):
                t0 = tl[:, 0].min()
                np.save(os.environ["SLIP_TIMELINE"], np.concatenate([tl - t0, flags.reshape(-1, 1).astype(np.int64)], axis=1))
            sub = tr[9 * i["K"]:17 * i["K"]].reshape(-1, 8)
            names = ["sweep tail", "loads+classify", "rho staging+barrier", "reduce+barrier", "marking+barrier", "cand multiply+barrier", "search", "publish issue (7b: then drain)"]
            good = sub[:, 1] > 0
            print("    chain sub-steps, median cycles: " + ", ".join(f"{n} {int(np.median(sub[good, q]))}" for q, n in enumerate(names)))
            tr = tr[:8 * i["K"]].reshape(-1, 8)
            hop = (seen[1:] - tr[:-1, 6].astype(np.int64)) & 0xFFFFFFFF
            hop = hop[(seen[1:] != 0) & (hop < 10 ** 7)] * 10.0        # ns
            if len(hop):
                print(f"    hand-off (F stored by column k-1 -> seen by column k's worker): mean {hop.mean():.0f} ns median {np.median(hop):.0f} ns p90 {np.percentile(hop, 90):.0f} ns")
            c = tr[:, 0].astype(np.float64)
            ok = c >= 0
            print(f"    commit-chain cycles per column: n={ok.sum()} mean {c[ok].mean():.0f} median {np.median(c[ok]):.0f} "
                  f"p90 {np.percentile(c[ok], 90):.0f} max {c[ok].max():.0f}; early {int(tr[:, 1].sum())} of {len(tr)}")
            worst = np.argsort(-c)[:12]
            for kcol in worst:
                print(f"      col {kcol}: chain {tr[kcol, 0]} early {tr[kcol, 1]} cand {tr[kcol, 2]} rows {tr[kcol, 3]} "
                      f"sweep-tail+snapshot {tr[kcol, 4]} pass+publish {tr[kcol, 5]} worker {tr[kcol, 7]} path {int(flags[kcol]) & 0xFFFF:#x}")
            e = tr[:, 1] == 1
            if e.any():
                print(f"    early columns: sweep-tail+snapshot mean {tr[e, 4].mean():.0f}, pass+publish mean {tr[e, 5].mean():.0f} "
                      f"median {np.median(tr[e, 5]):.0f}")
    if hasattr(f.lib, "slip_hip_factor_heavy_trace") and os.environ.get("SLIP_HEAVY"):
        import numpy as np
        hv = np.zeros(2048, np.int32)
        f.lib.slip_hip_factor_heavy_trace.argtypes = [C.c_void_p, C.c_void_p]
        f.lib.slip_hip_factor_heavy_trace(f.h, hv.ctypes.data)
        hv = hv.reshape(64, 32)
        for rec in hv:
            if rec[30] <= 0:
                continue
            t0 = int(rec[29]) & 0xFFFFFFFF
            stamps = sorted(((int(rec[s_]) - t0) & 0xFFFFFFFF, s_) for s_ in range(24) if rec[s_] != 0)
            print(f"    heavy col {rec[31]} rows {rec[30]}: " + ", ".join(f"{SLOTS.get(s_, s_)}@{t / 100.0:.0f}us" for t, s_ in stamps))
    f.close()
