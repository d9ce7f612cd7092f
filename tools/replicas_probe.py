#!/usr/bin/env python3
"""Throughput view (never the headline): B independent copies of one window on B handles of ONE GPU, each with its own HIP
stream and W workers, run concurrently from B host threads; aggregate L+U nonzeros per second against one copy alone.
usage: replicas_probe.py case B[,B...] W[,W...] [reps]"""
import ctypes as C, json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slip_lu_amd as sl
from conftest import load_case


def run_replicas(entry, fix, B, W, reps=3):
    hip = C.CDLL("libamdhip64.so")
    hs = [sl.Factorization(entry["n"], fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"], fix["q"], pivot=entry["pivot"],
                           tol=entry["tol"], limb_cap=entry["cap"], workers=W) for _ in range(B)]
    streams = []
    for _ in range(B):
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        streams.append(s)
    best = None
    try:
        for r in range(reps + 1):
            for h in hs:
                h.reset()
            rcs = [None] * B

            def work(t):
                rcs[t] = hs[t].run(entry["kmax"], stream=streams[t].value, check=False)
            th = [threading.Thread(target=work, args=(t,)) for t in range(B)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt = time.perf_counter() - t0
            infos = [h.info() for h in hs]
            ok = all(i["K"] == entry["K"] and i["b_read"] == entry["counters"]["B_read"] for i in infos) and all(rc == entry["status"] for rc in rcs)
            if r and (best is None or dt < best[0]):
                best = (dt, ok, [round(i["kernel_ms"], 3) for i in infos], infos[0]["workers"])
    finally:
        for h in hs: h.close()
        for s in streams: hip.hipStreamDestroy(s)
    nnz = entry["lnz"] + entry["unz"] - entry["K"] if "lnz" in entry else None
    return dict(replicas=B, workers_each=best[3], wall_ms=round(1e3 * best[0], 3), ok=best[1], kernel_ms_each=best[2])


if __name__ == "__main__":
    name = sys.argv[1]
    Bs = [int(x) for x in sys.argv[2].split(",")]
    Ws = [int(x) for x in sys.argv[3].split(",")]
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    entry, fix = load_case(name)
    for W in Ws:
        for B in Bs:
            out = run_replicas(entry, fix, B, W, reps)
            out["case"] = name
            out["windows_per_s"] = round(B / (out["wall_ms"] * 1e-3), 1)
            print(json.dumps(out), flush=True)
