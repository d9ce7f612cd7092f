#!/usr/bin/env python3
"""bench.py -- the hot path's headline benchmark on MI355X.

Metric (BASELINE.json): L+U nonzeros/sec on the random n=100k, 0.1%-dense integer CSC
(configs[3], "limb cap 64"), plus achieved HBM GB/s against the gfx950 peak.

One *step* = one pass of the hot path over the workload: the left-looking REF LU column loop
(SLIP_LU_factorize) over the column window [0, K) of the matrix, K = the first column that
holds a value of more than 64 limbs (the full factorisation is infeasible for
anyone, so CPU and GPU run the same K).  Inputs (A, q) are resident in HBM before the timed
region; every step starts from column 0 (device-side reset included in the timing).

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: one rank per GPU under torch.distributed.run (started here as a child process when the
caller did not).  The path has no exploitable elimination-tree subtrees on this matrix class
(multi-GPU), so ranks are independent replicas of the same workload -- weak scaling,
no data-path collective; RCCL is used only for the barrier and the max-over-ranks of the step time.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# integer side: a 64x64-bit limb multiply-accumulate is four 32x32->64 v_mad_u64_u32; that instruction issues at a
# quarter of the 32-lanes-per-cycle VALU rate: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz / 4 digit-MACs per second
DIGIT_MAC_PEAK = 256 * 4 * 32 * 2.4e9 / 4
WORKLOAD = dict(n=100000, density=0.001, bits=16, seed=1, limb_cap=64, golden="C4_n100k_c64")
SECONDARY = ("10teams", "prob159", "NSR8K_w600", "rl5934", "rail4284", "fome12", "NSR8K")


def load_q(name):
    import slabfile
    return slabfile.load(os.path.join(ROOT, "tests", "golden", name + ".slab.gz"))["q"]


def relaunch_distributed(args):
    """--gpus N > 1 without a torchrun environment: start one rank per GPU as a CHILD process (nothing in this
    process has touched the GPU yet) and leave with its exit code."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__),
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_secondary:
        cmd.append("--no-secondary")
    if getattr(args, "farm", False):
        cmd.append("--farm")
    return cmd


def ref_driver_path():
    p = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    return p if os.path.exists(p) else None


def cpu_baseline(expect_K, expect_nnz):
    """The reference's own column loop on this host's cores (1 core: it is single-threaded),
    on the same window.  Prefers the compiled reference (oracle/_ref), else the C restatement."""
    w = WORKLOAD
    drv = ref_driver_path()
    sample = f"same workload: columns [0,{expect_K}) of gen:{w['n']},{w['density']},{w['bits']},{w['seed']}"
    if drv:
        try:
            import slabfile
            out = os.path.join("/tmp", f"slip_ref_{os.getpid()}.slab")
            spec = f"gen:{w['n']},{w['density']},{w['bits']},{w['seed']}"
            subprocess.run([drv, "window", spec, out, "0", str(w["limb_cap"])], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
            d = slabfile.load(out)
            os.unlink(out)
            K = int(d["K"][0])
            nnz = len(d["Li"]) + len(d["Ui"]) - K
            if K == expect_K and nnz == expect_nnz:
                t = float(d["timing"][0])
                return dict(value=nnz / t, unit="L+U nonzeros/s", cores=1, kind="reference",
                            seconds=t, sample=sample + " (reference's COLAMD order, compiled reference + GMP)")
        except Exception as e:                       # fall through to the port
            print(f"[bench] reference baseline unavailable: {e}", file=sys.stderr)
    import numpy as np
    import oracle_lib
    Ap, Ai, Ax = oracle_lib.matgen(w["n"], w["density"], w["bits"], w["seed"])
    r = oracle_lib.factorize(w["n"], Ap, Ai, np.sign(Ax).astype(np.int32), np.abs(Ax).astype(np.uint64),
                             load_q(w["golden"]), cap=w["limb_cap"])
    nnz = len(r["Li"]) + len(r["Ui"]) - r["K"]
    return dict(value=nnz / r["seconds"], unit="L+U nonzeros/s", cores=1, kind="port",
                seconds=r["seconds"], sample=sample + " (CPU restatement oracle/ref_lu_oracle.c)")


def write_triplet(path, fix, n):
    """the reference's triplet text format (Demo/demos.c:245-331), 1-based"""
    import numpy as np
    Ap, Ai, Alen, Alimbs = fix["Ap"], fix["Ai"], fix["Alen"], fix["Alimbs"]
    off = np.concatenate([[0], np.cumsum(np.abs(Alen))])
    with open(path, "w") as f:
        f.write(f"{n} {n} {len(Ai)}\n")
        for j in range(n):
            for p in range(int(Ap[j]), int(Ap[j + 1])):
                v = 0
                for t in range(abs(int(Alen[p]))):
                    v |= int(Alimbs[off[p] + t]) << (64 * t)
                if Alen[p] < 0:
                    v = -v
                f.write(f"{int(Ai[p]) + 1} {j + 1} {v}\n")


def reference_seconds(name, entry, fix):
    """(factor seconds, solve seconds or None) of the compiled reference on THIS host for a secondary workload, or
    (None, None) when oracle/_ref is not there.  Complete runs time SLIP_LU_factorize / SLIP_LU_solve themselves."""
    drv = ref_driver_path()
    if not drv:
        return None, None
    import slabfile
    base = os.path.join("/tmp", f"slip_sec_{os.getpid()}_{name}")
    trip, out, qf = base + ".txt", base + ".slab", os.path.join(ROOT, "tests", "golden", name + ".slab.gz")
    try:
        write_triplet(trip, fix, entry["n"])
        # the golden holds q; ref_driver reads plain slab files
        qplain = base + "_q.slab"
        slabfile.save(qplain, {"q": fix["q"]})
        if entry["kmax"] or entry["cap"]:
            cmd = [drv, "window", "trip:" + trip, out, str(entry["kmax"]), str(entry["cap"]), str(entry["pivot"]), "1", str(entry["tol"]), "q:" + qplain]
        else:
            cmd = [drv, "solve", "trip:" + trip, out, str(entry["pivot"]), "1", str(entry["tol"]), "q:" + qplain]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        d = slabfile.load(out)
        ts = float(d["solve_seconds"][0]) if "solve_seconds" in d else None
        return float(d["timing"][0]), ts
    except Exception as e:
        print(f"[bench] reference timing of {name} unavailable: {e}", file=sys.stderr)
        return None, None
    finally:
        for p in (trip, out, base + "_q.slab"):
            if os.path.exists(p):
                os.unlink(p)


def profile_traffic(kernel_ms):
    """HBM bytes per launch from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    passes, reduced by tools/profile_summary.py; counters cannot be read from inside this process) -- but only if that
    profile was taken on the kernel being timed now: its recorded kernel time must agree within 10 %."""
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*", "*_pmc_summary.json")):
        rnd = os.path.basename(os.path.dirname(path))
        tag = os.path.basename(path).split("_")[0]
        try:
            key = (int(rnd[1:]), int(tag[1:]))
        except ValueError:
            continue
        if best is None or key > best[0]:
            best = (key, path)
    if best is None:
        return None, None
    try:
        s = json.load(open(best[1]))
        rec = s.get("kernel_ms_per_launch")
        if rec is None or abs(rec - kernel_ms) > 0.10 * kernel_ms:
            return None, os.path.relpath(best[1], ROOT) + " (stale: recorded kernel time differs from the measured one)"
        return s["hbm_bytes_per_launch"], os.path.relpath(best[1], ROOT)
    except Exception:
        return None, None


def chain_report(fac, K, kernel_ms, info):
    """The bound that actually holds on this workload: the columns commit in order (K hops), and a column
    whose pattern holds an earlier pivot row depends on that column's L (the source DAG).  From the factors just
    downloaded: sources per column = rows of U(:,k) above the pivot with a nonzero value; depth = longest chain of such
    dependencies.  hop_us_measured = kernel time / K; lower_bound_ms = K x the committer's measured serial time per column
    (profiles/r*/..._commit.json of the newest profile, if one is committed) + depth x one cross-chip hand-off (1 us,
    MI355X_MICROARCH.md handoff-1to1)."""
    import numpy as np
    Up, Ui, Ulen, pinv = fac["Up"], fac["Ui"], fac["Ulen"], fac["pinv"]
    depth = np.zeros(K, np.int64); nsrc = 0; ncols_src = 0
    for k in range(K):
        rows = Ui[Up[k]:Up[k + 1] - 1]; lens = Ulen[Up[k]:Up[k + 1] - 1]
        src = pinv[rows][lens != 0]
        if len(src):
            depth[k] = depth[src].max() + 1; nsrc += len(src); ncols_src += 1
    serial_us = None; src_file = None
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*", "*_commit.json")):
        try:
            key = (int(os.path.basename(os.path.dirname(path))[1:]), int(os.path.basename(path).split("_")[0][1:]))
        except ValueError:
            continue
        if best is None or key > best[0]:
            best = (key, path)
    if best:
        try:
            serial_us = json.load(open(best[1])).get("serial_us_per_column"); src_file = os.path.relpath(best[1], ROOT)
        except Exception:
            pass
    out = {"columns_on_chain": int(K), "hop_us_measured": 1e3 * kernel_ms / max(K, 1),
           "source_applications": int(nsrc), "columns_with_sources": int(ncols_src), "source_dag_depth": int(depth.max()) if K else 0,
           "committer_commits": info["committer_commits"], "engine_commits": info["engine_commits"],
           "committer_serial_us_per_column": serial_us, "committer_profile": src_file}
    if serial_us is not None:
        out["lower_bound_ms"] = (K * serial_us + int(depth.max()) * 1.0) * 1e-3
    return out


def farm_mode(args, sl, parallel, torch, dist, rank, world, stream):
    """--farm: the subtree farm of SURVEY 8(e) on a C5-shaped BLOCK matrix (n = 200k, 100 nonzeros per column, made of
    independent diagonal blocks whose columns interleave in the elimination order): every rank factorises its blocks'
    leading columns on its GPU, the blocks' pivot chains are exchanged with one all-gather of GPU tensors (RCCL), the
    scales are formed and every rank rescales its columns on the device.  One step = all of that."""
    import numpy as np
    B, nb, kcols = max(8, world), 25000, 60
    mine = [b for b in range(B) if b % world == rank]
    handles = {}
    for b in mine:
        Ap, Ai, Ax = sl.matgen(nb, 100.0 / nb, 16, 100 + b)
        Alen, Alimbs = sl.ints_to_slab(Ax)
        handles[b] = sl.Factorization(nb, Ap, Ai, Alen, Alimbs, np.arange(nb, dtype=np.int32))
    owner_all = [b for _ in range(kcols) for b in range(B)]          # columns of the blocks interleave round-robin

    def step():
        for f in handles.values():
            f.reset(); f.run(kcols, stream=stream)
        lens, limbs = [], []
        for b in mine:
            rl, rx = handles[b].pivots()
            lens += [int(v) for v in rl]; limbs += [int(v) for v in rx]
        gathered = parallel.allgather_bigints(dist, lens, limbs, device="cuda" if dist is not None else "cpu")
        chains = {}
        for r, (gl, gb) in enumerate(gathered):
            vals = parallel._to_ints(gl, gb)
            for t, b in enumerate([b for b in range(B) if b % world == r]):
                chains[b] = vals[t * kcols:(t + 1) * kcols]
        sigma = parallel.subtree_scales(owner_all, [chains[b] for b in range(B)])
        for b in mine:
            handles[b].rescale([sigma[k] for k in range(len(owner_all)) if owner_all[k] == b])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = parallel.max_over_ranks(dist, time.perf_counter() - t0, device="cuda")
    nnz = sum(f.info()["lnz"] + f.info()["unz"] - f.info()["K"] for f in handles.values())
    kms = sum(f.info()["kernel_ms"] for f in handles.values())
    limbs_out = sum(f.info()["l_limbs"] + f.info()["u_limbs"] for f in handles.values())
    nnz_all = parallel.sum_over_ranks(dist, nnz, device="cuda")
    for f in handles.values():
        f.close()
    if rank == 0:
        print(json.dumps({
            "metric": "L+U nonzeros/sec, subtree farm on a C5-shaped block matrix", "value": nnz_all / (elapsed / args.steps),
            "unit": "L+U nonzeros/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64 limbs (u32 digits on device)", "data": "synthetic",
            "config": {"workload": f"farm: {B} independent blocks of n={nb}, 100 nnz/col, |a|<2^16, first {kcols} columns of every "
                                   "block (interleaved order), pivot chains all-gathered over RCCL, columns rescaled on the device",
                       "columns": B * kcols, "lu_nnz": nnz_all, "parallelism": f"farm x{world}",
                       "rank0_factor_kernel_ms_per_step": kms, "rank0_rescaled_limbs": limbs_out}}))


def replicas_view(sl, w, Ap, Ai, Alen, Alimbs, q, expect, B=8, W=64, reps=3):
    """A throughput view next to the headline, NEVER the headline: B independent copies of the window on B
    handles of this one GPU, each with W workers and a HIP stream of its own, run concurrently from B host threads.  One
    window is a dependency chain and cannot load the chip; independent windows side by side can.  Every copy is checked
    against the reference window's counters."""
    import ctypes as C
    import threading
    hip = C.CDLL("libamdhip64.so")
    hs, streams = [], []
    try:
        for _ in range(B):
            hs.append(sl.Factorization(w["n"], Ap, Ai, Alen, Alimbs, q, limb_cap=w["limb_cap"], workers=W))
            s_ = C.c_void_p()
            if hip.hipStreamCreate(C.byref(s_)) != 0:
                raise RuntimeError("hipStreamCreate")
            streams.append(s_)
        best = None
        for r in range(reps + 1):
            for h in hs:
                h.reset()
            th = [threading.Thread(target=lambda t=t: hs[t].run(0, stream=streams[t].value)) for t in range(B)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt = time.perf_counter() - t0
            infos = [h.info() for h in hs]
            assert all(i["K"] == expect["K"] and i["b_read"] == expect["b_read"] and i["b_write"] == expect["b_write"] for i in infos)
            if r and (best is None or dt < best):
                best = dt
        nnz = expect["nnz"]
        return dict(replicas=B, workers_each=infos[0]["workers"], wall_ms=1e3 * best, lu_nnz_per_s=B * nnz / best,
                    kernel_ms_each=[round(i["kernel_ms"], 3) for i in infos],
                    note="B independent windows on B streams of one GPU, wall clock over all of them; a throughput view, not the metric")
    finally:
        for h in hs:
            h.close()
        for s_ in streams:
            hip.hipStreamDestroy(s_)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--farm", action="store_true", help="the subtree farm on a block matrix instead of the headline workload")
    ap.add_argument("--no-secondary", action="store_true",
                    help="only the headline workload (used under rocprofv3 so that the kernel statistics are the headline's)")
    args = ap.parse_args()

    if os.environ.get("SLIP_HIP_LIBRARY"):
        raise SystemExit("bench.py measures the in-tree HIP library only: unset SLIP_HIP_LIBRARY")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(subprocess.call(relaunch_distributed(args)))

    import numpy as np
    import torch
    import slip_lu_amd as sl
    from slip_lu_amd import parallel, _lib

    rank, world, local = parallel.env_rank()
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = parallel.init("nccl")          # RCCL; None when there is one rank

    if args.farm:
        farm_mode(args, sl, parallel, torch, dist, rank, world, torch.cuda.current_stream().cuda_stream)
        if dist is not None:
            dist.destroy_process_group()
        return

    w = WORKLOAD
    Ap, Ai, Ax = sl.matgen(w["n"], w["density"], w["bits"], w["seed"])
    Alen, Alimbs = sl.ints_to_slab(Ax)
    q = load_q(w["golden"])
    f = sl.Factorization(w["n"], Ap, Ai, Alen, Alimbs, q, limb_cap=w["limb_cap"])
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        f.reset()
        f.run(0, stream=stream)

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    for _ in range(args.steps):
        step()
        kernel_ms += f.info()["kernel_ms"]
    fence()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(dist, elapsed, device="cuda")

    info = f.info()
    K = info["K"]
    nnz = info["lnz"] + info["unz"] - K
    # parity guard: the run being timed must be the reference's window (golden index)
    idx = {e["name"]: e for e in json.load(open(os.path.join(ROOT, "tests", "golden", "index.json")))}[w["golden"]]
    assert K == idx["K"] and nnz == idx["lnz"] + idx["unz"] - idx["K"], "benchmark run differs from the reference window"
    assert info["b_read"] == idx["counters"]["B_read"] and info["b_write"] == idx["counters"]["B_write"]
    # ... and bit for bit: the factors of the LAST timed step against the reference's digest (a wrong pivot with equal counts
    # would otherwise be timed as a pass)
    import slabfile
    fac = f.download()
    assert slabfile.factor_digest(fac) == idx["digest"], "benchmark run: factor digest differs from the reference's"
    chain = chain_report(fac, K, kernel_ms / args.steps, info)

    ms_per_step = 1e3 * elapsed / args.steps
    kms = kernel_ms / args.steps                       # HIP-event time of the column-loop kernel per launch
    traffic, traffic_src = profile_traffic(kms)

    # secondary, complete-run workloads (LP bases / ExampleMats of the reference): one run each, with the
    # compiled reference timed on this host's cores (1 core) in the same run
    secondary = []
    if rank == 0 and world == 1 and not args.no_secondary:
        from conftest import load_case
        for name in SECONDARY:
            try:
                e, fx = load_case(name)
                g = sl.Factorization(e["n"], fx["Ap"], fx["Ai"], fx["Alen"], fx["Alimbs"], fx["q"], limb_cap=e["cap"])
                g.run(e["kmax"]); g.reset(); g.run(e["kmax"])
                gi = g.info()
                nz = gi["lnz"] + gi["unz"] - gi["K"]
                assert nz == e["lnz"] + e["unz"] - e["K"] and gi["b_read"] == e["counters"]["B_read"]
                rec = dict(workload=name, columns=gi["K"], lu_nnz=nz, max_limbs=gi["max_limbs"],
                           kernel_ms=gi["kernel_ms"], launches=gi["launches"], lu_nnz_per_s=nz / (gi["kernel_ms"] * 1e-3),
                           algorithmic_read_GBs=gi["b_read"] / (gi["kernel_ms"] * 1e-3) / 1e9)
                if gi["K"] == e["n"]:
                    # next row of the scope table: SLIP_LU_solve's substitutions on the resident factors, one
                    # right-hand side (the deterministic b of the solve goldens), second run timed
                    b = (np.arange(e["n"], dtype=np.int64) * 2654435761 % (1 << 32)) % 2001 - 1000
                    bl, bx = np.sign(b).astype(np.int32), np.abs(b[b != 0]).astype(np.uint64)
                    g.solve(bl, bx); g.solve(bl, bx)
                    rec["solve_kernel_ms"] = g.solve_ms()
                    # right-hand sides run side by side (one worker each): 16 of them, to be read against 16 x the reference's
                    # one-right-hand-side time below
                    bl16 = np.tile(bl, 16); bx16 = np.tile(bx, 16)
                    g.solve(bl16, bx16, nrhs=16)
                    rec["solve16_kernel_ms"] = g.solve_ms()
                g.close()
                if not args.no_cpu_baseline:
                    tf, ts = reference_seconds(name, e, fx)
                    rec["reference_cpu_seconds"] = tf
                    rec["reference_cpu_solve_seconds"] = ts
                    if tf:
                        rec["gpu_over_cpu"] = tf / (gi["kernel_ms"] * 1e-3)
                secondary.append(rec)
            except Exception as ex:                  # never let a side measurement break the headline
                secondary.append(dict(workload=name, error=str(ex)))

    replicas = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            replicas = replicas_view(sl, w, Ap, Ai, Alen, Alimbs, q, dict(K=K, nnz=nnz, b_read=info["b_read"], b_write=info["b_write"]))
        except Exception as ex:                      # never let a side measurement break the headline
            replicas = dict(error=str(ex))

    value = world * nnz / (elapsed / args.steps)
    achieved = info["b_read"] / (kms * 1e-3) / 1e9     # GB/s, algorithmic reads
    digit_macs = 4 * info["limb_macs"] / (kms * 1e-3)
    out = {
        "metric": "L+U nonzeros/sec on random n=100k 0.1%-dense CSC; achieved HBM GB/s vs peak",
        "value": value, "unit": "L+U nonzeros/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 limbs (u32 digits on device)", "data": "synthetic",
        "config": {"workload": "C4: random CSC n=100000 density=0.001 |a|<2^16 seed=1, COLAMD order (fixture), "
                               "default pivoting, column window until a value exceeds 64 limbs",
                   "columns": K, "lu_nnz": nnz, "n_upd": info["n_upd"], "max_limbs": info["max_limbs"],
                   "parallelism": "replicas" if world > 1 else "1 GPU",
                   "library": os.path.relpath(_lib.library_path(), ROOT), "version": f.lib.slip_hip_version().decode()},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "slip_factor_kernel", "kernel_ms_per_launch": kms, "workers": info["workers"], "waves": info["waves"],
                     "lds_bytes_per_worker": info["lds_bytes"],
                     "algorithmic_read_bytes": info["b_read"], "algorithmic_write_bytes": info["b_write"],
                     "achieved_read_plus_write": (info["b_read"] + info["b_write"]) / (kms * 1e-3) / 1e9,
                     "alu": {"limb_macs": info["limb_macs"], "digit_macs_per_s": digit_macs, "peak": DIGIT_MAC_PEAK,
                             "frac": digit_macs / DIGIT_MAC_PEAK,
                             "note": "algorithmic: 4 x (l(L_m) l(x_j) + l(x_i) l(rho_jn)) 32-bit multiply-adds per IPGE update "
                                     "(SURVEY 8(d)); peak = v_mad_u64_u32 at a quarter of the VALU lane rate"},
                     "chain": chain},
    }
    if secondary:
        out["secondary"] = secondary
    if replicas:
        out["replicas_on_one_gpu"] = replicas
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(K, nnz)
    f.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
