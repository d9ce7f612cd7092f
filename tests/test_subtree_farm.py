"""SURVEY 8(e) known-answer test of the subtree farm's scaling law: a matrix made of independent diagonal
blocks (rows and columns interleaved) factorised (i) whole and (ii) block by block + slip_lu_amd.parallel's
assembly (pivot chains exchanged, columns rescaled) must agree exactly.  The factoriser in this CPU test is
the oracle; tests/test_gpu_parity.py::test_gpu_subtree_farm_law runs the same check with the HIP path."""
import os
import subprocess
import sys
import textwrap

import numpy as np

import oracle_lib
from conftest import ROOT
from slip_lu_amd import parallel


def make_blocked(sizes, seed, bits=10, density=0.6):
    """block-diagonal integer matrix, rows and columns interleaved by the same permutation"""
    rng = np.random.default_rng(seed)
    n = sum(sizes)
    ids = rng.permutation(n)
    cols = [[] for _ in range(n)]
    o = 0
    for s in sizes:
        blk = ids[o:o + s]; o += s
        for a in range(s):
            for b in range(s):
                if a == b or rng.random() < density:
                    v = int(rng.integers(1, 2 ** bits)) * (1 if rng.random() < 0.5 else -1)
                    cols[int(blk[b])].append((int(blk[a]), v))
    Ap, Ai, Ax = [0], [], []
    for j in range(n):
        rng.shuffle(cols[j])
        for i, v in cols[j]:
            Ai.append(i); Ax.append(v)
        Ap.append(len(Ai))
    return n, np.array(Ap, np.int64), np.array(Ai, np.int32), np.array(Ax, np.int64)


def as_columns(r):
    """canonical factor dict -> dict(rho, piv_row, L, U) with {row: value} columns (ORIGINAL row ids)"""
    n = r["n"]
    Lx = oracle_lib.bigints(r["Llen"], r["Llimbs"]); Ux = oracle_lib.bigints(r["Ulen"], r["Ulimbs"])
    rho = oracle_lib.bigints(r["rholen"], r["rholimbs"])
    inv = np.argsort(r["pinv"])                 # position -> row
    L = [{int(r["Li"][p]): Lx[p] for p in range(r["Lp"][k], r["Lp"][k + 1])} for k in range(n)]
    U = [{int(r["Ui"][p]): Ux[p] for p in range(r["Up"][k], r["Up"][k + 1])} for k in range(n)]
    return dict(rho=rho, piv_row=[int(inv[k]) for k in range(n)], L=L, U=U)


def check_farm(factor, sizes=(3, 5, 4), seed=7):
    n, Ap, Ai, Ax = make_blocked(sizes, seed)
    q = np.random.default_rng(seed + 1).permutation(n).astype(np.int32)
    whole = as_columns(factor(n, Ap, Ai, Ax, q))
    blocks = parallel.diagonal_blocks(n, Ap, Ai)
    assert blocks is not None and sorted(map(len, blocks)) == sorted(sizes)
    local = []
    for ids in blocks:
        bp, bi, bx, bq = parallel.extract_block(ids, Ap, Ai, Ax, q)
        local.append(as_columns(factor(len(ids), bp, bi, np.array(bx, np.int64), bq)))
    got = parallel.assemble_blocks(blocks, q, local)
    assert got["rho"] == whole["rho"]
    assert got["piv_row"] == whole["piv_row"]
    assert got["L"] == whole["L"] and got["U"] == whole["U"]
    # the law is not vacuous: some scale differs from 1
    owner = [next(t for t, ids in enumerate(blocks) if int(c) in ids) for c in q]
    assert any(s != 1 for s in parallel.subtree_scales(owner, [l["rho"] for l in local]))


def oracle_factor(n, Ap, Ai, Ax, q):
    r = oracle_lib.factorize(n, Ap, Ai, np.sign(Ax).astype(np.int32), np.abs(Ax).astype(np.uint64), q)
    assert r["status"] == 0 and r["K"] == n
    return r


def test_subtree_scaling_law_on_oracle():
    check_farm(oracle_factor)
    check_farm(oracle_factor, sizes=(6, 2, 7, 1), seed=19)


def test_survey_known_answer():
    """two 3x3 blocks with local pivot chains a and b: diag(B1,B2) has rho = a0,a1,a2, b0*a2, b1*a2, b2*a2"""
    sig = parallel.subtree_scales([0, 0, 0, 1, 1, 1], [[-4, 24, 382], [-3, -42, -1165]])
    assert [r * s for r, s in zip([-4, 24, 382, -3, -42, -1165], sig)] == [-4, 24, 382, -1146, -16044, -445030]
    sig = parallel.subtree_scales([0, 1, 0, 1, 0, 1], [[-4, 24, 382], [-3, -42, -1165]])     # interleaved
    assert [r * s for r, s in zip([-4, -3, 24, -42, 382, -1165], sig)] == [-4, 12, -72, -1008, -16044, -445030]


WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    from slip_lu_amd import parallel
    import oracle_lib, test_subtree_farm as T
    dist = parallel.init("gloo")
    rank, world, _ = parallel.env_rank()
    n, Ap, Ai, Ax = T.make_blocked((4, 6, 3, 5), 31)
    q = np.random.default_rng(32).permutation(n).astype(np.int32)
    blocks, rest = parallel.leading_blocks(n, Ap, Ai, q, n)          # components of the leading columns' graph
    assert rest == [] and blocks == parallel.diagonal_blocks(n, Ap, Ai)
    bins = parallel.lpt_partition([len(b) ** 3 for b in blocks], world)
    mine = {{}}
    for t in bins[rank]:
        bp, bi, bx, bq = parallel.extract_block(blocks[t], Ap, Ai, Ax, q)
        mine[t] = T.as_columns(T.oracle_factor(len(blocks[t]), bp, bi, np.array(bx, np.int64), bq))
    # the exchange: every rank's pivot chains, packed big integers, one all-gather
    lens, limbs, tags = [], [], []
    for t in sorted(mine):
        for v in mine[t]["rho"]:
            a = abs(v); l = 0
            while a:
                limbs.append(a & (2 ** 64 - 1)); a >>= 64; l += 1
            lens.append(-l if v < 0 else l)
        tags.append((t, len(mine[t]["rho"])))
    gathered = parallel.allgather_bigints(dist, lens, limbs)
    chains = {{}}
    for r, (gl, gb) in enumerate(gathered):
        vals = oracle_lib.bigints(gl, gb); o = 0
        for t in sorted(bins[r]):
            chains[t] = vals[o:o + len(blocks[t])]; o += len(blocks[t])
    block_of = {{g: t for t, ids in enumerate(blocks) for g in ids}}
    owner = [block_of[int(c)] for c in q]
    sigma = parallel.subtree_scales(owner, [chains[t] for t in range(len(blocks))])
    # each rank rescales ITS columns; rank 0 checks them against the whole factorisation
    whole = T.as_columns(T.oracle_factor(n, Ap, Ai, Ax, q))
    done = [0] * len(blocks); ok = 1
    for k, t in enumerate(owner):
        kl = done[t]; done[t] += 1
        if t in mine:
            ok &= mine[t]["rho"][kl] * sigma[k] == whole["rho"][k]
            ok &= {{blocks[t][r]: v * sigma[k] for r, v in mine[t]["L"][kl].items()}} == whole["L"][k]
    total = parallel.sum_over_ranks(dist, int(ok))
    if rank == 0:
        print(json.dumps(dict(ok=total, world=world)))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
''')


def test_farm_exchange_two_ranks_gloo():
    path = os.path.join("/tmp", f"slip_farm_worker_{os.getpid()}.py")
    open(path, "w").write(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", path],
                         capture_output=True, text=True, env=env, timeout=600)
    os.unlink(path)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res == dict(ok=2, world=2)


def test_leading_blocks_detection():
    """components of the LEADING columns only: a prefix of the order may decouple where the whole matrix does not"""
    n, Ap, Ai, Ax = make_blocked((4, 6, 3, 5), 31)
    q = np.random.default_rng(32).permutation(n).astype(np.int32)
    blocks, rest = parallel.leading_blocks(n, Ap, Ai, q, n)
    assert blocks == parallel.diagonal_blocks(n, Ap, Ai) and rest == []
    # couple everything through one extra dense LAST column: the whole matrix is one component, its leading n-1 columns are not
    last = int(q[-1])
    cols = [[(int(Ai[p]), int(Ax[p])) for p in range(Ap[j], Ap[j + 1])] for j in range(n)]
    cols[last] = [(i, 1 + i) for i in range(n)]
    Ap2, Ai2 = [0], []
    for j in range(n):
        Ai2 += [i for i, _ in cols[j]]; Ap2.append(len(Ai2))
    Ap2, Ai2 = np.array(Ap2, np.int64), np.array(Ai2, np.int32)
    assert parallel.diagonal_blocks(n, Ap2, Ai2) in (None, [sorted(range(n))])
    lead, rest = parallel.leading_blocks(n, Ap2, Ai2, q, n - 1)
    assert rest and rest[-1] == last
    assert all(set(b) <= set(range(n)) - {last} for b in lead) and len(lead) >= 2


def farm_vs_whole(factor_whole, make, sizes=(3, 5, 4), seed=7, kcols=0, **kw):
    """slip_lu_amd.parallel.farm_factorize (blocks on the device, pivot chains exchanged, columns rescaled ON THE DEVICE)
    against the factorisation of the whole matrix: same pivots, same values entry by entry"""
    n, Ap, Ai, Ax = make_blocked(sizes, seed)
    q = np.random.default_rng(seed + 1).permutation(n).astype(np.int32)
    whole = as_columns(factor_whole(n, Ap, Ai, Ax, q))
    blocks, rest = parallel.leading_blocks(n, Ap, Ai, q, n)
    assert rest == []
    mine, owner, sigma = parallel.farm_factorize(None, n, Ap, Ai, Ax, q, blocks, kcols=kcols, make=make, **kw)
    assert any(s != 1 for s in sigma)
    done = [0] * len(blocks)
    for k, t in enumerate(owner):
        kl = done[t]; done[t] += 1
        loc = as_columns_partial(mine[t])
        ids = blocks[t]
        assert loc["rho"][kl] == whole["rho"][k], k
        assert {ids[r]: v for r, v in loc["L"][kl].items()} == whole["L"][k], k
        assert {ids[r]: v for r, v in loc["U"][kl].items()} == whole["U"][k], k


def as_columns_partial(r):
    K = r["K"]
    Lx = oracle_lib.bigints(r["Llen"], r["Llimbs"]); Ux = oracle_lib.bigints(r["Ulen"], r["Ulimbs"])
    rho = oracle_lib.bigints(r["rholen"], r["rholimbs"])
    L = [{int(r["Li"][p]): Lx[p] for p in range(r["Lp"][k], r["Lp"][k + 1])} for k in range(K)]
    U = [{int(r["Ui"][p]): Ux[p] for p in range(r["Up"][k], r["Up"][k + 1])} for k in range(K)]
    return dict(rho=rho, L=L, U=U)


def test_farm_device_path_on_emulator():
    """the farm's device path (factorise blocks, rescale on the device) with the CPU emulation build of the kernel source"""
    import slip_lu_amd as sl
    emu = os.path.join(ROOT, "tests", "emu", "libslip_emu.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu.so"])

    def make(n, Ap, Ai, Alen, Alimbs, q, **kw):
        return sl.Factorization(n, Ap, Ai, Alen, Alimbs, q, lib_path=emu, waves=1, workers=2, **kw)
    farm_vs_whole(oracle_factor, make)


def test_bench_builds_the_torchrun_command():
    """bench.py --gpus 2 without a torchrun environment starts one rank per GPU as a child process"""
    sys.path.insert(0, ROOT)
    import bench
    args = type("A", (), dict(gpus=2, steps=3, warmup=1, no_cpu_baseline=True, no_secondary=True))()
    cmd = bench.relaunch_distributed(args)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--gpus") + 1] == "2" and "--no-cpu-baseline" in cmd and os.path.basename(cmd[cmd.index("--gpus") - 1]) == "bench.py"


WORKER_FARM = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    import slip_lu_amd as sl
    from slip_lu_amd import parallel
    import test_subtree_farm as T
    dist = parallel.init("gloo")
    rank, world, _ = parallel.env_rank()
    emu = os.path.join({root!r}, "tests", "emu", "libslip_emu.so")
    n, Ap, Ai, Ax = T.make_blocked((4, 6, 3, 5, 7), 41)
    q = np.random.default_rng(42).permutation(n).astype(np.int32)
    blocks, rest = parallel.leading_blocks(n, Ap, Ai, q, n)
    assert rest == []

    def make(n_, Ap_, Ai_, Alen_, Alimbs_, q_, **kw):
        return sl.Factorization(n_, Ap_, Ai_, Alen_, Alimbs_, q_, lib_path=emu, waves=1, workers=2, **kw)
    # THE function under test, with two ranks: bins[rank], the count gather, done_cols, the device rescale
    mine, owner, sigma = parallel.farm_factorize(dist, n, Ap, Ai, Ax, q, blocks, make=make)
    whole = T.as_columns(T.oracle_factor(n, Ap, Ai, Ax, q))
    assert len(owner) == n and any(s != 1 for s in sigma)
    done = [0] * len(blocks); ok = 1; checked = 0
    for k, t in enumerate(owner):
        kl = done[t]; done[t] += 1
        if t in mine:
            loc = T.as_columns_partial(mine[t]); ids = blocks[t]
            ok &= loc["rho"][kl] == whole["rho"][k]
            ok &= {{ids[r]: v for r, v in loc["L"][kl].items()}} == whole["L"][k]
            ok &= {{ids[r]: v for r, v in loc["U"][kl].items()}} == whole["U"][k]
            checked += 1
    tot_ok = parallel.sum_over_ranks(dist, int(ok)); tot_cols = parallel.sum_over_ranks(dist, checked)
    nmine = parallel.sum_over_ranks(dist, len(mine))
    if rank == 0:
        print(json.dumps(dict(ok=tot_ok, world=world, cols=tot_cols, blocks=nmine, mine0=len(mine))))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
''')


def test_farm_factorize_two_ranks_gloo():
    """parallel.farm_factorize ITSELF under torch.distributed with two ranks (gloo; the emulator build stands in for the GPU):
    every rank factorises its bin of blocks, the pivot chains and the column counts are all-gathered, every rank rescales
    its columns on the 'device' -- each column of the whole matrix is checked by exactly one rank"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu.so"])
    path = os.path.join("/tmp", f"slip_farm2_worker_{os.getpid()}.py")
    open(path, "w").write(WORKER_FARM.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", path],
                         capture_output=True, text=True, env=env, timeout=900)
    os.unlink(path)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] == 2 and res["world"] == 2 and res["cols"] == 25 and res["blocks"] == 5 and 0 < res["mine0"] < 5, res


# ---------------------------------------------------------------------------------------------
# Farm completion: blocks on their ranks, gathered, then the separator columns on top of them
# ---------------------------------------------------------------------------------------------
def make_bordered(sizes, border, seed, bits=8, density=0.6):
    """bordered block-diagonal integer matrix: independent diagonal blocks (ids interleaved) plus `border` separator columns
    with entries in every block's rows (and a dense border-by-border corner); q eliminates the blocks' columns first (in a
    random interleaving), the border last"""
    rng = np.random.default_rng(seed)
    n0 = sum(sizes); n = n0 + border
    ids = rng.permutation(n)
    cols = [[] for _ in range(n)]
    val = lambda: int(rng.integers(1, 2 ** bits)) * (1 if rng.random() < 0.5 else -1)
    o = 0
    for s in sizes:
        blk = ids[o:o + s]; o += s
        for a in range(s):
            for b in range(s):
                if a == b or rng.random() < density:
                    cols[int(blk[b])].append((int(blk[a]), val()))
    bord = [int(v) for v in ids[n0:]]
    inner = [int(v) for v in ids[:n0]]
    for b in bord:
        for i in inner:
            if rng.random() < 0.5:
                cols[b].append((i, val()))          # border column, block row (a block column with a border row could take its
                                                    # pivot there: such a component is not farmable, leading_blocks leaves it out)
        for b2 in bord:
            if b2 == b or rng.random() < 0.7:
                cols[b].append((b2, val()))
    Ap, Ai, Ax = [0], [], []
    for j in range(n):
        rng.shuffle(cols[j])
        for i, v in cols[j]:
            Ai.append(i); Ax.append(v)
        Ap.append(len(Ai))
    q = np.array([inner[t] for t in rng.permutation(n0)] + [bord[t] for t in rng.permutation(border)], np.int32)
    return n, np.array(Ap, np.int64), np.array(Ai, np.int32), np.array(Ax, np.int64), q, n0


def same_factors(a, b):
    import slabfile
    for k in slabfile.FACTOR_KEYS:
        if k in a and k in b:
            assert np.array_equal(np.asarray(a[k]).astype(np.int64), np.asarray(b[k]).astype(np.int64)), k
    assert a["K"] == b["K"]


def test_prefix_assembly_reproduces_the_reference_order():
    """assemble_prefix's entry order (global positions replayed from the pivot rows) on a block-diagonal matrix: the assembled
    columns are the whole-matrix oracle's arrays word for word -- the form slip_hip_factor_set_prefix takes"""
    n, Ap, Ai, Ax = make_blocked((3, 5, 4, 6), 23)
    q = np.random.default_rng(24).permutation(n).astype(np.int32)
    whole = oracle_factor(n, Ap, Ai, Ax, q)
    blocks, rest = parallel.leading_blocks(n, Ap, Ai, q, n)
    assert rest == []

    class OracleHandle:                 # the oracle behind the handle interface farm_factorize drives
        def __init__(self, n_, Ap_, Ai_, Alen_, Alimbs_, q_, **kw):
            self.args = (n_, Ap_, Ai_, Alen_, Alimbs_, q_)
        def run(self, kmax=0):
            self.r = oracle_lib.factorize(*self.args)
        def download(self):
            return self.r
        def rescale(self, scales):
            r = self.r
            Lx = oracle_lib.bigints(r["Llen"], r["Llimbs"]); Ux = oracle_lib.bigints(r["Ulen"], r["Ulimbs"])
            pinv = r["pinv"]
            Lx = [v * scales[k] for k in range(r["K"]) for v in Lx[r["Lp"][k]:r["Lp"][k + 1]]]
            Ux = [Ux[p] * scales[int(pinv[int(r["Ui"][p])])] for k in range(r["K"]) for p in range(r["Up"][k], r["Up"][k + 1])]
            for key, vals in (("L", Lx), ("U", Ux)):
                lens, limbs = [], []
                for v in vals:
                    a = abs(v); l = []
                    while a:
                        l.append(a & (2 ** 64 - 1)); a >>= 64
                    lens.append(len(l) if v >= 0 else -len(l)); limbs += l
                r[key + "len"] = np.array(lens, np.int32); r[key + "limbs"] = np.array(limbs, np.uint64)
        def close(self):
            pass
    mine, owner, sigma = parallel.farm_factorize(None, n, Ap, Ai, Ax, q, blocks, make=OracleHandle)
    facs = parallel.unpack_factors(parallel.pack_factors(mine))           # through the wire format
    fac, piv_row = parallel.assemble_prefix(n, blocks, [int(c) for c in q], facs, owner)
    assert [int(v) for v in piv_row] == [int(v) for v in np.argsort(whole["pinv"])]
    for k in ("Lp", "Li", "Llen", "Llimbs", "Up", "Ui", "Ulen", "Ulimbs"):
        assert np.array_equal(np.asarray(fac[k]).astype(np.int64), np.asarray(whole[k])[:len(fac[k])].astype(np.int64)), k


def test_farm_complete_on_emulator():
    """blocks -> rescale -> gather -> slip_hip_factor_set_prefix -> the separator columns, with the CPU emulation build of the
    kernel source: equal to the oracle's factorisation of the whole bordered matrix"""
    import slip_lu_amd as sl
    emu = os.path.join(ROOT, "tests", "emu", "libslip_emu.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu.so"])

    def make(n, Ap, Ai, Alen, Alimbs, q, **kw):
        return sl.Factorization(n, Ap, Ai, Alen, Alimbs, q, lib_path=emu, waves=2, workers=3, **kw)
    n, Ap, Ai, Ax, q, n0 = make_bordered((4, 6, 5), 4, 77)
    whole = oracle_factor(n, Ap, Ai, Ax, q)
    got = parallel.farm_complete(None, n, Ap, Ai, Ax, q, n0, make=make)
    assert got["farm_prefix"] == n0
    same_factors(got, whole)
    # a leading window that cuts a component: the farm takes what is farmable and the handle does the rest
    got = parallel.farm_complete(None, n, Ap, Ai, Ax, q, n0 - 1, make=make)
    assert 0 <= got["farm_prefix"] < n0
    same_factors(got, whole)


WORKER_COMPLETE = textwrap.dedent('''
    import json, os, sys
    import numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import slip_lu_amd as sl
    from slip_lu_amd import parallel
    import test_subtree_farm as T
    dist = parallel.init("gloo")
    rank, world, _ = parallel.env_rank()
    emu = os.path.join({root!r}, "tests", "emu", "libslip_emu.so")
    n, Ap, Ai, Ax, q, n0 = T.make_bordered((5, 3, 6, 4), 3, 91)

    def make(n_, Ap_, Ai_, Alen_, Alimbs_, q_, **kw):
        return sl.Factorization(n_, Ap_, Ai_, Alen_, Alimbs_, q_, lib_path=emu, waves=1, workers=2, **kw)
    got = parallel.farm_complete(dist, n, Ap, Ai, Ax, q, n0, make=make, finish_on=0)
    ok = 1
    if rank == 0:
        whole = T.oracle_factor(n, Ap, Ai, Ax, q)
        T.same_factors(got, whole)
        print(json.dumps(dict(ok=1, world=world, prefix=got["farm_prefix"], K=got["K"])))
    else:
        assert got is None
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
''')


def test_farm_complete_two_ranks_gloo():
    """the farm end to end with two ranks (gloo; the emulator build stands in for the GPU): each rank factorises its blocks,
    the rescaled columns are all-gathered, rank 0 continues with the separator columns -- equal to the whole-matrix oracle"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "emu"), "libslip_emu.so"])
    path = os.path.join("/tmp", f"slip_farm3_worker_{os.getpid()}.py")
    open(path, "w").write(WORKER_COMPLETE.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29549", path],
                         capture_output=True, text=True, env=env, timeout=900)
    os.unlink(path)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["ok"] == 1 and res["world"] == 2 and res["prefix"] == 18 and res["K"] == 21, res
