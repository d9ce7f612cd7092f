"""The N>1 plumbing (slip_lu_amd/parallel.py) under gloo, world_size 2, on CPU: sharding of independent
factorisations, the max-over-ranks timing reduction and the packed big-integer all-gather used for pivot
chains.  The factorisation backend in this CPU test is the oracle (the HIP path needs a GPU)."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    from slip_lu_amd import parallel
    import oracle_lib
    dist = parallel.init("gloo")
    rank, world, _ = parallel.env_rank()
    # four independent small factorisations (seeds), sharded over the ranks
    seeds = parallel.shard([21, 22, 23, 24], rank, world)
    nnz, pivs_len, pivs_limbs = 0, [], []
    for s in seeds:
        Ap, Ai, Ax = oracle_lib.matgen(30, 0.2, 12, s)
        r = oracle_lib.factorize(30, Ap, Ai, np.sign(Ax).astype(np.int32), np.abs(Ax).astype(np.uint64),
                                 np.arange(30, dtype=np.int32))
        assert r["status"] == 0
        nnz += len(r["Li"]) + len(r["Ui"]) - r["K"]
        # last pivot (= determinant up to sign) of each factorisation: what a subtree farm exchanges
        l = int(abs(r["rholen"][-1])); off = int(np.abs(r["rholen"][:-1]).sum())
        pivs_len.append(int(r["rholen"][-1])); pivs_limbs += list(r["rholimbs"][off:off + l])
    total = parallel.sum_over_ranks(dist, nnz)
    tmax = parallel.max_over_ranks(dist, 1.0 + rank)
    gathered = parallel.allgather_bigints(dist, pivs_len, pivs_limbs)
    if rank == 0:
        print(json.dumps(dict(total=total, tmax=tmax, lens=[g[0].tolist() for g in gathered],
                              limbs=[[int(x) for x in g[1]] for g in gathered])))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()
''')


def _run(world):
    code = WORKER.format(root=ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", "29533", "-c", code] if world > 1 else [sys.executable, "-c", code]
    if world > 1:
        # torch.distributed.run wants a script path
        path = os.path.join("/tmp", f"slip_parallel_worker_{os.getpid()}.py")
        open(path, "w").write(code)
        cmd = cmd[:-2] + [path]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_two_ranks_equal_one_rank():
    one, two = _run(1), _run(2)
    assert one["total"] == two["total"]                 # all four factorisations done exactly once
    assert two["tmax"] == 2.0 and one["tmax"] == 1.0    # MAX over ranks
    # rank-ordered gather holds every factorisation's last pivot: same multiset as the 1-rank run
    flat = lambda d: sorted(zip([x for l in d["lens"] for x in l], map(tuple, _split(d))))
    assert flat(one) == flat(two)


def _split(d):
    out = []
    for lens, limbs in zip(d["lens"], d["limbs"]):
        o = 0
        for l in lens:
            out.append(limbs[o:o + abs(l)]); o += abs(l)
    return out


def test_lpt_partition_balances():
    from slip_lu_amd import parallel
    bins = parallel.lpt_partition([9, 7, 6, 5, 4, 3], 2)
    loads = [sum([9, 7, 6, 5, 4, 3][t] for t in b) for b in bins]
    assert sorted(sum(bins, [])) == list(range(6)) and abs(loads[0] - loads[1]) <= 1
