"""Multi-GPU plumbing for the hot path: one process per GPU (torch.distributed; "nccl" is RCCL on ROCm).

The path shards only over INDEPENDENT factorisations (separate matrices, or the diagonal blocks /
elimination-tree subtrees of one matrix -- DESIGN.md section 7): there is no data-path collective
inside a factorisation.  What ranks exchange is tiny: a barrier, the max of the step time, and (for
the subtree farm) the packed pivot chains.  The helpers here are backend-agnostic so the same code is
exercised with "gloo" on CPU in tests/test_parallel.py.
"""
import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend):
    """Initialise torch.distributed from the torchrun environment; returns the module or None (1 rank)."""
    rank, world, _ = env_rank()
    if world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def shard(items, rank, world):
    """Static round-robin assignment of independent work units (matrices / blocks) to ranks."""
    return [it for t, it in enumerate(items) if t % world == rank]


def lpt_partition(weights, world):
    """Longest-processing-time bin packing of predicted work: returns bins[rank] = indices."""
    order = sorted(range(len(weights)), key=lambda t: -weights[t])
    load = [0.0] * world
    bins = [[] for _ in range(world)]
    for t in order:
        r = min(range(world), key=lambda x: load[x])
        bins[r].append(t)
        load[r] += weights[t]
    return bins


def max_over_ranks(dist, value, device="cpu"):
    """MAX all-reduce of a python float (the benchmark's step time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def pack_bigints(lens, limbs):
    """[count][signed limb counts][limbs] as one int64 buffer -- the wire format of pivot chains."""
    lens = np.asarray(lens, dtype=np.int64)
    limbs = np.asarray(limbs, dtype=np.uint64).view(np.int64)
    return np.concatenate([np.array([lens.size], dtype=np.int64), lens, limbs])


def unpack_bigints(buf):
    buf = np.asarray(buf, dtype=np.int64)
    cnt = int(buf[0])
    lens = buf[1:1 + cnt].astype(np.int32)
    nl = int(np.abs(lens).sum())
    limbs = buf[1 + cnt:1 + cnt + nl].view(np.uint64)
    return lens, limbs


def allgather_bigints(dist, lens, limbs, device="cpu"):
    """All-gather variable-length big-integer lists (one list per rank): returns [(lens, limbs)] by rank."""
    mine = pack_bigints(lens, limbs)
    if dist is None:
        return [unpack_bigints(mine)]
    import torch
    world = dist.get_world_size()
    size = torch.tensor([mine.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size)
    cap = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros(cap, dtype=torch.int64, device=device)
    pad[:mine.size] = torch.from_numpy(mine).to(device)
    outs = [torch.zeros(cap, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [unpack_bigints(o.cpu().numpy()[:int(s.item())]) for o, s in zip(outs, sizes)]
