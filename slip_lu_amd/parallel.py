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


# ---------------------------------------------------------------------------------------------
# Subtree farm: independent diagonal blocks of one matrix on different GPUs.
# Host logic only -- index work and python integers; the factorisations themselves are the HIP path.
# ---------------------------------------------------------------------------------------------
def diagonal_blocks(n, Ap, Ai):
    """Connected components of the row/column graph of A: blocks[t] = sorted ids of block t, or None when
    some component's row-id set differs from its column-id set (the reference's "diagonal" pivot test
    reads row id == column id, slip_get_pivot.c:96, so only symmetric id sets can be renumbered locally)."""
    parent = list(range(2 * n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for j in range(n):
        for p in range(int(Ap[j]), int(Ap[j + 1])):
            a, b = find(n + j), find(int(Ai[p]))
            if a != b:
                parent[a] = b
    rows, cols = {}, {}
    for i in range(n):
        rows.setdefault(find(i), []).append(i)
        cols.setdefault(find(n + i), []).append(i)
    blocks = []
    for root, cs in cols.items():
        if rows.get(root) != cs:
            return None
        blocks.append(cs)
    if sum(len(b) for b in blocks) != n:
        return None
    return sorted(blocks)


def extract_block(ids, Ap, Ai, Ax, q):
    """The block's own CSC (ids renumbered by rank, entry order kept) and its column order (q restricted)."""
    loc = {g: t for t, g in enumerate(ids)}
    bp, bi, bx = [0], [], []
    for g in ids:
        for p in range(int(Ap[g]), int(Ap[g + 1])):
            bi.append(loc[int(Ai[p])]); bx.append(Ax[p])
        bp.append(len(bi))
    bq = [loc[int(c)] for c in q if int(c) in loc]
    return (np.array(bp, np.int64), np.array(bi, np.int32), bx, np.array(bq, np.int32))


def subtree_scales(owner, local_rhos):
    """sigma[k] for every global column k (in elimination order): the product, over the OTHER blocks, of the
    last local pivot each of them produced among the global columns < k (1 if none).  owner[k] = block of
    global column k; local_rhos[t] = that block's pivot chain as python ints.
    Then  rho[k] = rho_T[k_local] * sigma[k],  L(:,k) = L_T(:,k_local) * sigma[k],  and an entry of U in the
    row whose pivot sits at global position p is  U_T * sigma[p]  (SURVEY 8(e))."""
    nb = len(local_rhos)
    last = [1] * nb
    done = [0] * nb
    sigma = []
    for t in owner:
        s = 1
        for u in range(nb):
            if u != t:
                s *= last[u]
        sigma.append(s)
        last[t] = local_rhos[t][done[t]]
        done[t] += 1
    return sigma


def assemble_blocks(blocks, q, local):
    """Recombine per-block factorisations into the global one.  local[t] = dict(rho=[ints], piv_row=[local row id
    of the pivot of local column k], L=[{local row: int}], U=[{local row: int}]) in local elimination order.
    Returns dict(rho, piv_row, L, U) with global row ids, columns in global elimination order, entries as
    {row: value} maps (the entry ORDER inside a column follows the global pinv history and is re-derived by
    the caller from piv_row; values and patterns are what this function reconstructs)."""
    block_of = {}
    for t, ids in enumerate(blocks):
        for g in ids:
            block_of[g] = t
    owner = [block_of[int(c)] for c in q]
    sigma = subtree_scales(owner, [l["rho"] for l in local])
    done = [0] * len(blocks)
    pos_of_row = {}
    out = dict(rho=[], piv_row=[], L=[], U=[])
    for k, t in enumerate(owner):
        kl = done[t]; done[t] += 1
        ids, l = blocks[t], local[t]
        prow = ids[l["piv_row"][kl]]
        pos_of_row[prow] = k
        out["rho"].append(l["rho"][kl] * sigma[k])
        out["piv_row"].append(prow)
        out["L"].append({ids[r]: v * sigma[k] for r, v in l["L"][kl].items()})
        out["U"].append({ids[r]: v * sigma[pos_of_row[ids[r]]] for r, v in l["U"][kl].items()})
    return out


def leading_blocks(n, Ap, Ai, q, t):
    """Independent subtrees of the LEADING t columns of the elimination order (the column elimination tree is
    not given by the reference -- SLIP_LU_analyze returns only q -- so the subtrees are found as the connected components of
    the row/column graph of columns q[0:t]).  A component can be factorised on its own, with the block's local pivot chain,
    when its row-id set equals its column-id set (the reference's diagonal pivot test reads row id == column id,
    slip_get_pivot.c:96).  Returns (blocks, rest): blocks[b] = sorted ids of a farmable component, rest = the columns of
    q that belong to no such component (the top separator: they stay with rank 0), in elimination order.
    t = n on a block-diagonal matrix gives diagonal_blocks()."""
    q = [int(c) for c in q]
    lead = q[:t]
    parent = {}

    def find(a):
        while parent.setdefault(a, a) != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for j in lead:
        for p in range(int(Ap[j]), int(Ap[j + 1])):
            a, b = find(("c", j)), find(("r", int(Ai[p])))
            if a != b:
                parent[a] = b
    comps = {}
    for j in lead:
        comps.setdefault(find(("c", j)), [set(), set()])[0].add(j)
    for key in list(parent):
        if key[0] == "r":
            root = find(key)
            if root in comps:
                comps[root][1].add(key[1])
    blocks = sorted(sorted(cs) for cs, rs in comps.values() if cs == rs)
    inblock = set(g for b in blocks for g in b)
    rest = [c for c in q if c not in inblock]
    return blocks, rest


def farm_factorize(dist, n, Ap, Ai, Ax, q, blocks, kcols=0, device="cpu", make=None, **kw):
    """The subtree farm's data path on this rank (one process per GPU): factorise this rank's share of the independent
    blocks on the device, exchange the blocks' pivot chains with ONE all-gather (RCCL on `device`="cuda" tensors under the
    nccl backend; gloo on CPU tensors in the tests), form the scales, rescale this rank's columns on the device
    (slip_hip_factor_rescale) and download them.  kcols > 0: only the first kcols columns of every block (a column window).
    Returns (mine, owner, sigma): mine[t] = canonical factor dict of block t (values already global), owner[k] = block of
    the k-th committed global column, sigma[k] its scale."""
    import slip_lu_amd as sl
    rank, world, _ = env_rank()
    bins = lpt_partition([len(b) ** 2 for b in blocks], world)
    handles, ncols = {}, {}
    for t in bins[rank]:
        ids = blocks[t]
        bp, bi, bx, bq = extract_block(ids, Ap, Ai, Ax, q)
        alen, alimbs = sl.ints_to_slab(np.array(bx, dtype=np.int64))
        f = (make or sl.Factorization)(len(ids), bp, bi, alen, alimbs, bq, **kw)
        f.run(kcols)
        handles[t] = f
    # the exchange: every rank's pivot chains (packed big integers), tagged by the columns each block committed
    lens, limbs, counts = [], [], []
    for t in sorted(handles):
        d = handles[t].download()
        lens += [int(v) for v in d["rholen"]]; limbs += [int(v) for v in d["rholimbs"]]
        counts.append(d["K"])
    # (the column counts travel in a second, tiny gather so that the packed format stays "big integers only")
    cnt = allgather_bigints(dist, [1 if c else 0 for c in counts], [c for c in counts if c], device=device)
    chains_g = allgather_bigints(dist, lens, limbs, device=device)
    chains, done_cols = {}, {}
    for r in range(len(chains_g)):
        vals = _to_ints(*chains_g[r])
        cvals = _to_ints(*cnt[r])
        o = 0
        for t, c in zip(sorted(bins[r]), cvals):
            chains[t] = vals[o:o + c]; done_cols[t] = c; o += c
    block_of = {g: t for t, ids in enumerate(blocks) for g in ids}
    seen = [0] * len(blocks)
    owner = []
    for c in q:
        t = block_of.get(int(c))
        if t is None:
            continue
        if seen[t] < done_cols[t]:
            owner.append(t)
        seen[t] += 1
    sigma = subtree_scales(owner, [chains[t] for t in range(len(blocks))])
    # this rank rescales ITS columns on the device: local column kl of block t is the kl-th global column owned by t
    glob = {t: [k for k, o_ in enumerate(owner) if o_ == t] for t in handles}
    mine = {}
    for t, f in handles.items():
        f.rescale([sigma[k] for k in glob[t]])
        mine[t] = f.download()
        f.close()
    return mine, owner, sigma


def _to_ints(lens, limbs):
    out, o = [], 0
    for l in lens:
        a = 0
        for t in range(abs(int(l))):
            a |= int(limbs[o + t]) << (64 * t)
        o += abs(int(l))
        out.append(-a if l < 0 else a)
    return out
