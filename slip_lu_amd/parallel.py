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


# ---------------------------------------------------------------------------------------------
# Farm completion ("completed L columns gathered", then the separator columns)
# ---------------------------------------------------------------------------------------------
_FKEYS = (("Lp", np.int64), ("Li", np.int32), ("Llen", np.int32), ("Up", np.int64), ("Ui", np.int32), ("Ulen", np.int32), ("pinv", np.int32))


def allgather_i64(dist, arr, device="cpu"):
    """All-gather one variable-length int64 array per rank (sizes first, then the padded payload): [array] by rank."""
    mine = np.ascontiguousarray(arr, dtype=np.int64)
    if dist is None:
        return [mine]
    import torch
    world = dist.get_world_size()
    size = torch.tensor([mine.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, size)
    cap = max(1, int(max(int(s.item()) for s in sizes)))
    pad = torch.zeros(cap, dtype=torch.int64, device=device)
    if mine.size:
        pad[:mine.size] = torch.from_numpy(mine).to(device)
    outs = [torch.zeros(cap, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o.cpu().numpy()[:int(s.item())].copy() for o, s in zip(outs, sizes)]


def pack_factors(mine):
    """{block: canonical factor dict} -> one int64 buffer: per block [t, n_local, K, lnz, unz, lnl, unl], the index arrays,
    the two limb slabs (uint64 viewed as int64) -- the wire format of the gathered columns"""
    parts = [np.array([len(mine)], dtype=np.int64)]
    for t in sorted(mine):
        d = mine[t]
        K = int(d["K"])
        lnz, unz = int(d["Lp"][K]), int(d["Up"][K])
        lnl, unl = int(np.abs(np.asarray(d["Llen"][:lnz], dtype=np.int64)).sum()), int(np.abs(np.asarray(d["Ulen"][:unz], dtype=np.int64)).sum())
        parts.append(np.array([t, int(d["n"]), K, lnz, unz, lnl, unl], dtype=np.int64))
        for key, size in (("Lp", K + 1), ("Li", lnz), ("Llen", lnz), ("Up", K + 1), ("Ui", unz), ("Ulen", unz), ("pinv", int(d["n"]))):
            parts.append(np.asarray(d[key][:size], dtype=np.int64))
        parts.append(np.asarray(d["Llimbs"][:lnl], dtype=np.uint64).view(np.int64))
        parts.append(np.asarray(d["Ulimbs"][:unl], dtype=np.uint64).view(np.int64))
    return np.concatenate(parts)


def unpack_factors(buf):
    buf = np.asarray(buf, dtype=np.int64)
    out, o = {}, 1
    for _ in range(int(buf[0])):
        t, nl_, K, lnz, unz, lnl, unl = (int(v) for v in buf[o:o + 7]); o += 7
        d = dict(n=nl_, K=K)
        for (key, dt), size in zip(_FKEYS, (K + 1, lnz, lnz, K + 1, unz, unz, nl_)):
            d[key] = buf[o:o + size].astype(dt); o += size
        d["Llimbs"] = buf[o:o + lnl].view(np.uint64).copy(); o += lnl
        d["Ulimbs"] = buf[o:o + unl].view(np.uint64).copy(); o += unl
        out[t] = d
    return out


def allgather_factors(dist, mine, device="cpu"):
    """every rank's rescaled block columns on every rank: {block: canonical factor dict} (ONE all-gather of the packed
    columns: RCCL on `device`="cuda" tensors under nccl, gloo on CPU tensors in the tests)"""
    out = {}
    for buf in allgather_i64(dist, pack_factors(mine), device=device):
        out.update(unpack_factors(buf))
    return out


def assemble_prefix(n, blocks, q, facs, owner):
    """The first K = len(owner) columns of the WHOLE matrix's factorisation from the blocks' rescaled columns, in the form
    slip_hip_factor_set_prefix takes: global row ids and the reference's entry order, which is an order of GLOBAL positions
    (slip_sort_xi sorts the pattern by pinv before the pivot search; the output loop of SLIP_LU_factorize.c:226-263 walks
    it in that order after the swap): U(:,k) = the rows pivotal before k by position, then the pivot; L(:,k) = the other
    rows by their position BEFORE column k's swap.  The positions come from replaying the swaps (slip_get_pivot.c:164-176).
    Returns (factor dict with Lp/Li/Llen/Llimbs/Up/Ui/Ulen/Ulimbs, piv_row)."""
    K = len(owner)
    pinv = list(range(n)); row_at = list(range(n))
    done = [0] * len(blocks)
    off = {}
    for t, d in facs.items():
        lo = np.concatenate([[0], np.cumsum(np.abs(np.asarray(d["Llen"], dtype=np.int64)))])
        uo = np.concatenate([[0], np.cumsum(np.abs(np.asarray(d["Ulen"], dtype=np.int64)))])
        off[t] = (lo, uo)
    out = {k: [] for k in ("Li", "Llen", "Llimbs", "Ui", "Ulen", "Ulimbs")}
    Lp, Up, piv_row = [0], [0], []
    for k in range(K):
        t = owner[k]; kl = done[t]; done[t] += 1
        d, ids = facs[t], blocks[t]
        lo, uo = off[t]
        inv_local = np.argsort(d["pinv"])                       # local position -> local row
        prow = int(ids[int(inv_local[kl])])
        ent_l = [(pinv[int(ids[int(d["Li"][p])])], int(ids[int(d["Li"][p])]), p) for p in range(int(d["Lp"][kl]), int(d["Lp"][kl + 1]))]
        ent_u = [(pinv[int(ids[int(d["Ui"][p])])], int(ids[int(d["Ui"][p])]), p) for p in range(int(d["Up"][kl]), int(d["Up"][kl + 1]))]
        ent_l.sort()
        ent_u = sorted(e for e in ent_u if e[1] != prow) + [e for e in ent_u if e[1] == prow]
        for _, r, p in ent_l:
            out["Li"].append(r); out["Llen"].append(int(d["Llen"][p])); out["Llimbs"].append(d["Llimbs"][int(lo[p]):int(lo[p + 1])])
        for _, r, p in ent_u:
            out["Ui"].append(r); out["Ulen"].append(int(d["Ulen"][p])); out["Ulimbs"].append(d["Ulimbs"][int(uo[p]):int(uo[p + 1])])
        Lp.append(len(out["Li"])); Up.append(len(out["Ui"]))
        piv_row.append(prow)
        p_, d_ = pinv[prow], row_at[k]
        row_at[k], row_at[p_] = prow, d_
        pinv[prow], pinv[d_] = k, p_
        if p_ == k:
            pinv[prow] = k; row_at[k] = prow
    fac = dict(Lp=np.array(Lp, np.int64), Up=np.array(Up, np.int64),
               Li=np.array(out["Li"], np.int32), Ui=np.array(out["Ui"], np.int32),
               Llen=np.array(out["Llen"], np.int32), Ulen=np.array(out["Ulen"], np.int32),
               Llimbs=np.concatenate(out["Llimbs"]).astype(np.uint64) if out["Llimbs"] else np.zeros(0, np.uint64),
               Ulimbs=np.concatenate(out["Ulimbs"]).astype(np.uint64) if out["Ulimbs"] else np.zeros(0, np.uint64))
    return fac, np.array(piv_row, np.int32)


def farm_complete(dist, n, Ap, Ai, Ax, q, t, device="cpu", make=None, finish_on=0, **kw):
    """The subtree farm end to end: the independent components of the leading t columns of q are factorised on
    their ranks (farm_factorize), their rescaled columns are all-gathered, and rank `finish_on` (None: every rank) puts them
    into a handle of the WHOLE matrix as its first K columns (slip_hip_factor_set_prefix) and factorises the remaining
    columns -- the separator -- from there.  Returns the canonical factor dict of the whole matrix on the finishing rank(s),
    None elsewhere; equal to factorising the whole matrix in one piece.  t is cut back to the leading columns that belong to
    farmable components."""
    import slip_lu_amd as sl
    rank, world, _ = env_rank()
    qi = [int(c) for c in q]
    while t > 0:
        blocks, rest = leading_blocks(n, Ap, Ai, qi, t)
        rs = set(rest)
        bad = [k for k in range(t) if qi[k] in rs]
        if not bad:
            break
        t = bad[0]
    if t <= 0:
        blocks = []
    facs, owner = {}, []
    if blocks:
        mine, owner, _ = farm_factorize(dist, n, Ap, Ai, Ax, q, blocks, device=device, make=make, **kw)
        facs = allgather_factors(dist, mine, device=device)
    if finish_on is not None and rank != finish_on:
        return None
    alen, alimbs = sl.ints_to_slab(np.asarray(Ax, dtype=np.int64))
    f = (make or sl.Factorization)(n, Ap, Ai, alen, alimbs, np.asarray(q, dtype=np.int32), **kw)
    if owner:
        fac, piv_row = assemble_prefix(n, blocks, qi, facs, owner)
        f.set_prefix(len(owner), fac, piv_row)
    f.run()
    res = f.download()
    res.update(f.info())
    res["farm_prefix"] = len(owner)
    f.close()
    return res


def _to_ints(lens, limbs):
    out, o = [], 0
    for l in lens:
        a = 0
        for t in range(abs(int(l))):
            a |= int(limbs[o + t]) << (64 * t)
        o += abs(int(l))
        out.append(-a if l < 0 else a)
    return out
