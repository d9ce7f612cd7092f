/* ref_lu_pipe_commit.h -- the committer: ONE workgroup of the launch that runs the commit chain of the column loop.
 *
 * The commit chain (choose the pivot of column k, publish rho_k and the row swap: slip_get_pivot.c:30-183 inside
 * SLIP_LU_factorize.c:190-264) is the serial part of the factorisation: columns commit in order, and column k's values
 * need rho_{k-1}.  When every column worker runs its own commit, each hop of the chain crosses the chip.  The workers
 * therefore export PACKAGES, and the committer -- block 0, a persistent loop with rho_{k-1}, the slab cursors, the last
 * 512 swaps and pivots in its LDS -- commits them without leaving its CU.  Two kinds of package:
 *
 *   kind 0 (candidates): a column whose pivot candidates are one-limb values that were never updated (class S).  All of
 *     them carry the same factor rho_{k-1}, so the pivot choice (slip_get_smallest_pivot.c:25-101: smallest |x|, ties by
 *     pattern position; the tolerance rule of slip_get_pivot.c:89-118) is decided on the one-limb values themselves; only
 *     the pivot's product a * rho_{k-1} is formed (it is rho_k).  A source that arrives after the export sends the package back.
 *
 *   kind 1 (FULL, round 3): a short column whose non-pivotal rows are all one-limb values travels with its whole state
 *     (row, value, sign, history level).  The committer's CHAIN ENGINE keeps pinv mirrored in LDS (16 bit, n <= 16384)
 *     and the L columns it has committed this way in an LDS ring, so it applies the sources that arrived after the export
 *     ITSELF (slip_REF_triangular_solve.c:124-241 in one-limb arithmetic, fill included), brings the rows to level k-1
 *     (:248-257), searches, commits, and hands the finished rows with their positions back to the worker, which only
 *     sorts and stores them (stage 2).  A run of short dependent columns is thereby committed without a single hop
 *     across the chip; anything the engine cannot do exactly in 64-bit values (or a source whose column is not in its
 *     ring) sends the package back and the worker carries on as before.
 *
 * Per batch of up to SLIP_CB columns: (a) wave 0 polls the headers; (b) one round of loads brings the packages into LDS
 * (one wave per column); (c) wave 0 alone commits the columns one after the other on LDS only and leaves a publish record
 * per column; (d) the waves issue the stage-1 stores of the columns side by side; (e) one drain, the verdicts, the frontier.
 *
 * Package of column k: slot k % nworkers of P.pkg (offsets SLIP_PKG_* in ref_lu_pipe.h); the header is a seqlock
 * {k+1, version} (odd: being written / retracted), read before and after the contents.  The outcome goes to the
 * exporting worker's MAILBOX: verdict word {version, +-(k+1)}, pivot row, position, length, bits, slab offset, limbs
 * handed out; for a full package also the rows handed back.  Everything the committer stores for other workgroups is
 * written through (sc1) and drained before the verdicts and the frontier. */
#ifndef SLIP_REF_LU_PIPE_COMMIT_H
#define SLIP_REF_LU_PIPE_COMMIT_H

#ifndef SLIP_CB
#define SLIP_CB          8                  /* columns per batch */
#endif
#define SLIP_CB_RING     512                /* swaps and pivots the committer remembers (more than the columns in flight) */
#define SLIP_CBW         640                /* one batch column in LDS: 32 header words, 96 candidate words, 512 row words */
#define SLIP_CB_SLOTW    264                /* rho_j: a product of a one-limb value and a pivot of at most 256 digits, whole limbs */
#define SLIP_LRING       3072               /* entries of the engine's ring of L columns (3 words each) */
#define SLIP_PUBW        32

/* where the committer keeps what in its LDS (words from lds + SLIP_LDS_WORK); the host sizes the launch with it */
struct SlipCommitLayout {
    int cbuf, pub, stage, ring_row, ring_disp, ring_opos, pr_lo0, pr_lo1, pr_inv0, pr_inv1, pr_meta, ld_start, ld_cnt, ld_col, lring,
        est_row, est_vlo, est_vhi, est_meta, est_hash, misc, Ms, scr, pinvm, total;
};
#define SLIP_COMMIT_SCR  272                /* digits per scratch buffer of the exact tolerance comparison (pivots up to 256 digits) */
static inline
#if !defined(SLIP_EMULATE)
__host__ __device__
#endif
SlipCommitLayout slip_commit_layout(int n, int engine)
{
    SlipCommitLayout L; int o = 0;
    L.cbuf = o; o += SLIP_CB * SLIP_CBW;
    L.pub = o; o += SLIP_CB * SLIP_PUBW;
    L.stage = o; o += SLIP_CB * SLIP_CB_SLOTW;
    L.ring_row = o; o += SLIP_CB_RING; L.ring_disp = o; o += SLIP_CB_RING; L.ring_opos = o; o += SLIP_CB_RING;
    L.pr_lo0 = o; o += SLIP_CB_RING; L.pr_lo1 = o; o += SLIP_CB_RING; L.pr_inv0 = o; o += SLIP_CB_RING; L.pr_inv1 = o; o += SLIP_CB_RING;
    L.pr_meta = o; o += SLIP_CB_RING;
    L.ld_start = o; o += SLIP_CB_RING; L.ld_cnt = o; o += SLIP_CB_RING; L.ld_col = o; o += SLIP_CB_RING;
    L.lring = o; o += engine ? 3 * SLIP_LRING : 0;
    L.est_row = o; o += engine ? SLIP_ENG_ROWS : 0; L.est_vlo = o; o += engine ? SLIP_ENG_ROWS : 0;
    L.est_vhi = o; o += engine ? SLIP_ENG_ROWS : 0; L.est_meta = o; o += engine ? SLIP_ENG_ROWS : 0;
    L.est_hash = o; o += engine ? 2 * ((n + 7) / 8) : 0;       /* the engine's row -> place map, one byte per row of the matrix */
    L.misc = o; o += 192;
    L.Ms = o; o += SLIP_CB_SLOTW;
    L.scr = o; o += 3 * SLIP_COMMIT_SCR;
    L.pinvm = o; o += engine ? (n + 1) / 2 : 0;
    L.total = o;
    return L;
}
/* dynamic LDS words a launch with a committer needs at least */
static inline
#if !defined(SLIP_EMULATE)
__host__ __device__
#endif
int slip_commit_lds_words(int n, int engine) { return SLIP_LDS_WORK + slip_commit_layout(n, engine).total + 8; }

/* the worker's side: export the package the pre-pass has just prepared (all threads; barriers inside).  A column may
 * export again after a retraction: the version in the header, in the sums and in every candidate record tells the
 * committer which package it is looking at. */
/* F0: the frontier the pre-pass ran at (positions, history levels); Fl: the frontier up to which the worker has checked its rows since */
SLIP_DEV void slip_export_package(const SlipParams &P, const int k, uint32_t *lds, const int F0, const int Fl)
{
    const int tid = slip_tid(), T = slip_nthreads();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    const uint32_t *f_row = lds + SLIP_LDS_TAB, *f_pos = f_row + SLIP_TAB_CAP, *f_aux = f_row + 3 * SLIP_TAB_CAP;
    const uint32_t *f_k0 = lds + SLIP_LDS_KEYS, *f_k1 = f_k0 + SLIP_PAT_CAP;
    const uint32_t *cl = lds + SLIP_LDS_WORK + SLIP_CAND_CAP;
    uint32_t *pk = P.pkg.at() + (int64_t)(k % P.nworkers) * SLIP_PKG_WORDS;
    const int nrows = sv[SV_NROWS], ncand = sv[SV_PP + 1];
    const uint32_t ver = ((uint32_t) sv[SV_PKGVER] | 1u) + 1u;          /* 2, 4, 6, ... */
    if (tid == 0) { slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t)(ver - 1u) << 32) | (uint32_t)(k + 1)); slip_vm_drain(); }
    slip_block_sync();
    for (int t = tid; t < nrows; t += T) slip_st_u32(pk + SLIP_PKG_ROWS + t, f_row[t]);
    for (int c = tid; c < ncand; c += T) {
        const int t = (int) cl[c];
        uint32_t *cr = pk + SLIP_PKG_CAND + 6 * c;
        slip_st_u32(cr, (uint32_t) t); slip_st_u32(cr + 1, f_k0[t]); slip_st_u32(cr + 2, f_k1[t]); slip_st_u32(cr + 3, f_aux[t]); slip_st_u32(cr + 4, f_pos[t]);
        slip_st_u32(cr + 5, ver);
    }
    if (tid < SLIP_PP_WORDS) slip_st_u32(pk + SLIP_PKG_SUMS + tid, (uint32_t) sv[SV_PP + tid]);
    if (tid == SLIP_PP_WORDS) {
        slip_st_u32(pk + SLIP_PKG_STAMP, (uint32_t) Fl); slip_st_u32(pk + SLIP_PKG_STAMP0, (uint32_t) F0);
        slip_st_u32(pk + SLIP_PKG_NROWS, (uint32_t) nrows); slip_st_u32(pk + SLIP_PKG_VER, ver); slip_st_u32(pk + SLIP_PKG_WORKER, (uint32_t) P.worker);
        slip_st_u32(pk + SLIP_PKG_KIND, 0u); slip_st_u32(pk + SLIP_PKG_NFULL, 0u);
        slip_st_u32(P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS + SLIP_PKG_OUT, 0u);
    }
    slip_vm_drain();
    slip_block_sync();
    if (tid == 0) {
        /* the header also says how much there is to load: rows, candidates, kind (the committer reads it when it polls) */
        slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t)(ver | ((uint32_t) nrows << 8) | ((uint32_t) ncand << 18)) << 32) | (uint32_t)(k + 1));
        sv[SV_PKGVER] = (int32_t) ver; sv[SV_PKGX] = 1; sv[SV_PKGK] = 0;
    }
    slip_block_sync();
}

/* the worker's side: a FULL package -- every non-pivotal row with its one-limb value, sign and history level as the
 * pre-pass left them (slip_prepass: f_k0/f_k1, f_meta, places f_npi).  All threads; barriers inside. */
SLIP_DEV void slip_export_full(const SlipParams &P, const int k, uint32_t *lds, const int F0, const int Fl)
{
    const int tid = slip_tid(), T = slip_nthreads();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    const uint32_t *f_row = lds + SLIP_LDS_TAB;
    const uint32_t *f_k0 = lds + SLIP_LDS_KEYS, *f_k1 = f_k0 + SLIP_PAT_CAP;
    const uint32_t *f_meta = lds + SLIP_LDS_DIROFF, *f_npi = lds + SLIP_LDS_ROWS;
    uint32_t *pk = P.pkg.at() + (int64_t)(k % P.nworkers) * SLIP_PKG_WORDS;
    const int nrows = sv[SV_NROWS], nnp = sv[SV_PPF];
    const uint32_t ver = ((uint32_t) sv[SV_PKGVER] | 1u) + 1u;
    if (tid == 0) { slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t)(ver - 1u) << 32) | (uint32_t)(k + 1)); slip_vm_drain(); }
    slip_block_sync();
    for (int t = tid; t < nrows; t += T) {
        const uint32_t pi = f_npi[t];
        if (pi >> 31) continue;                                 /* pivotal: final, stays with the worker */
        uint32_t *a = pk + SLIP_PKG_ROWS + pi;                  /* four arrays of nnp words, back to back */
        slip_st_u32(a, f_row[t]); slip_st_u32(a + nnp, f_k0[t]); slip_st_u32(a + 2 * nnp, f_k1[t]);
        slip_st_u32(a + 3 * nnp, f_meta[t]);
    }
    if (tid < SLIP_PP_WORDS) slip_st_u32(pk + SLIP_PKG_SUMS + tid, (uint32_t) sv[SV_PP + tid]);
    if (tid == SLIP_PP_WORDS) {
        slip_st_u32(pk + SLIP_PKG_STAMP, (uint32_t) Fl); slip_st_u32(pk + SLIP_PKG_STAMP0, (uint32_t) F0);
        slip_st_u32(pk + SLIP_PKG_NROWS, (uint32_t) nrows); slip_st_u32(pk + SLIP_PKG_VER, ver); slip_st_u32(pk + SLIP_PKG_WORKER, (uint32_t) P.worker);
        slip_st_u32(pk + SLIP_PKG_KIND, 1u); slip_st_u32(pk + SLIP_PKG_NFULL, (uint32_t) nnp);
        slip_st_u32(P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS + SLIP_PKG_OUT, 0u);
    }
    slip_vm_drain();
    slip_block_sync();
    if (tid == 0) {
        slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t)(ver | ((uint32_t)(4 * nnp) << 8) | (1u << 23)) << 32) | (uint32_t)(k + 1));
        sv[SV_PKGVER] = (int32_t) ver; sv[SV_PKGX] = 1; sv[SV_PKGK] = 1;
    }
    slip_block_sync();
}

/* the worker's side: the package no longer describes the rows (thread 0) */
SLIP_DEV void slip_retract_package(const SlipParams &P, const int k, volatile int32_t *sv)
{
    uint32_t *pk = P.pkg.at() + (int64_t)(k % P.nworkers) * SLIP_PKG_WORDS;
    const uint32_t ver = (uint32_t) sv[SV_PKGVER] | 1u;
    slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t) ver << 32) | (uint32_t)(k + 1));
    sv[SV_PKGVER] = (int32_t) ver; sv[SV_PKGX] = 0;
    slip_agent_add_u64(&P.st->c_retract, 1ull);
}

/* one candidate: the one-limb value a (nd digits) times rho[k-1] (in registers) -> LDS slot, search key, length */
template <int D> SLIP_DEV void slip_commit_mul(const WR<D> &Mr, uint32_t a0, uint32_t a1, int nd, dig_t *slotp, dig_t *slot2, int kind, uint64_t *key_out, int *len_out,
                                               uint64_t *lo_out)
{
    const int lane = slip_lane();
    WR<D> Y;
    if (nd == 1) Y = wr_mul_digit<D>(a0, Mr);
    else {
        WR<D> A = wr_zero<D>();
        if (lane == 0) A.d[0] = a0;
        if (lane == 1) A.d[0] = a1;
        Y = wr_mul<D>(A, nd, Mr);
    }
    const int len = wr_len<D>(Y);
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; if (c < ((len + 1) & ~1)) { slotp[c] = Y.d[q]; slot2[c] = Y.d[q]; } }
    const uint32_t d1 = len ? wr_digit<D>(Y, len - 1) : 0u, d2 = len >= 2 ? wr_digit<D>(Y, len - 2) : 0u, d3 = len >= 3 ? wr_digit<D>(Y, len - 3) : 0u;
    uint64_t top = ((uint64_t) d1 << 32) | d2;
    const int sh = len ? slip_clz32(d1) : 0;
    if (sh) top = (top << sh) | (uint64_t)(d3 >> (32 - sh));
    const int bits = len ? 32 * len - sh : 0;
    uint64_t key = ((uint64_t) bits << 40) | (top >> 24);
    if (kind == 1) key = ~key;
    *key_out = key; *len_out = len;
    *lo_out = (uint64_t) slip_readlane(Y.d[0], 0) | ((uint64_t) slip_readlane(Y.d[0], 1) << 32);
}

SLIP_DEV int slip_bits64(uint64_t v) { return v ? 64 - slip_clz64(v) : 0; }

/* is  num * 2^(-te) >= tol_m * den ?  (the tolerance rule of slip_get_pivot.c:89-118 on one-limb magnitudes; tol_m has
 * exactly 53 bits).  1 / 0, or -1 when the comparison does not fit 128 bits */
SLIP_DEV int slip_tol_small(uint64_t tol_m, int te, uint64_t num, uint64_t den)
{
    const int sa = te < 0 ? -te : 0, sb = te > 0 ? te : 0;
    const int bn = slip_bits64(num) + sa, bd = slip_bits64(den) + sb;
    if (bn < 52 + bd) return 0;
    if (bn > 53 + bd) return 1;
    if (bn > 126 || bd + 53 > 126) return -1;
    const slip_u128 lhs = (slip_u128) num << sa, rhs = ((slip_u128) tol_m * den) << sb;
    return lhs >= rhs ? 1 : 0;
}

/* a pivot of the committer's ring: one-limb pivots keep what the in-lane arithmetic needs */
struct SlipSmallPiv { uint64_t lo, inv; int ctz, sgn, small, bits; };

/* the kernel body of the committer (block 0 of a launch with P.committer set) */
template <bool FAST>
SLIP_DEV void slip_committer(const SlipParams &P, SlipState *st, uint32_t *lds)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    volatile int64_t *sv64 = (volatile int64_t *)(lds + SLIP_LDS_VARS);
    const SlipCommitLayout Ly = slip_commit_layout(P.n, P.engine);
    uint32_t *base = lds + SLIP_LDS_WORK;
    uint32_t *cbuf = base + Ly.cbuf, *pub = base + Ly.pub;
    dig_t *stage = base + Ly.stage;
    uint32_t *ring_row = base + Ly.ring_row, *ring_disp = base + Ly.ring_disp, *ring_opos = base + Ly.ring_opos;
    uint32_t *pr_lo0 = base + Ly.pr_lo0, *pr_lo1 = base + Ly.pr_lo1, *pr_inv0 = base + Ly.pr_inv0, *pr_inv1 = base + Ly.pr_inv1, *pr_meta = base + Ly.pr_meta;
    uint32_t *ld_start = base + Ly.ld_start, *ld_cnt = base + Ly.ld_cnt, *ld_col = base + Ly.ld_col, *lring = base + Ly.lring;
    uint32_t *est_row = base + Ly.est_row, *est_vlo = base + Ly.est_vlo, *est_vhi = base + Ly.est_vhi, *est_meta = base + Ly.est_meta;
    uint8_t *slotm = (uint8_t *)(base + Ly.est_hash);              /* engine: place + 1 of a row in the column being committed, 0 = not in it */
    uint32_t *misc = base + Ly.misc;
    uint32_t *ck_pos = misc, *hver = misc + 64;                     /* candidate positions; the versions of the batch's packages as the poll saw them */
    unsigned long long *eacc = (unsigned long long *)(misc + 128);  /* the engine's algorithmic counters (8 x 64 bit) */
    dig_t *Ms = base + Ly.Ms;                                       /* rho[j-1]'s digits */
    dig_t *b0 = base + Ly.scr, *b1 = b0 + SLIP_COMMIT_SCR, *b2 = b1 + SLIP_COMMIT_SCR;
    uint16_t *pinvm = (uint16_t *)(base + Ly.pinvm);                /* engine: pinv as it stands at the column being committed */
    SlipPiv *Mrec = (SlipPiv *)(lds + SLIP_LDS_SCAN);               /* rho[j-1]'s record */
    const int wcap = SLIP_COMMIT_SCR;
    const int scheme = P.pivot_scheme;
    const int kind = (scheme == 4 || scheme == 5) ? 1 : 0;
    const int diagpref = scheme == 1 || scheme == 3 || scheme == 4;
    const bool mirror = P.engine != 0;
    enum { C_K = SV_PP + SLIP_PP_WORDS, C_HAVE, C_GO, C_RING0, C_REJ, C_REJV, C_ST, C_LASTPR, C_NBC, C_LW, C_PR0 };
    const uint32_t BIG = 0x7FFFFFFFu;
    /* ring entry of pivot c for the in-lane arithmetic */
    auto pring_get = [&](int c) -> SlipSmallPiv {
        const int s_ = c & (SLIP_CB_RING - 1);
        SlipSmallPiv p; const uint32_t m = pr_meta[s_];
        p.lo = (uint64_t) pr_lo0[s_] | ((uint64_t) pr_lo1[s_] << 32); p.inv = (uint64_t) pr_inv0[s_] | ((uint64_t) pr_inv1[s_] << 32);
        p.ctz = (int)(m & 0xFFu); p.sgn = (m >> 8) & 1u ? -1 : 1; p.small = (int)((m >> 9) & 1u); p.bits = (int)(m >> 16);
        return p;
    };
    auto pring_put = [&](int c, const SlipPiv &pv) {                 /* one lane */
        const int s_ = c & (SLIP_CB_RING - 1);
        const int small = slip_abs(pv.len) <= 2 && pv.len != 0;
        pr_lo0[s_] = (uint32_t) pv.lo; pr_lo1[s_] = (uint32_t)(pv.lo >> 32); pr_inv0[s_] = (uint32_t) pv.inv64; pr_inv1[s_] = (uint32_t)(pv.inv64 >> 32);
        pr_meta[s_] = (uint32_t)(pv.ctz & 0xFF) | (pv.len < 0 ? 0x100u : 0u) | (small ? 0x200u : 0u) | ((uint32_t)(small ? pv.bits : 0) << 16);
    };
    auto est_lookup = [&](uint32_t row) -> int { return (int) slotm[row] - 1; };
    auto est_insert = [&](uint32_t row, int idx) { slotm[row] = (uint8_t)(idx + 1); };
    if (tid == 0) {
        int pr_; const int F0 = slip_ld_frontier(st, &pr_);
        sv[C_K] = F0; sv[C_HAVE] = 0; sv[C_RING0] = F0; sv[C_REJ] = -1; sv[C_REJV] = 0; sv[C_LW] = 0; sv[C_PR0] = F0;
        for (int q = 0; q < 8; q++) eacc[q] = 0ull;
        if (F0 >= 1) { const SlipPiv pv = slip_ld_piv(P.piv.at(F0 - 1)); pring_put(F0 - 1, pv); sv[C_PR0] = F0 - 1; }
    }
    for (int w = tid; w < SLIP_CB_RING; w += T) { ld_col[w] = 0xFFFFFFFFu; ld_cnt[w] = 0xFFFFFFFFu; }
    if (mirror) for (int i = tid; i < P.n; i += T) { pinvm[i] = (uint16_t) slip_ld_i32(P.pinv.at(i)); slotm[i] = 0; }
    slip_block_sync();
    if (tid == 0) slip_agent_store_i32(&st->committer_up, 1);       /* from now on packages are answered */
#ifdef SLIP_PROFILE_COMMIT
    /* calibration, in place: what a dependent LDS read, a dependent VALU op and an s_memrealtime stamp cost HERE (shader
     * cycles and 10 ns ticks) -- slots 20..23 of the phase record */
    unsigned long long cal_[4] = {0, 0, 0, 0};
    if (wave == 0) {
        for (int w = lane; w < 64; w += SLIP_WAVE) misc[w] = (uint32_t)((w * 37 + 11) & 63);
        slip_wave_sync_lds();
        uint32_t idx = (uint32_t) lane & 63u;
        const unsigned long long c0 = slip_clock(), r0 = slip_realtime();
        for (int r = 0; r < 256; r++) idx = misc[idx];
        const unsigned long long c1 = slip_clock(), r1 = slip_realtime();
        uint32_t acc = idx;
        for (int r = 0; r < 1024; r++) acc = acc * 3u + 1u;
        const unsigned long long c2 = slip_clock(), r2 = slip_realtime();
        unsigned long long r3 = r2;
        for (int r = 0; r < 64; r++) r3 += slip_realtime() & 1ull;
        const unsigned long long c3 = slip_clock();
        if (acc == 0x12345u) misc[0] = (uint32_t) r3;
        cal_[0] = ((c1 - c0) << 32) | (r1 - r0); cal_[1] = ((c2 - c1) << 32) | (r2 - r1); cal_[2] = c3 - c2;
    }
    unsigned long long tq_ = slip_realtime(), tacc_[24] = {0};
    tacc_[20] = cal_[0]; tacc_[21] = cal_[1]; tacc_[22] = cal_[2];
#define SLIP_CT(i) do { if (tid == 0) { const unsigned long long n_ = slip_realtime(); tacc_[i] += n_ - tq_; tq_ = n_; } } while (0)
#else
#define SLIP_CT(i) do { } while (0)
#endif
    for (;;) {
        /* (a) the next columns whose packages are there; a column committed by its worker moves the frontier instead */
        if (wave == 0) {
            int go = 0; unsigned long long spins = 0;
            for (;;) {
                /* one lane looks, every lane acts on the same values */
                const int kprev = sv[C_K];
                slip_wave_sync_lds();
                if (lane == 0) {
                    int pr_; const int F = slip_ld_frontier(st, &pr_);
                    sv64[SV_LEXACT / 2] = slip_ld_i64(&st->stop);
                    if (F > kprev) { sv[C_K] = F; sv[C_HAVE] = 0; }
                }
                slip_wave_sync_lds();
                const int kc = sv[C_K];
                if (kc > kprev) {
                    /* columns committed by their workers: their swaps from the log every publisher keeps, their pivots */
                    const int lo_c = kc - kprev > SLIP_CB_RING ? kc - SLIP_CB_RING : kprev;
                    for (int c = lo_c + lane; c < kc; c += SLIP_WAVE) {
                        ring_row[c & (SLIP_CB_RING - 1)] = (uint32_t) slip_ld_i32(P.row_perm.at(c));
                        ring_disp[c & (SLIP_CB_RING - 1)] = (uint32_t) slip_ld_i32(P.sw_row.at(c));
                        ring_opos[c & (SLIP_CB_RING - 1)] = (uint32_t) slip_ld_i32(P.sw_pos.at(c));
                        const SlipPiv pv = slip_ld_piv(P.piv.at(c));
                        pring_put(c, pv);
                        ld_col[c & (SLIP_CB_RING - 1)] = 0xFFFFFFFFu;       /* its L column is not in the engine's ring */
                    }
                    slip_wave_sync_lds();
                    if (mirror) {
                        if (lo_c > kprev) { for (int i = lane; i < P.n; i += SLIP_WAVE) pinvm[i] = (uint16_t) slip_ld_i32(P.pinv.at(i)); }     /* fell behind the ring: start over from memory (every swap below kc has landed) */
                        else if (lane == 0) for (int c = kprev; c < kc; c++) {
                            const int s_ = c & (SLIP_CB_RING - 1);
#if defined(SLIP_EMULATE) && defined(SLIP_EMU_TRACE)
                            fprintf(stderr, "resync col %d: pivot row %u, row %u to pos %u (mirror had %d / %d)\n", c, ring_row[s_], ring_disp[s_], ring_opos[s_], (int) pinvm[ring_row[s_]], (int) pinvm[ring_disp[s_]]);
#endif
                            pinvm[ring_row[s_]] = (uint16_t) c; pinvm[ring_disp[s_]] = (uint16_t) ring_opos[s_];
                        }
                    }
                    if (lane == 0) {
                        if (kc - sv[C_RING0] > SLIP_CB_RING) sv[C_RING0] = kc - SLIP_CB_RING;
                        if (lo_c > kprev) sv[C_PR0] = lo_c; else if (kc - sv[C_PR0] > SLIP_CB_RING) sv[C_PR0] = kc - SLIP_CB_RING;
                    }
                    slip_wave_sync_lds();
                }
                const int64_t stop = sv64[SV_LEXACT / 2];
                if (kc >= P.k_stop || (stop >> 8) <= (int64_t) kc) { go = 0; break; }
                const int j = kc + lane;
                uint64_t h = 0;
                if (lane < SLIP_CB && j < P.k_stop && (stop >> 8) > (int64_t) j) h = slip_ld_u64((const uint64_t *)(P.pkg.at() + (int64_t)(j % P.nworkers) * SLIP_PKG_WORDS + SLIP_PKG_HDR));
                const uint32_t hw_ = (uint32_t)(h >> 32), hv = hw_ & 0xFFu;        /* version; above it the sizes of the package */
                const int rdy = (uint32_t) h == (uint32_t)(j + 1) && hv >= 2u && !(hv & 1u) && !(j == sv[C_REJ] && hv == (uint32_t) sv[C_REJV]);
                if (lane < SLIP_CB) { hver[lane] = hv; hver[SLIP_CB + lane] = hw_; }
                const int nb = slip_ctz64(~slip_ballot(rdy));
                if (nb >= 1) { go = nb < SLIP_CB ? nb : SLIP_CB; break; }
                slip_sleep_short();
                if (++spins > SLIP_SPIN_LIMIT) { if (lane == 0) { st->dbg_who = 4; st->dbg_k = kc; slip_raise_stop(st, 0, SLIPDEV_INTERNAL); } go = 0; break; }
            }
            if (lane == 0) sv[C_GO] = go;
        }
        slip_block_sync();
        const int nb = sv[C_GO];
        if (!nb) break;
        SLIP_CT(0);                                  /* 0: waiting for packages */
        const int kc = sv[C_K];
        const int have = sv[C_HAVE];
        /* (b) the packages into LDS (one wave per column) and what can be said about each by itself */
        for (int i = wave; i < nb; i += nw) {
            const int j = kc + i;
            const uint32_t *pk = P.pkg.at() + (int64_t)(j % P.nworkers) * SLIP_PKG_WORDS;
            uint32_t *cb = cbuf + i * SLIP_CBW;
            if (lane < SLIP_PP_WORDS) cb[lane] = slip_ld_u32(pk + SLIP_PKG_SUMS + lane);
            else if (lane == 14) cb[14] = slip_ld_u32(pk + SLIP_PKG_STAMP);
            else if (lane == 15) cb[15] = slip_ld_u32(pk + SLIP_PKG_STAMP0);
            else if (lane == 16) cb[16] = slip_ld_u32(pk + SLIP_PKG_NROWS);
            else if (lane == 17) cb[17] = (uint32_t) slip_ld_i32(P.row_perm.at(j));
            else if (lane == 18) cb[18] = 0u;
            else if (lane == 19) cb[19] = slip_ld_u32(pk + SLIP_PKG_VER);
            else if (lane == 20) cb[20] = slip_ld_u32(pk + SLIP_PKG_WORKER);
            else if (lane == 21) cb[21] = slip_ld_u32(pk + SLIP_PKG_KIND);
            else if (lane == 22) cb[22] = slip_ld_u32(pk + SLIP_PKG_NFULL);
            {
                const uint32_t hw_ = hver[SLIP_CB + i];
                int nload = (int)((hw_ >> 8) & 0x3FFu), ncl = 6 * (int)((hw_ >> 18) & 0x1Fu);
                if (nload > 512) nload = 512;
                if (ncl > 6 * SLIP_PKG_CANDS) ncl = 6 * SLIP_PKG_CANDS;
                for (int c = lane; c < ncl; c += SLIP_WAVE) cb[32 + c] = slip_ld_u32(pk + SLIP_PKG_CAND + c);
                for (int c = lane; c < nload; c += SLIP_WAVE) cb[128 + c] = slip_ld_u32(pk + SLIP_PKG_ROWS + c);
            }
            /* the header again, behind the contents (seqlock): a package that is being rewritten is offered again later */
            const uint64_t h2 = slip_ld_u64((const uint64_t *)(pk + SLIP_PKG_HDR));
            slip_wave_sync_lds();
            const int kindp = (int) cb[21];
            const int nrows = (int) cb[16], ncand = (int) cb[1], stamp = (int) cb[14], stamp0 = (int) cb[15];
            int hit = (uint32_t) h2 != (uint32_t)(j + 1) || (uint32_t)(h2 >> 32) != hver[SLIP_CB + i] || cb[19] != hver[i] || cb[20] >= (uint32_t) P.nworkers
                      || stamp < stamp0 || stamp > j || stamp0 < sv[C_RING0] || j - stamp0 > SLIP_CB_RING - SLIP_CB - 1;
            if (kindp == 0) {
                if (nrows < 1 || nrows > SLIP_PKG_NROWMAX || ncand < 1 || ncand > SLIP_PKG_CANDS || cb[12] != 0 || (hver[SLIP_CB + i] >> 8) != (uint32_t)(nrows | (ncand << 10))) hit = 1;
                if (!hit && lane < ncand && cb[32 + 6 * lane + 5] != hver[i]) hit = 1;
                const uint32_t *rows = cb + 128;
                if (!hit) {
                    /* the rows against the pivots its worker has not seen (those committed before this batch) */
                    if (mirror) {
                        for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE && 64 * q < nrows; q++) {
                            const int t = lane + 64 * q;
                            if (t < nrows) { const int p_ = (int) pinvm[rows[t]]; if (p_ >= stamp && p_ < kc) hit = 1; }
                        }
                    } else {
                        uint32_t rr[SLIP_PKG_NROWMAX / SLIP_WAVE];
#pragma unroll
                        for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) rr[q] = lane + 64 * q < nrows ? rows[lane + 64 * q] : 0xFFFFFFFFu;
                        for (int c = stamp; c < kc; c++) {
                            const uint32_t r = ring_row[c & (SLIP_CB_RING - 1)];
#pragma unroll
                            for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) if (rr[q] == r) hit = 1;
                        }
                    }
                }
                if (!slip_ballot(hit)) {
                    /* for the serial step: every candidate's ROW and its position as the reference has it at column kc (the mirror,
                     * or the value the worker read at frontier stamp0 and where the LAST swap since then that displaced the row put
                     * it); which candidate is the diagonal row */
                    const uint32_t myrow = lane < ncand ? rows[cb[32 + 6 * lane]] : 0xFFFFFFFFu;
                    uint32_t mypos = 0x7FFFFFFFu;
                    if (mirror) { if (lane < ncand) mypos = (uint32_t) pinvm[myrow]; }
                    else {
                        int last = -1;
                        for (int e0 = stamp0; e0 < kc; e0 += SLIP_WAVE) {
                            const int e = e0 + lane;
                            const uint32_t d = e < kc ? ring_disp[e & (SLIP_CB_RING - 1)] : 0xFFFFFFFEu;
                            for (int c = 0; c < ncand; c++) {
                                const uint64_t m = slip_ballot(d == slip_readlane(myrow, c));
                                if (m && lane == c) last = e0 + 63 - slip_clz64(m);
                            }
                        }
                        if (lane < ncand) mypos = last >= 0 ? ring_opos[last & (SLIP_CB_RING - 1)] : cb[32 + 6 * lane + 4];
                    }
                    const int diag_t = (int) cb[13] - 1;
                    const uint64_t dm = slip_ballot(lane < ncand && (int) cb[32 + 6 * lane] == diag_t);
                    slip_wave_sync_lds();
                    if (lane < ncand) { cb[32 + 6 * lane] = myrow; cb[32 + 6 * lane + 4] = mypos; }
                    const int dc = diag_t < 0 ? 0xFF : (dm ? slip_ctz64(dm) : 0xFE);      /* 0xFF: no diagonal row; 0xFE: it is not among the candidates */
                    if (lane == 0) cb[23] = (uint32_t) dc;
                    /* the choice itself, HERE, side by side for the columns of the batch: every comparison slip_get_pivot makes
                     * is among this column's own candidates (class S: a * rho[j-1] with one rho for all, so |a| decides,
                     * slip_get_smallest_pivot.c:58-101 / slip_get_largest_pivot.c; the diagonal rule, slip_get_pivot.c:68-146, is a
                     * ratio of two of them).  Only a tie is broken by positions, which a swap earlier in this batch may still
                     * change: such a column is chosen again in the serial step (word 24, bit 8). */
                    {
                        const bool isc = lane < ncand;
                        const uint32_t c_a0 = isc ? cb[32 + 6 * lane + 1] : 0u, c_a1 = isc ? cb[32 + 6 * lane + 2] : 0u, c_ax = isc ? cb[32 + 6 * lane + 3] : 0u;
                        const uint64_t av = (uint64_t) c_a0 | ((uint64_t) c_a1 << 32);
                        const uint64_t mykey = isc ? (kind == 0 ? av : ~av) : ~0ull;
                        const uint32_t mh = slip_wave_min_u32((uint32_t)(mykey >> 32));
                        const uint32_t ml = slip_wave_min_u32((uint32_t)(mykey >> 32) == mh ? (uint32_t) mykey : 0xFFFFFFFFu);
                        const uint64_t mk = ((uint64_t) mh << 32) | ml;
                        const uint64_t tie = slip_ballot(isc && mykey == mk);
                        uint32_t spec = 0;
                        int bc = 0;
                        if (slip_popc64(tie) > 1) spec = 0x100u;
                        else if (!tie) spec = 0x400u;
                        else {
                            bc = slip_ctz64(tie);
                            if (diagpref && dc != 0xFF && dc != bc) {
                                if (dc == 0xFE) spec = 0x200u;               /* the diagonal row is not among the candidates sent: the worker decides */
                                else if (scheme == 1 || P.tol_mode == 0) bc = dc;
                                else {
                                    const uint64_t ab = (uint64_t) slip_readlane(c_a0, bc) | ((uint64_t) slip_readlane(c_a1, bc) << 32);
                                    const uint64_t ad = (uint64_t) slip_readlane(c_a0, dc) | ((uint64_t) slip_readlane(c_a1, dc) << 32);
                                    const int tk = slip_tol_small(P.tol_m, P.tol_e, scheme == 3 ? ab : ad, scheme == 3 ? ad : ab);
                                    if (tk < 0) spec = 0x200u; else if (tk) bc = dc;
                                }
                            }
                        }
                        const uint32_t s_row = slip_readlane(myrow, bc), s_a0 = slip_readlane(c_a0, bc), s_a1 = slip_readlane(c_a1, bc);
                        const uint32_t s_ax = slip_readlane(c_ax, bc), s_pos = slip_readlane(mypos, bc);
                        if (lane == 0) { cb[24] = spec | (uint32_t) bc; cb[25] = s_row; cb[26] = s_a0; cb[27] = s_a1; cb[28] = s_ax; cb[29] = s_pos; }
                    }
                }
            } else if (kindp == 1) {
                if (!mirror || (int) cb[22] < 1 || (int) cb[22] > SLIP_PKG_FULLMAX || stamp0 < 1 || stamp0 - 1 < sv[C_PR0] || (hver[SLIP_CB + i] >> 8) != (uint32_t)(4 * (int) cb[22]) + (1u << 15)) hit = 1;
            } else hit = 1;
            if (slip_ballot(hit) && lane == 0) cb[18] = 1u;
        }
        if (!have) {
            if (tid == T - 1) sv64[SV_LNZ / 2] = slip_ld_i64(&P.Lp[kc]);
            if (tid == T - 2) sv64[SV_LNL / 2] = slip_ld_i64(&P.Lo[kc]);
            if (tid == T - 3) sv64[SV_UNZ / 2] = slip_ld_i64(&P.Up[kc]);
            if (tid == T - 4) sv64[SV_UNL / 2] = slip_ld_i64(&P.Uo[kc]);
            if (tid == T - 5) *Mrec = slip_ld_piv(P.piv.at(kc - 1));       /* kc >= 1: column 0 is never packaged */
        }
        slip_block_sync();
        /* (b2) a row of the pattern that becomes pivotal earlier in this batch sends the package back (the pivots before the batch
         * were checked above).  The pivots chosen above are the ones the serial step commits -- or the batch ends before this column
         * -- so each wave checks its column against them; columns whose pivot is not known yet (a tie, a full package) are left to
         * the serial step (word 30: their mask) */
        for (int i = wave; i < nb; i += nw) {
            uint32_t *cb = cbuf + i * SLIP_CBW;
            uint32_t unk = 0;
            if (!cb[18] && cb[21] == 0u) {
                const int nrows = (int) cb[16];
                const uint32_t *rows = cb + 128;
                uint32_t rr[SLIP_PKG_NROWMAX / SLIP_WAVE];
#pragma unroll
                for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) rr[q] = lane + 64 * q < nrows ? rows[lane + 64 * q] : 0xFFFFFFFFu;
                /* lane e: column kc + e's state */
                const uint32_t *ce = cbuf + (lane < nb ? lane : 0) * SLIP_CBW;
                const uint32_t e_pre = ce[18], e_kind = ce[21], e_spec = ce[24], e_row = ce[25];
                const uint64_t known = slip_ballot(lane < i && !e_pre && e_kind == 0u && !(e_spec & 0x700u));
                unk = (uint32_t)(slip_ballot(lane < i && !e_pre) & ~known);
                int hit = 0;
                for (int e = 0; e < i; e++) {
                    const uint32_t r = slip_readlane(e_row, e);
                    if ((known >> e) & 1ull) {
#pragma unroll
                        for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) if (rr[q] == r) hit = 1;
                    }
                }
                const uint64_t anyhit = slip_ballot(hit);
                slip_wave_sync_lds();
                if (anyhit && lane == 0) cb[18] = 1u;
            }
            if (lane == 0) cb[30] = unk;
        }
        slip_block_sync();
        SLIP_CT(1);                                  /* 1: the packages into LDS */
        if (!have) {
            const SlipPiv M0 = *Mrec;
            const int l0 = slip_abs(M0.len);
            if (l0 <= SLIP_CB_SLOTW - 6) { const dig_t *Mg = slip_piv_digits(P, M0); for (int c = tid; c < l0; c += T) Ms[c] = slip_ld_u32(Mg + c); }
        }
        slip_block_sync();
        SLIP_CT(2);                                  /* 2: rho after a resynchronisation */
        /* (c) the columns of the batch, one after the other, by wave 0 on LDS only */
        if (wave == 0) {
            int nbc = 0, rej = -1;
            /* what the chain carries from column to column lives in registers: the slab cursors, rho[j-1]'s record, the swaps of
             * this batch (lane e: column kc + e).  LDS is touched with whole-wave reads only (a dependent LDS round trip is
             * 30-40 ns here, and the old code made a hundred of them per column). */
            int64_t Lnz_ = sv64[SV_LNZ / 2], Lnl_ = sv64[SV_LNL / 2], Unz_ = sv64[SV_UNZ / 2], Unl_ = sv64[SV_UNL / 2];
            SlipPiv M = *Mrec;
            uint32_t bs_row = 0xFFFFFFFFu, bs_disp = 0xFFFFFFFFu, bs_opos = 0xFFFFFFFFu;
            const int xcap_ = P.xcap, wcapP = P.wcap, invcap_ = P.invcap, limb_cap_ = P.limb_cap, nworkers_ = P.nworkers;
            const int64_t Lcap_nz_ = P.Lcap_nz, Lcap_nl_ = P.Lcap_nl, Ucap_nz_ = P.Ucap_nz, Ucap_nl_ = P.Ucap_nl;
            uint32_t *const mbox0 = P.pkg.at() + (int64_t) nworkers_ * SLIP_PKG_WORDS;
            for (int i = 0; i < nb; i++) {
                const int j = kc + i;
                uint32_t *cb = cbuf + i * SLIP_CBW;
                uint32_t *pb = pub + i * SLIP_PUBW;
                const uint32_t *cands = cb + 32, *rows = cb + 128;
                /* the header in one read; its fields through the scalar unit */
                const uint32_t hdr = lane < 32 ? cb[lane] : 0u;
#define HF(w) ((int) slip_readlane(hdr, (w)))
                const int nrows = HF(16), ncand = HF(1), stamp0 = HF(15), kindp = HF(21);
                uint32_t *mbx = mbox0 + (int64_t)((uint32_t) HF(20) < (uint32_t) nworkers_ ? HF(20) : 0) * SLIP_MBOX_WORDS;     /* the worker's mailbox */
                const int lm = slip_abs(M.len), brho = M.bits, slot = (lm + 3) >> 1;
                const int slotw = (lm + 5) & ~1;
                dig_t *sl = stage + i * SLIP_CB_SLOTW;
                int reject = HF(18);                            /* 1: offered again / the worker's business; 2: for good */
                /* what both kinds leave for the publish step */
                int e_pivrow = 0, e_pivpos = 0, lp_ = 0, pneg = 0, pbits = 0, nUc_all = 0, nLc = 0, nfin = 0, nlate = 0;
                uint64_t U_l = 0, Lb_total = 0, plimbs = 0, lalloc = 0, plo = 0; int64_t poff = 0;
                unsigned long long ec_src = 0, ec_read = 0, ec_str = 0, ec_upd = 0, ec_mac = 0;
                /* the row at position j (the one the pivot changes places with): as loaded at the start of the batch, or the row a
                 * swap of this batch displaced to j */
                int intermed2 = HF(17);
                { const uint64_t pm_ = slip_ballot(lane < i && bs_opos == (uint32_t) j); if (pm_) intermed2 = (int) slip_readlane(bs_disp, 63 - slip_clz64(pm_)); }      /* (the last such swap counts) */
                SLIP_CT(10);
                if (!reject && kindp == 0) {
                    /* ---- kind 0: candidates only ---- */
                    const int nS = HF(2), nB = HF(6);
                    nUc_all = HF(3); U_l = (uint64_t)(uint32_t) HF(4);
                    const int nA = lm > 2 ? nS : 0;
                    const int maxc = HF(9) - SLIP_PP_BIAS + brho;
                    const int maxub_all = maxc > HF(10) ? maxc : HF(10);
                    const uint64_t L_b = (uint64_t)(uint32_t) HF(5) + (uint64_t) nB * (uint64_t)((brho + 63) >> 6) + (lm <= 2 ? 2ull * (uint64_t) nS : 0ull);
                    const uint64_t preserve = (uint64_t)((maxub_all + 63) >> 6) + 1;
                    Lb_total = (uint64_t) nA * (uint64_t) slot + preserve + L_b;
                    const uint64_t Ub_total = U_l + preserve;
                    nLc = nrows - nUc_all;
                    /* the choice was made when the package was loaded (word 24; bit 8: a tie, decided below by positions as they
                     * are NOW), and the pattern was checked against the pivots of this batch that were known then; the others
                     * (word 30: ties, full packages) are checked here */
                    const uint32_t spec = (uint32_t) HF(24), unk = (uint32_t) HF(30);
                    if (unk) {
                        uint32_t rr[SLIP_PKG_NROWMAX / SLIP_WAVE];
#pragma unroll
                        for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) rr[q] = lane + 64 * q < nrows ? rows[lane + 64 * q] : 0xFFFFFFFFu;
                        int hit = 0;
                        for (int e = 0; e < i; e++) {
                            const uint32_t r = slip_readlane(bs_row, e);
                            if ((unk >> e) & 1u) {
#pragma unroll
                                for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) if (rr[q] == r) hit = 1;
                            }
                        }
                        if (slip_ballot(hit)) reject = 1;
                    }
                    SLIP_CT(11);
                    if (!reject) {
                        const bool A_ok = lm + 2 <= xcap_ && lm + 2 <= 256;
                        if (lm > SLIP_CB_SLOTW - 6 || slotw > SLIP_CB_SLOTW || (lm > 2 && !A_ok)) reject = 2;
                        if (nB > 0) {
                            const int Wn = ((HF(7) - SLIP_PP_BIAS + brho + 31) >> 5) + ((HF(8) + 31) >> 5) + 1;
                            if (Wn > wcapP || Wn > xcap_ || Wn > invcap_) reject = 2;
                        }
                        if (Lnz_ + nLc > Lcap_nz_ || Lnl_ + (int64_t) Lb_total > Lcap_nl_) reject = 2;
                        if (Unz_ + nUc_all + 1 > Ucap_nz_ || Unl_ + (int64_t) Ub_total > Ucap_nl_) reject = 2;
                        if (limb_cap_ > 0 && (int)((maxub_all + 63) >> 6) > limb_cap_) reject = 2;
                    }
                    SLIP_CT(12);
                    uint32_t a0 = (uint32_t) HF(26), a1 = (uint32_t) HF(27), ax = (uint32_t) HF(28);
                    e_pivrow = HF(25); e_pivpos = HF(29);
                    if (!reject && (spec & 0x100u)) {
                        /* a tie: equal values by pattern position (slip_get_smallest_pivot.c:58-101), the positions as the swaps of
                         * this batch have left them; then the diagonal preference as above */
                        const bool isc = lane < ncand;
                        const uint32_t c_row = isc ? cands[6 * lane] : 0xFFFFFFFFu, c_a0 = isc ? cands[6 * lane + 1] : 0u, c_a1 = isc ? cands[6 * lane + 2] : 0u;
                        const uint32_t c_ax = isc ? cands[6 * lane + 3] : 0u;
                        uint32_t mypos = isc ? cands[6 * lane + 4] : BIG;
                        for (int e = 0; e < i; e++) {
                            const uint32_t d = slip_readlane(bs_disp, e), o = slip_readlane(bs_opos, e);
                            if (c_row == d) mypos = o;
                        }
                        const uint64_t av = (uint64_t) c_a0 | ((uint64_t) c_a1 << 32);
                        const uint64_t mykey = isc ? (kind == 0 ? av : ~av) : ~0ull;
                        const uint32_t mh = slip_wave_min_u32((uint32_t)(mykey >> 32));
                        const uint32_t ml = slip_wave_min_u32((uint32_t)(mykey >> 32) == mh ? (uint32_t) mykey : 0xFFFFFFFFu);
                        const uint64_t mk = ((uint64_t) mh << 32) | ml;
                        const uint64_t tie = slip_ballot(isc && mykey == mk);
                        int est = 0, bc = 0;
                        const uint32_t bp = slip_wave_min_u32(((tie >> lane) & 1ull) ? mypos : BIG);
                        const uint64_t bm_ = slip_ballot(((tie >> lane) & 1ull) && mypos == bp);
                        est = bm_ ? 0 : SLIPDEV_INTERNAL;
                        bc = bm_ ? slip_ctz64(bm_) : 0;
                        const int dc = HF(23);
                        if (!est && diagpref && dc != 0xFF && dc != bc) {
                            if (dc == 0xFE) est = -1;                   /* not among the candidates it sent: the worker decides */
                            else if (scheme == 1 || P.tol_mode == 0) bc = dc;
                            else {
                                const uint64_t ab = (uint64_t) slip_readlane(c_a0, bc) | ((uint64_t) slip_readlane(c_a1, bc) << 32);
                                const uint64_t ad = (uint64_t) slip_readlane(c_a0, dc) | ((uint64_t) slip_readlane(c_a1, dc) << 32);
                                const int tk = slip_tol_small(P.tol_m, P.tol_e, scheme == 3 ? ab : ad, scheme == 3 ? ad : ab);
                                if (tk < 0) est = -1; else if (tk) bc = dc;
                            }
                        }
                        if (est > 0) { if (lane == 0) { if (!st->dbg_who) { st->dbg_who = 120; st->dbg_k = j; st->dbg_a = ncand; st->dbg_b = (int32_t) tie; } slip_raise_stop(st, 0, SLIPDEV_INTERNAL); } reject = 2; }
                        else if (est < 0) reject = 2;
                        a0 = slip_readlane(c_a0, bc); a1 = slip_readlane(c_a1, bc); ax = slip_readlane(c_ax, bc);
                        e_pivrow = (int) slip_readlane(c_row, bc); e_pivpos = (int) slip_readlane(mypos, bc);
                    } else if (!reject) {
                        if (spec & 0x400u) { if (lane == 0) { if (!st->dbg_who) { st->dbg_who = 120; st->dbg_k = j; st->dbg_a = ncand; st->dbg_b = (int32_t) spec; } slip_raise_stop(st, 0, SLIPDEV_INTERNAL); } reject = 2; }
                        else if (spec & 0x200u) reject = 2;
                        else {
                            /* the pivot row may have been displaced by a swap of this batch: the last one counts */
                            const uint64_t dm_ = slip_ballot(lane < i && bs_disp == (uint32_t) e_pivrow);
                            if (dm_) e_pivpos = (int) slip_readlane(bs_opos, 63 - slip_clz64(dm_));
                        }
                    }
                    SLIP_CT(13);
                    if (!reject) {
                        /* rho[j] = the pivot's one-limb value times rho[j-1]: into the stage slot (for the publish step) and into Ms */
                        const int nd = (int)((ax >> 12) & 3u);
                        if (lm <= 2) {
                            const slip_u128 y = (slip_u128)((uint64_t) a0 | ((uint64_t) a1 << 32)) * M.lo;
                            pbits = slip_bits128(y); lp_ = (pbits + 31) >> 5; plo = (uint64_t) y;
                            if (lane < 4) { const uint32_t dgt = (uint32_t)(y >> (32 * lane)); sl[lane] = dgt; Ms[lane] = dgt; }
                        } else {
                            const int Dm = (lm + 2 + 63) >> 6;
                            uint64_t key; int len;
                            if (Dm <= 1) slip_commit_mul<1>(wr_load<1>(Ms, lm), a0, a1, nd, sl, Ms, kind, &key, &len, &plo);
                            else if (Dm == 2) slip_commit_mul<2>(wr_load<2>(Ms, lm), a0, a1, nd, sl, Ms, kind, &key, &len, &plo);
                            else if (Dm == 3) slip_commit_mul<3>(wr_load<3>(Ms, lm), a0, a1, nd, sl, Ms, kind, &key, &len, &plo);
                            else slip_commit_mul<4>(wr_load<4>(Ms, lm), a0, a1, nd, sl, Ms, kind, &key, &len, &plo);
                            if (kind == 1) key = ~key;
                            pbits = (int)(key >> 40); lp_ = len;
                        }
                        pneg = (int)((ax >> 14) & 1u) ^ (M.len < 0);
                        plimbs = (uint64_t)((lp_ + 1) >> 1);
                        poff = lm > 2 ? Lnl_ + (int64_t)(ax & 0x3FFu) * slot : Lnl_ + (int64_t) nA * slot;
                        lalloc = (uint64_t) nA * (uint64_t) slot + (lm > 2 ? 0ull : plimbs);
                        if (lane == 0) ld_col[j & (SLIP_CB_RING - 1)] = 0xFFFFFFFFu;     /* its L values are long: not in the engine's ring */
                    }
                    SLIP_CT(15);
                } else if (!reject && kindp == 1) {
                    /* ---- kind 1: the chain engine ---- */
                    const int nfull = (int) cb[22];
                    const uint32_t *prow = cb + 128, *pvlo = prow + nfull, *pvhi = pvlo + nfull, *pmeta = pvhi + nfull;
                    const int pr0 = sv[C_PR0];
                    uint32_t lw = (uint32_t) sv[C_LW];
                    const int col = P.q[j];
                    SlipSmallPiv Mp; Mp.lo = M.lo; Mp.inv = M.inv64; Mp.ctz = M.ctz; Mp.sgn = M.len < 0 ? -1 : 1; Mp.small = lm <= 2 && lm >= 1; Mp.bits = M.bits;
                    if (!Mp.small) reject = 2;
                    int nst = nfull;
                    uint32_t lsrc[4] = {BIG, BIG, BIG, BIG};
                    if (!reject) {
                        /* E1: the state arrays and the row -> place table */
                        for (int q = 0; q < 2; q++) {
                            const int t = lane + 64 * q;
                            if (t < nfull) { est_row[t] = prow[t]; est_vlo[t] = pvlo[t]; est_vhi[t] = pvhi[t]; est_meta[t] = pmeta[t] & 0xBFFFFFFFu; }
                        }
                        slip_wave_sync_lds();
                        for (int q = 0; q < 2; q++) { const int t = lane + 64 * q; if (t < nfull) est_insert(prow[t], t); }
                        slip_wave_sync_lds();
                        /* E2: the rows that have become pivotal since the export are the sources still to be applied */
#pragma unroll
                        for (int q = 0; q < 4; q++) {                   /* (constant indices: the array stays in registers) */
                            const int t = lane + 64 * q;
                            if (t < nst) { const uint32_t p = (uint32_t) pinvm[est_row[t]]; if ((int) p < j) lsrc[q] = p; }
                        }
                    }
                    uint32_t ulate = 0;
                    SLIP_CT(17);
                    /* E3: the sources in ascending pivot position (slip_REF_triangular_solve.c:124-241) */
                    while (!reject) {
                        uint32_t mloc = lsrc[0] < lsrc[1] ? lsrc[0] : lsrc[1];
                        { const uint32_t m2 = lsrc[2] < lsrc[3] ? lsrc[2] : lsrc[3]; if (m2 < mloc) mloc = m2; }
                        const uint32_t cu = slip_wave_min_u32(mloc);
                        if (cu == BIG) break;
                        const int c = (int) cu;
                        int oq = -1;
#pragma unroll
                        for (int q = 0; q < 4; q++) if (lsrc[q] == cu) { oq = q; lsrc[q] = BIG; }
                        const uint64_t ob = slip_ballot(oq >= 0);
                        const int ol = slip_ctz64(ob);
                        const int jt = ol + 64 * (int) slip_readlane((uint32_t)(oq < 0 ? 0 : oq), ol);
                        const int cs = c & (SLIP_CB_RING - 1);
                        if (c - 1 < pr0) { reject = 1; break; }
                        if (ld_col[cs] != (uint32_t) c || (int32_t) ld_cnt[cs] < 0 || (uint32_t)(lw - ld_start[cs]) > (uint32_t) SLIP_LRING) {
                            /* L(:,c) is not in the ring (its worker committed it, or it was pushed out): from memory if its worker has
                             * published it (stage 2, Lready[c]) -- otherwise the package goes back and its worker waits for that */
                            int rdy = 0;
                            if (lane == 0) rdy = slip_agent_load_i32(P.Lready.at(c));
                            rdy = (int) slip_bcast0_u32((uint32_t) rdy);
                            if (!rdy) { reject = 1; break; }
                            const int64_t m0 = slip_ld_i64(&P.Lp[c]), m1 = slip_ld_i64(&P.Lp[c + 1]);
                            if (m1 - m0 > (int64_t) SLIP_ENG_ROWS || m1 < m0) { reject = 2; break; }
                            const int cnt = (int)(m1 - m0);
                            int bad = 0;
                            for (int e0 = 0; e0 < cnt; e0 += SLIP_WAVE) {
                                const int e = e0 + lane;
                                if (e < cnt) {
                                    const int ri = slip_ld_i32(&P.Li[m0 + e]);
                                    const SlipEnt le = slip_ld_ent(&P.Le[m0 + e]);
                                    uint64_t v = 0;
                                    if (slip_abs(le.len) > 2) bad = 1;
                                    else if (le.len != 0) v = slip_limb0_s((const dig_t *)(P.Llimbs + le.off));
                                    const uint32_t at = 3u * ((lw + (uint32_t) e) % (uint32_t) SLIP_LRING);
                                    lring[at] = (uint32_t) ri | (le.len < 0 ? 0x80000000u : 0u); lring[at + 1] = (uint32_t) v; lring[at + 2] = (uint32_t)(v >> 32);
                                }
                            }
                            if (slip_ballot(bad)) { reject = 2; break; }
                            if (lane == 0) { ld_start[cs] = lw; ld_cnt[cs] = (uint32_t) cnt; ld_col[cs] = (uint32_t) c; }
                            lw += (uint32_t) cnt;
                            slip_wave_sync_lds();
                        }
                        const SlipSmallPiv R = pring_get(c), Dv = pring_get(c - 1);
                        if (!R.small || !Dv.small) { reject = 2; break; }
                        const uint32_t jrow = est_row[jt];
                        uint64_t xj = (uint64_t) est_vlo[jt] | ((uint64_t) est_vhi[jt] << 32);
                        const uint32_t mj = est_meta[jt];
                        int sj = xj ? ((mj >> 31) ? -1 : 1) : 0;
                        const int hj = (int)(mj & 0x3FFFFFFFu) - 1;
                        /* bring x[j] to its final value: history update to level c-1 (:139-149) */
                        if (sj != 0 && hj < c - 1) {
                            const slip_u128 y = (slip_u128) xj * Dv.lo; sj *= Dv.sgn;
                            if (hj >= 0) {
                                const SlipSmallPiv H = pring_get(hj);
                                if (hj < pr0 || !H.small || slip_bits128(y) - H.bits + 1 > 64) { reject = 2; break; }
                                xj = slip_divexact_to64(y, H.ctz, H.inv); sj *= H.sgn;      /* the quotient fits one limb: one multiply */
                            } else {
                                if ((uint64_t)(y >> 64)) { reject = 2; break; }
                                xj = (uint64_t) y;
                            }
                        }
                        slip_wave_sync_lds();                        /* every lane has read the row before one lane rewrites it */
                        if (lane == 0) { est_vlo[jt] = (uint32_t) xj; est_vhi[jt] = (uint32_t)(xj >> 32); est_meta[jt] = (sj < 0 ? 0x80000000u : 0u) | 0x40000000u | (uint32_t)(hj + 1); }
                        slip_wave_sync_lds();
                        nlate++; ulate += xj != 0;
                        const int src_nz = sj != 0, bxj = slip_bits64(xj);
                        if (src_nz) { ec_src++; ec_read += 16; }
                        const int dcnt = (int) ld_cnt[cs]; const uint32_t ds = ld_start[cs];
                        for (int e0 = 0; e0 < dcnt && !reject; e0 += SLIP_WAVE) {
                            const int e = e0 + lane, has = e < dcnt;
                            uint32_t w0 = 0, lvl = 0, lvh = 0;
                            if (has) { const uint32_t at = 3u * ((ds + (uint32_t) e) % (uint32_t) SLIP_LRING); w0 = lring[at]; lvl = lring[at + 1]; lvh = lring[at + 2]; }
                            const uint32_t ri = w0 & 0x7FFFFFFFu;
                            const uint64_t lv = (uint64_t) lvl | ((uint64_t) lvh << 32);
                            int idx = has ? est_lookup(ri) : 0;
                            const int fresh = has && idx < 0;
                            const uint64_t fm = slip_ballot(fresh);
                            const int nf = slip_popc64(fm);
                            if (nst + nf > SLIP_ENG_ROWS - 1) { reject = 2; break; }       /* places travel as bytes: 255 rows */
                            if (fresh) {
                                /* structural discovery (what the reference's DFS does): a row the column did not hold yet */
                                idx = nst + slip_popc64(fm & ((1ull << lane) - 1ull));
                                est_row[idx] = ri; est_vlo[idx] = 0u; est_vhi[idx] = 0u; est_meta[idx] = 0u;
                                est_insert(ri, idx);
                            }
                            const int nst_old = nst; nst += nf;
                            int ovf = 0;
                            const int upd = has && src_nz && ri != jrow && lv != 0;
                            if (src_nz) {
                                const uint64_t hm = slip_ballot(has), nzm = slip_ballot(has && lv != 0);
                                ec_str += (unsigned long long) slip_popc64(hm); ec_read += 4ull * slip_popc64(hm) + 8ull * slip_popc64(nzm);
                            }
                            int had_x = 0;
                            if (upd) {
                                /* one IPGE update (:175-237) in 128-bit arithmetic */
                                const uint64_t xi = (uint64_t) est_vlo[idx] | ((uint64_t) est_vhi[idx] << 32);
                                const uint32_t mi = est_meta[idx];
                                const int si = xi ? ((mi >> 31) ? -1 : 1) : 0, hi_ = (int)(mi & 0x3FFFFFFFu) - 1;
                                const int lx = si != 0, hist = lx && hi_ < c - 1, hdiv = hist && hi_ > -1;
                                had_x = lx;
                                SlipSmallPiv H; H.lo = 1; H.inv = 1; H.ctz = 0; H.sgn = 1; H.small = 1; H.bits = 1;
                                if (hdiv) { H = pring_get(hi_); if (hi_ < pr0 || !H.small) ovf = 1; }
                                /* hist(x_i) = x_i * rho[c-1] / rho[h] must fit one limb, like every value the engine keeps */
                                uint64_t yh = xi; int s1 = si * R.sgn;
                                if (hist) {
                                    const slip_u128 y = (slip_u128) xi * Dv.lo; s1 *= Dv.sgn;
                                    if (hdiv) {
                                        if (slip_bits128(y) - H.bits + 1 > 64) ovf = 1;
                                        yh = slip_divexact_to64(y, H.ctz, H.inv); s1 *= H.sgn;
                                    } else { if ((uint64_t)(y >> 64)) ovf = 1; yh = (uint64_t) y; }
                                }
                                const int b1b = lx ? slip_bits64(yh) + R.bits : 0, b2b = slip_bits64(lv) + bxj;
                                if ((b1b > b2b ? b1b : b2b) + 1 > 126) ovf = 1;
                                if (!ovf) {
                                    const slip_u128 y = lx ? (slip_u128) yh * R.lo : (slip_u128) 0;
                                    const slip_u128 p2 = (slip_u128) lv * xj;
                                    const int s2 = ((w0 >> 31) ? -1 : 1) * sj;
                                    slip_u128 mag; int sT;
                                    if (!lx) { mag = p2; sT = -s2; }
                                    else if (s1 == s2) { if (y >= p2) { mag = y - p2; sT = s1; } else { mag = p2 - y; sT = -s1; } }
                                    else { mag = y + p2; sT = s1; }
                                    if (mag != 0 && slip_bits128(mag) - Dv.bits + 1 > 64) ovf = 1;      /* c >= 1 here: the division by rho[c-1] */
                                    else {
                                        const uint64_t nv = slip_divexact_to64(mag, Dv.ctz, Dv.inv); sT *= Dv.sgn;
                                        est_vlo[idx] = (uint32_t) nv; est_vhi[idx] = (uint32_t)(nv >> 32);
                                        est_meta[idx] = (nv && sT < 0 ? 0x80000000u : 0u) | (uint32_t)(c + 1);
                                    }
                                }
                            }
                            {
                                const uint64_t um = slip_ballot(upd), xm = slip_ballot(upd && had_x);
                                ec_upd += (unsigned long long) slip_popc64(um); ec_mac += (unsigned long long)(slip_popc64(um) + slip_popc64(xm));
                            }
                            if (slip_ballot(ovf)) { reject = 2; break; }
                            slip_wave_sync_lds();
                            if (nf)
#pragma unroll
                            for (int q = 0; q < 4; q++) {       /* a filled-in row may itself have become pivotal meanwhile: a later source */
                                const int t = lane + 64 * q;
                                if (t >= nst_old && t < nst) { const uint32_t p = (uint32_t) pinvm[est_row[t]]; lsrc[q] = (int) p < j ? p : BIG; }
                            }
                        }
                    }
                    SLIP_CT(18);
                    /* E4: the non-pivotal rows to level j-1 (:248-257), the pivot search among them */
                    int pt = -1;
                    if (!reject) {
                        uint64_t fin[4]; int fneg[4], isL[4]; int ovf = 0;
                        for (int q = 0; q < 4; q++) { fin[q] = 0; fneg[q] = 0; isL[q] = 0; }
                        for (int q = 0; q < 4 && 64 * q < nst; q++) {
                            const int t = lane + 64 * q;
                            if (t < nst) {
                                const uint32_t m = est_meta[t];
                                if (!((m >> 30) & 1u)) {
                                    isL[q] = 1;
                                    uint64_t x = (uint64_t) est_vlo[t] | ((uint64_t) est_vhi[t] << 32);
                                    if (x) {
                                        int s_ = (m >> 31) ? -1 : 1; const int h = (int)(m & 0x3FFFFFFFu) - 1;
                                        if (h < j - 1) {
                                            const slip_u128 y = (slip_u128) x * Mp.lo; s_ *= Mp.sgn;
                                            if (h >= 0) {
                                                const SlipSmallPiv H = pring_get(h);
                                                if (h < pr0 || !H.small || slip_bits128(y) - H.bits + 1 > 64) ovf = 1;
                                                else { x = slip_divexact_to64(y, H.ctz, H.inv); s_ *= H.sgn; }
                                            } else { if ((uint64_t)(y >> 64)) ovf = 1; x = (uint64_t) y; }
                                        }
                                        fin[q] = x; fneg[q] = s_ < 0;
                                    }
                                }
                            }
                        }
                        if (slip_ballot(ovf)) reject = 2;
                        if (!reject) {
                            uint64_t Lex = 0; int nLl = 0;
                            uint64_t kmin = ~0ull;
                            for (int q = 0; q < 4 && 64 * q < nst; q++) {
                                const int t = lane + 64 * q;
                                if (isL[q]) { est_vlo[t] = (uint32_t) fin[q]; est_vhi[t] = (uint32_t)(fin[q] >> 32); est_meta[t] = (fneg[q] ? 0x80000000u : 0u) | (uint32_t) j; }
                                nLl += slip_popc64(slip_ballot(isL[q])); Lex += (uint64_t) slip_popc64(slip_ballot(isL[q] && fin[q] != 0));
                                const uint64_t key = (isL[q] && fin[q]) ? (kind == 0 ? fin[q] : ~fin[q]) : ~0ull;
                                if (key < kmin) kmin = key;
                            }
                            const uint32_t mh = slip_wave_min_u32((uint32_t)(kmin >> 32));
                            const uint32_t ml = slip_wave_min_u32((uint32_t)(kmin >> 32) == mh ? (uint32_t) kmin : 0xFFFFFFFFu);
                            const uint64_t mk = ((uint64_t) mh << 32) | ml;
                            if (mk == ~0ull) reject = 2;            /* no nonzero non-pivotal row: the worker reports the singular column */
                            else {
                                /* equal values: the earlier pattern position wins (slip_get_smallest_pivot.c:79) */
                                uint32_t bpos = BIG; int bq = -1;
                                for (int q = 0; q < 4 && 64 * q < nst; q++) {
                                    const int t = lane + 64 * q;
                                    const uint64_t key = (isL[q] && fin[q]) ? (kind == 0 ? fin[q] : ~fin[q]) : ~0ull;
                                    if (key == mk) { const uint32_t p = (uint32_t) pinvm[est_row[t]]; if (p < bpos) { bpos = p; bq = q; } }
                                }
                                const uint32_t bp = slip_wave_min_u32(bpos);
                                const uint64_t bmk = slip_ballot(bq >= 0 && bpos == bp);
                                const int bl = slip_ctz64(bmk);
                                pt = bl + 64 * (int) slip_readlane((uint32_t)(bq < 0 ? 0 : bq), bl);
                                /* the diagonal preference on the final values (slip_get_pivot.c:68-76, 89-118, 126-146) */
                                if (diagpref && est_row[pt] != (uint32_t) col) {
                                    int dt = -1;
                                    if (lane == 0) dt = est_lookup((uint32_t) col);
                                    dt = (int) slip_bcast0_u32((uint32_t) dt);
                                    if (dt >= 0 && !((est_meta[dt] >> 30) & 1u)) {
                                        const uint64_t dv = (uint64_t) est_vlo[dt] | ((uint64_t) est_vhi[dt] << 32);
                                        const uint64_t bv = (uint64_t) est_vlo[pt] | ((uint64_t) est_vhi[pt] << 32);
                                        if (dv != 0) {
                                            if (scheme == 1 || P.tol_mode == 0) pt = dt;
                                            else {
                                                const int tk = slip_tol_small(P.tol_m, P.tol_e, scheme == 3 ? bv : dv, scheme == 3 ? dv : bv);
                                                if (tk < 0) reject = 2; else if (tk) pt = dt;
                                            }
                                        }
                                    }
                                }
                            }
                            /* E5: the sums of stage 1 */
                            if (!reject) {
                                nUc_all = (int) cb[3] + nlate; U_l = (uint64_t) cb[4] + (uint64_t) ulate; nLc = nLl;
                                Lb_total = Lex; plimbs = 1; lalloc = 1; poff = Lnl_;
                                if (Lnz_ + nLc > P.Lcap_nz || Lnl_ + (int64_t) Lb_total > P.Lcap_nl) reject = 2;
                                if (Unz_ + nUc_all + 1 > P.Ucap_nz || Unl_ + (int64_t)(U_l + plimbs) > P.Ucap_nl) reject = 2;
                            }
                            if (!reject) {
                                const uint64_t pv = (uint64_t) est_vlo[pt] | ((uint64_t) est_vhi[pt] << 32);
                                e_pivrow = (int) est_row[pt]; e_pivpos = (int) pinvm[est_row[pt]];
                                pneg = (int)(est_meta[pt] >> 31); pbits = slip_bits64(pv); lp_ = (pbits + 31) >> 5; plo = pv;
                                if (lane < 2) { const uint32_t dgt = (uint32_t)(pv >> (32 * lane)); sl[lane] = dgt; Ms[lane] = dgt; }
                                nfin = nst;
                                /* the rows go back to the worker: values final, positions as the reference has them at column j (the
                                 * stores are issued here and drained with the batch) */
                                uint32_t *hb = mbx + SLIP_MBOX_HDR;
                                for (int q = 0; q < 4 && 64 * q < nst; q++) {
                                    const int t = lane + 64 * q;
                                    if (t < nst) {
                                        const uint32_t m = est_meta[t];
                                        slip_st_u32(hb + t, est_row[t]); slip_st_u32(hb + SLIP_ENG_ROWS + t, est_vlo[t]); slip_st_u32(hb + 2 * SLIP_ENG_ROWS + t, est_vhi[t]);
                                        slip_st_u32(hb + 3 * SLIP_ENG_ROWS + t, (m & 0xC0000000u) | (uint32_t) pinvm[est_row[t]]);
                                    }
                                }
                                /* L(:,j) as the later columns of the run will read it: into the ring */
                                {
                                    uint32_t at0 = lw;
                                    for (int q = 0; q < 4 && 64 * q < nst; q++) {
                                        const int t = lane + 64 * q;
                                        const uint64_t lm_ = slip_ballot(isL[q]);
                                        if (isL[q]) {
                                            const uint32_t at = 3u * ((at0 + (uint32_t) slip_popc64(lm_ & ((1ull << lane) - 1ull))) % (uint32_t) SLIP_LRING);
                                            lring[at] = est_row[t] | (fneg[q] ? 0x80000000u : 0u); lring[at + 1] = (uint32_t) fin[q]; lring[at + 2] = (uint32_t)(fin[q] >> 32);
                                        }
                                        at0 += (uint32_t) slip_popc64(lm_);
                                    }
                                    if (lane == 0) { ld_start[j & (SLIP_CB_RING - 1)] = lw; ld_cnt[j & (SLIP_CB_RING - 1)] = (uint32_t) nLc; ld_col[j & (SLIP_CB_RING - 1)] = (uint32_t) j; }
                                    lw += (uint32_t) nLc;
                                }
                            }
                        }
                    }
                    /* the row -> place map is the next column's: leave it empty */
                    slip_wave_sync_lds();
                    for (int q = 0; q < 4 && 64 * q < nst; q++) { const int t = lane + 64 * q; if (t < nst) slotm[est_row[t]] = 0; }
                    SLIP_CT(19);
                    if (lane == 0) sv[C_LW] = (int32_t) lw;                                                   /* (columns fetched from memory stay in the ring also when this package goes back) */
                    if (reject && lane == 0) slip_st_u32(mbx + SLIP_PKG_OUT + 1, (uint32_t) reject);      /* why it goes back (1: try again later) */
                }
                slip_wave_sync_lds();
#if defined(SLIP_EMULATE) && defined(SLIP_EMU_TRACE)
                if (lane == 0) fprintf(stderr, "committer: col %d kind %d reject %d pre %d (stamp0 %d nfull %d nrows %d nlate %d pr0 %d ring0 %d)\n", j, kindp, reject, (int) cb[18], stamp0, (int) cb[22], nrows, nlate, sv[C_PR0], sv[C_RING0]);
#endif
                if (reject) { rej = j; break; }
                /* ---- the column is committed: what the next one needs stays in registers; the rings and the publish record in LDS ---- */
                {
                    SlipPiv pr; pr.off = poff; pr.len = pneg ? -lp_ : lp_; pr.bits = pbits; pr.invlen = 0; pr.pad = 0;
                    pr.lo = plo;
                    int z = 0;
                    if (lp_ <= 2) z = slip_ctz64(pr.lo);
                    pr.ctz = z; pr.inv64 = lp_ <= 2 ? slip_inv64(pr.lo >> z) : 0;      /* (ctz of a long pivot is found while its digits are published; nobody reads it here) */
                    const int64_t nUnz = Unz_ + nUc_all + 1, nLnz = Lnz_ + nLc;
                    const int64_t nUnl = Unl_ + (int64_t)(U_l + plimbs), nLnl = Lnl_ + (int64_t) Lb_total;
                    M = pr; Lnz_ = nLnz; Lnl_ = nLnl; Unz_ = nUnz; Unl_ = nUnl;
                    if (lane == i) { bs_row = (uint32_t) e_pivrow; bs_disp = (uint32_t) intermed2; bs_opos = (uint32_t) e_pivpos; }
#if defined(SLIP_EMULATE) && defined(SLIP_EMU_TRACE)
                    if (lane == 0) fprintf(stderr, "commit col %d kind %d: pivot row %d from pos %d, row %d goes there\n", j, kindp, e_pivrow, e_pivpos, intermed2);
#endif
                    if (lane == 0) {
                        pring_put(j, pr);
                        ring_row[j & (SLIP_CB_RING - 1)] = (uint32_t) e_pivrow; ring_disp[j & (SLIP_CB_RING - 1)] = (uint32_t) intermed2;
                        ring_opos[j & (SLIP_CB_RING - 1)] = (uint32_t) e_pivpos;
                        if (mirror) { pinvm[e_pivrow] = (uint16_t) j; pinvm[intermed2] = (uint16_t) e_pivpos; }
                        if (kindp == 1) {
                            eacc[0] += ec_src; eacc[1] += ec_read; eacc[2] += ec_str; eacc[3] += ec_upd; eacc[4] += ec_mac;
                            eacc[5] += 1ull; eacc[6] += (unsigned long long) nlate;
                        }
                    }
                    /* the publish record: lane w writes word w */
                    {
                        uint32_t v = 0;
                        const uint32_t wk_ = (uint32_t) HF(20);
                        switch (lane) {
                            case 0: v = (uint32_t) e_pivrow; break; case 1: v = (uint32_t) e_pivpos; break; case 2: v = (uint32_t) intermed2; break;
                            case 3: v = (uint32_t)(pneg ? -lp_ : lp_); break; case 4: v = (uint32_t) pbits; break; case 5: v = (uint32_t) lp_; break;
                            case 6: v = (uint32_t) nfin; break; case 7: v = (uint32_t) nlate; break;
                            case 8: v = (uint32_t) poff; break; case 9: v = (uint32_t)((uint64_t) poff >> 32); break;
                            case 10: v = (uint32_t) lalloc; break; case 11: v = (uint32_t)(lalloc >> 32); break;
                            case 12: v = (uint32_t) nUnz; break; case 13: v = (uint32_t)((uint64_t) nUnz >> 32); break;
                            case 14: v = (uint32_t) nLnz; break; case 15: v = (uint32_t)((uint64_t) nLnz >> 32); break;
                            case 16: v = (uint32_t) nUnl; break; case 17: v = (uint32_t)((uint64_t) nUnl >> 32); break;
                            case 18: v = (uint32_t) nLnl; break; case 19: v = (uint32_t)((uint64_t) nLnl >> 32); break;
                            case 20: v = wk_; break; case 21: v = (uint32_t) kindp; break;
                            default: break;
                        }
                        if (lane < 22) pb[lane] = v;
                    }
                    /* the permutation swap (slip_get_pivot.c:164-176) is stored HERE, by this one wave, column after column:
                     * successive columns of a batch write the same words (a row displaced to position p, the next pivot taken from
                     * p), and only stores of one wave to one address keep their order */
                    {
                        uint32_t *a4 = (uint32_t *) 0; uint32_t v4 = 0;
                        switch (lane) {
                            case 0: a4 = (uint32_t *) P.row_perm.at(j); v4 = (uint32_t) e_pivrow; break;
                            case 1: a4 = (uint32_t *) P.row_perm.at(e_pivpos); v4 = (uint32_t) intermed2; break;
                            case 2: a4 = (uint32_t *) P.pinv.at(e_pivrow); v4 = (uint32_t) j; break;
                            case 3: a4 = (uint32_t *) P.pinv.at(intermed2); v4 = (uint32_t) e_pivpos; break;
                            default: break;
                        }
                        /* (lanes of ONE store instruction to one address have no order: when the pivot already sits at position j,
                         * lanes 0/1 and 2/3 write identical values) */
#ifndef SLIP_EMU_BUG_PERM
                        if (a4) slip_st_u32(a4, v4);
#else
                        (void) a4; (void) v4;      /* test build: the round-3 race put back (the swaps stored by the publishing waves below) */
#endif
                    }
                    slip_wave_sync_lds();
                }
                SLIP_CT(16);
                nbc = i + 1;
            }
#undef HF
            const uint32_t lastpr_ = slip_readlane(bs_row, nbc > 0 ? nbc - 1 : 0);
            if (lane == 0) {
                sv[C_NBC] = nbc; sv[C_ST] = rej;
                if (nbc > 0) {
                    *Mrec = M; sv64[SV_LNZ / 2] = Lnz_; sv64[SV_LNL / 2] = Lnl_; sv64[SV_UNZ / 2] = Unz_; sv64[SV_UNL / 2] = Unl_;
                    sv[C_LASTPR] = (int32_t) lastpr_;
                    if (kc + nbc - sv[C_RING0] > SLIP_CB_RING) sv[C_RING0] = kc + nbc - SLIP_CB_RING;
                    if (kc + nbc - sv[C_PR0] > SLIP_CB_RING) sv[C_PR0] = kc + nbc - SLIP_CB_RING;
                }
            }
        }
        slip_block_sync();
        SLIP_CT(3);                                  /* 3: the serial part */
        const int nbc = sv[C_NBC], rej = sv[C_ST];
        /* (d) stage 1 of the committed columns, side by side: the pivot's digits written through, its record, the swap and its
         *     log, the column pointers, the outcome for the worker */
        for (int i = wave; i < nbc; i += nw) {
            const int j = kc + i;
            const uint32_t *pb = pub + i * SLIP_PUBW;
            const dig_t *src = stage + i * SLIP_CB_SLOTW;
            uint32_t *mbx = P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) pb[20] * SLIP_MBOX_WORDS;
            const int e_pivrow = (int) pb[0], e_pivpos = (int) pb[1], intermed2 = (int) pb[2], lp_ = (int) pb[5];
            const int64_t poff = (int64_t)((uint64_t) pb[8] | ((uint64_t) pb[9] << 32));
            const int z = slip_publish_digits((dig_t *)(P.Llimbs + poff), src, 0, lp_);
            SlipPiv pr; pr.off = poff; pr.len = (int32_t) pb[3]; pr.bits = (int32_t) pb[4]; pr.ctz = z; pr.invlen = 0;
            pr.lo = *(const uint64_t *) src; pr.inv64 = 0; pr.pad = 0;
            if (lp_ <= 2) pr.inv64 = slip_inv64(pr.lo >> z);
            /* every lane computes the same values; lane q issues store q: two store instructions instead of twenty-odd */
            {
                uint64_t *a8 = (uint64_t *) 0; uint64_t v8 = 0;
                uint64_t *pw = (uint64_t *) P.piv.at(j);
                switch (lane) {
                    case 0: a8 = pw; v8 = (uint64_t) pr.off; break;
                    case 1: a8 = pw + 1; v8 = (uint64_t)(uint32_t) pr.len | ((uint64_t)(uint32_t) pr.bits << 32); break;
                    case 2: a8 = pw + 2; v8 = pr.lo; break;
                    case 3: a8 = pw + 3; v8 = (uint64_t)(uint32_t) pr.ctz; break;
                    case 4: a8 = pw + 4; v8 = pr.inv64; break;
                    case 5: a8 = (uint64_t *) &P.Up[j + 1]; v8 = (uint64_t) pb[12] | ((uint64_t) pb[13] << 32); break;
                    case 6: a8 = (uint64_t *) &P.Lp[j + 1]; v8 = (uint64_t) pb[14] | ((uint64_t) pb[15] << 32); break;
                    case 7: a8 = (uint64_t *) &P.Uo[j + 1]; v8 = (uint64_t) pb[16] | ((uint64_t) pb[17] << 32); break;
                    case 8: a8 = (uint64_t *) &P.Lo[j + 1]; v8 = (uint64_t) pb[18] | ((uint64_t) pb[19] << 32); break;
                    case 9: a8 = (uint64_t *)(mbx + SLIP_PKG_OUT + 6); v8 = (uint64_t) poff; break;
                    case 10: a8 = (uint64_t *)(mbx + SLIP_PKG_OUT + 8); v8 = (uint64_t) pb[10] | ((uint64_t) pb[11] << 32); break;
                    case 11: a8 = pw + 5; v8 = 0ull; break;                   /* (nobody has divided by this pivot yet) */
                    default: break;
                }
                if (a8) slip_st_u64(a8, v8);
                uint32_t *a4 = (uint32_t *) 0; uint32_t v4 = 0;
                switch (lane) {
#ifdef SLIP_EMU_BUG_PERM
                    case 0: a4 = (uint32_t *) P.row_perm.at(j); v4 = (uint32_t) e_pivrow; break;
                    case 1: a4 = (uint32_t *) P.row_perm.at(e_pivpos); v4 = (uint32_t) intermed2; break;
                    case 2: a4 = (uint32_t *) P.pinv.at(e_pivrow); v4 = (uint32_t) j; break;
                    case 3: a4 = (uint32_t *) P.pinv.at(intermed2); v4 = (uint32_t) e_pivpos; break;
#endif
                    case 4: a4 = (uint32_t *) P.sw_row.at(j); v4 = (uint32_t) intermed2; break;
                    case 5: a4 = (uint32_t *) P.sw_pos.at(j); v4 = (uint32_t) e_pivpos; break;
                    case 6: a4 = mbx + SLIP_PKG_OUT + 1; v4 = (uint32_t) e_pivrow; break;
                    case 7: a4 = mbx + SLIP_PKG_OUT + 2; v4 = (uint32_t) e_pivpos; break;
                    case 8: a4 = mbx + SLIP_PKG_OUT + 3; v4 = pb[3]; break;
                    case 9: a4 = mbx + SLIP_PKG_OUT + 4; v4 = pb[4]; break;
                    case 10: a4 = mbx + SLIP_PKG_OUT + 5; v4 = pb[6]; break;
                    case 11: a4 = mbx + SLIP_PKG_OUT + 10; v4 = pb[7]; break;
                    default: break;
                }
                if (a4) slip_st_u32(a4, v4);
            }
        }
        slip_vm_drain();                             /* every wave's stores of this batch have left */
        slip_block_sync();
        SLIP_CT(4);                                  /* 4: publish + drain */
        /* (e) the verdicts and the frontier */
        if (wave == 0) {
            slip_vm_drain();
            if (lane < nbc) slip_st_u32(P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t)(pub + lane * SLIP_PUBW)[20] * SLIP_MBOX_WORDS + SLIP_PKG_OUT, (hver[lane] << 24) | (uint32_t)(kc + lane + 1));
#ifdef SLIP_PROFILE_PHASES
            if (lane < nbc) P.dbg[18 * (int64_t) P.n + 6 * (int64_t)(kc + lane) + 2] = (int32_t) slip_realtime();  /* time line 2: committed by the committer */
#endif
            if (lane == 0) {
                if (rej >= 0) {
                    const uint32_t rv = hver[rej - kc];
                    const uint32_t rw = (cbuf + (rej - kc) * SLIP_CBW)[20];
                    if (rw < (uint32_t) P.nworkers) slip_st_u32(P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) rw * SLIP_MBOX_WORDS + SLIP_PKG_OUT, (uint32_t)(-(int32_t)((rv << 24) | (uint32_t)(rej + 1))));
                    sv[C_REJ] = rej; sv[C_REJV] = (int32_t) rv;
                }
                if (nbc > 0) {
                    slip_st_frontier(st, kc + nbc, sv[C_LASTPR]);
                    slip_agent_add_u64(&st->c_short, (unsigned long long) nbc | ((unsigned long long) nbc << 32));      /* high word: by the committer */
                    sv[C_K] = kc + nbc; sv[C_HAVE] = 1;
                }
            }
        }
        SLIP_CT(5);                                  /* 5: verdicts + frontier */
#ifdef SLIP_PROFILE_COMMIT
        if (tid == 0) { tacc_[6] += 1; tacc_[7] += (unsigned long long) nbc; tacc_[8] += rej >= 0; tacc_[9] += (unsigned long long) nb; }
#endif
        slip_block_sync();
    }
    /* the engine's share of the algorithmic counters: the sources it applied in place of the workers */
    if (tid == 0) {
        if (eacc[0]) slip_agent_add_u64(&st->c_src, eacc[0]);
        if (eacc[1]) slip_agent_add_u64(&st->c_read, eacc[1]);
        if (eacc[2]) slip_agent_add_u64(&st->c_streamed, eacc[2]);
        if (eacc[3]) slip_agent_add_u64(&st->c_upd, eacc[3]);
        if (eacc[4]) slip_agent_add_u64(&st->c_macs, eacc[4]);
        if (eacc[5] || eacc[6]) slip_agent_add_u64(&st->c_eng, eacc[5] | (eacc[6] << 32));
    }
#ifdef SLIP_PROFILE_COMMIT
    if (tid == 0) for (int q = 0; q < 24; q++) st->prof[q] = tacc_[q];      /* (the workers' own slots are added on top: read the committer's with workers that do not stamp) */
#endif
}

#endif /* SLIP_REF_LU_PIPE_COMMIT_H */
