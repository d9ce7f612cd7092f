/* ref_lu_pipe_commit.h -- the committer: ONE workgroup of the launch that runs the commit chain of the column loop.
 *
 * The commit chain (choose the pivot of column k, publish rho_k and the row swap: slip_get_pivot.c:30-183 inside
 * SLIP_LU_factorize.c:190-264) is the serial part of the factorisation: column k's pivot search needs rho_{k-1}.  When
 * every column worker runs its own commit, each hop of the chain crosses the chip: the frontier word, rho_{k-1}'s record
 * and digits travel through memory, and the worker executes code that left its instruction cache milliseconds ago.  A
 * column whose rows were not touched by the last sources does not need any of that: what the pivot choice needs except
 * rho_{k-1} is known long before (slip_prepass) and fits a small PACKAGE: the pivot candidates (one-limb values never
 * updated, class S) and a few sums for the capacity checks.  The workers export such packages; the committer -- block 0,
 * a persistent loop over a few hundred instructions that stay in its instruction cache, with rho_{k-1}'s digits and the
 * slab cursors still in its LDS from the previous column -- multiplies the candidates, searches, applies the diagonal
 * rule, publishes stage 1 and moves the frontier; the worker is told the outcome and carries on with the bulk of its column
 * (pattern, remaining rows, L/U stores) as after its own early commit.  Columns without a valid package (a source arrived
 * late, long candidates, bounds too close to a capacity) are committed by their worker as before: the committer sees the
 * frontier pass and resynchronises.
 *
 * Package of column k: slot k % nworkers of P.pkg, SLIP_PKG_WORDS words:
 *   HDR   64-bit {k+1, version}; version odd: being written or retracted (seqlock: the committer reads the header, the
 *         package, and the header again)
 *   STAMP the frontier the worker has checked its rows against (none of row_perm[c], c < stamp, is a non-pivotal row of the
 *         package); the committer checks [stamp, k) itself
 *   SUMS  sv[SV_PP ..] of the pre-pass;  CAND 4 words per candidate: table index, value (2 words), aux (slot, digits, sign)
 *   ROWS  the rows of the pattern in discovery order;  POS their positions (pinv) at column k, written BY THE COMMITTER
 *   OUT   the outcome, written by the committer, polled by the worker: state k+1 committed / -(k+1) rejected, pivot row,
 *         its position, signed length, bits, slab offset, limbs handed out in the slab
 * The committer never writes into a worker's private memory and never stores a candidate's product in the slab (the worker
 * recomputes the few candidates with the rest of its rows): only the pivot's digits, its record, the swap, the column
 * pointers, the positions and the outcome leave the committer, all written through and drained before the frontier moves. */
#ifndef SLIP_REF_LU_PIPE_COMMIT_H
#define SLIP_REF_LU_PIPE_COMMIT_H

/* the worker's side: export the package the pre-pass has just prepared (all threads; barriers inside).  A column may
 * export again after a retraction: the version in the header, in the sums and in every candidate record tells the
 * committer which package it is looking at (it reads header and contents in separate rounds of loads). */
SLIP_DEV void slip_export_package(const SlipParams &P, const int k, uint32_t *lds, const int Fl)
{
    const int tid = slip_tid(), T = slip_nthreads();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    const uint32_t *f_row = lds + SLIP_LDS_TAB, *f_pos = f_row + SLIP_TAB_CAP, *f_aux = f_row + 3 * SLIP_TAB_CAP;
    const uint32_t *f_k0 = lds + SLIP_LDS_KEYS, *f_k1 = f_k0 + SLIP_PAT_CAP;
    const uint32_t *cl = lds + SLIP_LDS_WORK + SLIP_CAND_CAP;
    uint32_t *pk = P.pkg + (int64_t)(k % P.nworkers) * SLIP_PKG_WORDS;
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
        slip_st_u32(pk + SLIP_PKG_STAMP, (uint32_t) Fl); slip_st_u32(pk + SLIP_PKG_STAMP0, (uint32_t) Fl);
        slip_st_u32(pk + SLIP_PKG_NROWS, (uint32_t) nrows); slip_st_u32(pk + SLIP_PKG_VER, ver); slip_st_u32(pk + SLIP_PKG_WORKER, (uint32_t) P.worker);
        slip_st_u32(P.pkg + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS + SLIP_PKG_OUT, 0u);
    }
    slip_vm_drain();
    slip_block_sync();
    if (tid == 0) {
        slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t) ver << 32) | (uint32_t)(k + 1));
        sv[SV_PKGVER] = (int32_t) ver; sv[SV_PKGX] = 1;
    }
    slip_block_sync();
}

/* the worker's side: the package no longer describes the rows (thread 0) */
SLIP_DEV void slip_retract_package(const SlipParams &P, const int k, volatile int32_t *sv)
{
    uint32_t *pk = P.pkg + (int64_t)(k % P.nworkers) * SLIP_PKG_WORDS;
    const uint32_t ver = (uint32_t) sv[SV_PKGVER] | 1u;
    slip_st_u64((uint64_t *)(pk + SLIP_PKG_HDR), ((uint64_t) ver << 32) | (uint32_t)(k + 1));
    sv[SV_PKGVER] = (int32_t) ver; sv[SV_PKGX] = 0;
}

/* one candidate: the one-limb value a (nd digits) times rho[k-1] (in registers) -> LDS slot, search key, length */
template <int D> SLIP_DEV void slip_commit_mul(const WR<D> &Mr, uint32_t a0, uint32_t a1, int nd, dig_t *slotp, int kind, uint64_t *key_out, int *len_out)
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
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; if (c < ((len + 1) & ~1)) slotp[c] = Y.d[q]; }
    const uint32_t d1 = len ? wr_digit<D>(Y, len - 1) : 0u, d2 = len >= 2 ? wr_digit<D>(Y, len - 2) : 0u, d3 = len >= 3 ? wr_digit<D>(Y, len - 3) : 0u;
    uint64_t top = ((uint64_t) d1 << 32) | d2;
    const int sh = len ? slip_clz32(d1) : 0;
    if (sh) top = (top << sh) | (uint64_t)(d3 >> (32 - sh));
    const int bits = len ? 32 * len - sh : 0;
    uint64_t key = ((uint64_t) bits << 40) | (top >> 24);
    if (kind == 1) key = ~key;
    *key_out = key; *len_out = len;
}

/* LDS of the committer, words from lds + SLIP_LDS_WORK (the lists, tables and keys of a column worker: 12288 words) */
#ifndef SLIP_CB
#define SLIP_CB          8                  /* columns per batch */
#endif
#define SLIP_CB_RING     512                /* swaps the committer remembers (more than the columns in flight) */
#define SLIP_CBW         (32 + 6 * SLIP_PKG_CANDS + SLIP_PKG_NROWMAX)      /* one batch column: sums, candidates, rows */
#define SLIP_CB_SLOTW    262                /* a product of a one-limb value and a pivot of at most 256 digits, whole limbs */

/* the kernel body of the committer (block 0 of a launch with P.committer set).
 * Per batch: (a) wave 0 polls the headers of the next SLIP_CB columns; (b) one round of loads brings the ready packages
 * into LDS, every wave checks one column's rows against the pivots its worker has not seen; (c) wave 0 alone, on LDS
 * only, commits the columns one after the other: rows against the pivots of this batch, capacity checks, candidates'
 * positions from the swaps it remembers, products, search, diagonal rule, stage-1 stores ISSUED (not waited for);
 * (d) one drain, then the outcomes and the frontier. */
template <bool FAST>
SLIP_DEV void slip_committer(const SlipParams &P, SlipState *st, uint32_t *lds)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    volatile int64_t *sv64 = (volatile int64_t *)(lds + SLIP_LDS_VARS);
    uint32_t *base = lds + SLIP_LDS_WORK;
    dig_t *stage = base + SLIP_CB * SLIP_CBW;
    uint32_t *ring_row = stage + SLIP_PKG_CANDS * SLIP_CB_SLOTW, *ring_disp = ring_row + SLIP_CB_RING, *ring_opos = ring_disp + SLIP_CB_RING;
    uint32_t *ck0 = ring_opos + SLIP_CB_RING, *ck1 = ck0 + SLIP_PKG_CANDS, *clen = ck1 + SLIP_PKG_CANDS, *cpos = clen + SLIP_PKG_CANDS;
    const int wcap = P.wcap;
    dig_t *b0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap : P.gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    dig_t *Ms = lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + nw * 3 * wcap;      /* rho[j-1]'s digits (the committer runs with LDS scratch only) */
    SlipPiv *Mrec = (SlipPiv *)(lds + SLIP_LDS_SCAN);            /* rho[j-1]'s record */
    const int scheme = P.pivot_scheme;
    const int kind = (scheme == 4 || scheme == 5) ? 1 : 0;
    enum { C_K = SV_PP + SLIP_PP_WORDS, C_HAVE, C_GO, C_RING0, C_REJ, C_REJV, C_ST, C_LASTPR, C_IM2 };
    uint32_t *hver = cpos + SLIP_PKG_CANDS;                      /* the versions of the batch's packages as the poll saw them */
    if (tid == 0) {
        int pr_; sv[C_K] = slip_ld_frontier(st, &pr_);
        sv[C_HAVE] = 0; sv[C_RING0] = sv[C_K]; sv[C_REJ] = -1; sv[C_REJV] = 0;
    }
    slip_block_sync();
#ifdef SLIP_PROFILE_COMMIT
    unsigned long long tq_ = slip_realtime(), tacc_[24] = {0};
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
                    /* columns committed by their workers: their swaps from the log every publisher keeps */
                    for (int c = (kc - kprev > SLIP_CB_RING ? kc - SLIP_CB_RING : kprev) + lane; c < kc; c += SLIP_WAVE) {
                        ring_row[c & (SLIP_CB_RING - 1)] = (uint32_t) slip_ld_i32(&P.row_perm[c]);
                        ring_disp[c & (SLIP_CB_RING - 1)] = (uint32_t) slip_ld_i32(&P.sw_row[c]);
                        ring_opos[c & (SLIP_CB_RING - 1)] = (uint32_t) slip_ld_i32(&P.sw_pos[c]);
                    }
                    if (lane == 0 && kc - sv[C_RING0] > SLIP_CB_RING) sv[C_RING0] = kc - SLIP_CB_RING;
                    slip_wave_sync_lds();
                }
                const int64_t stop = sv64[SV_LEXACT / 2];
                if (kc >= P.k_stop || (stop >> 8) <= (int64_t) kc) { go = 0; break; }
                const int j = kc + lane;
                uint64_t h = 0;
                if (lane < SLIP_CB && j < P.k_stop && (stop >> 8) > (int64_t) j) h = slip_ld_u64((const uint64_t *)(P.pkg + (int64_t)(j % P.nworkers) * SLIP_PKG_WORDS + SLIP_PKG_HDR));
                const uint32_t hv = (uint32_t)(h >> 32);
                const int rdy = (uint32_t) h == (uint32_t)(j + 1) && hv >= 2u && !(hv & 1u) && !(j == sv[C_REJ] && hv == (uint32_t) sv[C_REJV]);
                if (lane < SLIP_CB) hver[lane] = hv;
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
        /* (b) the packages into LDS (one wave per column); what the previous batch did not leave behind */
        for (int i = wave; i < nb; i += nw) {
            const int j = kc + i;
            const uint32_t *pk = P.pkg + (int64_t)(j % P.nworkers) * SLIP_PKG_WORDS;
            uint32_t *cb = base + i * SLIP_CBW;
            if (lane < SLIP_PP_WORDS) cb[lane] = slip_ld_u32(pk + SLIP_PKG_SUMS + lane);
            else if (lane == 14) cb[14] = slip_ld_u32(pk + SLIP_PKG_STAMP);
            else if (lane == 15) cb[15] = slip_ld_u32(pk + SLIP_PKG_STAMP0);
            else if (lane == 16) cb[16] = slip_ld_u32(pk + SLIP_PKG_NROWS);
            else if (lane == 17) cb[17] = (uint32_t) slip_ld_i32(&P.row_perm[j]);
            else if (lane == 18) cb[18] = 0u;
            else if (lane == 19) cb[19] = slip_ld_u32(pk + SLIP_PKG_VER);
            else if (lane == 20) cb[20] = slip_ld_u32(pk + SLIP_PKG_WORKER);
            for (int c = lane; c < 6 * SLIP_PKG_CANDS; c += SLIP_WAVE) cb[32 + c] = slip_ld_u32(pk + SLIP_PKG_CAND + c);
            for (int c = lane; c < SLIP_PKG_NROWMAX; c += SLIP_WAVE) cb[32 + 6 * SLIP_PKG_CANDS + c] = slip_ld_u32(pk + SLIP_PKG_ROWS + c);
        }
        if (!have) {
            if (tid == T - 1) sv64[SV_LNZ / 2] = slip_ld_i64(&P.Lp[kc]);
            if (tid == T - 2) sv64[SV_LNL / 2] = slip_ld_i64(&P.Lo[kc]);
            if (tid == T - 3) sv64[SV_UNZ / 2] = slip_ld_i64(&P.Up[kc]);
            if (tid == T - 4) sv64[SV_UNL / 2] = slip_ld_i64(&P.Uo[kc]);
            if (tid == T - 5) *Mrec = slip_ld_piv(&P.piv[kc - 1]);       /* kc >= 1: column 0 is never packaged */
        }
        slip_block_sync();
        SLIP_CT(1);                                  /* 1: the packages into LDS */
        if (!have) {
            const SlipPiv M0 = *Mrec;
            const int l0 = slip_abs(M0.len);
            if (l0 <= wcap) { const dig_t *Mg = slip_piv_digits(P, M0); for (int c = tid; c < l0; c += T) Ms[c] = slip_ld_u32(Mg + c); }
        }
        /* the rows of every batch column against the pivots its worker has not seen (those committed before this batch) */
        for (int i = wave; i < nb; i += nw) {
            const int j = kc + i;
            uint32_t *cb = base + i * SLIP_CBW;
            const int nrows = (int) cb[16], ncand = (int) cb[1], stamp = (int) cb[14], stamp0 = (int) cb[15];
            int hit = nrows < 1 || nrows > SLIP_PKG_NROWMAX || ncand < 1 || ncand > SLIP_PKG_CANDS || stamp < stamp0 || stamp > j
                      || stamp0 < sv[C_RING0] || j - stamp0 > SLIP_CB_RING - SLIP_CB || cb[12] != 0 || cb[19] != hver[i] || cb[20] >= (uint32_t) P.nworkers;
            /* (a package being rewritten: its parts carry different versions -- it will be offered again) */
            if (!hit && lane < ncand && cb[32 + 6 * lane + 5] != hver[i]) hit = 1;
            if (!hit) {
                const uint32_t *rows = cb + 32 + 6 * SLIP_PKG_CANDS;
                uint32_t rr[SLIP_PKG_NROWMAX / SLIP_WAVE];
#pragma unroll
                for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) rr[q] = lane + 64 * q < nrows ? rows[lane + 64 * q] : 0xFFFFFFFFu;
                for (int c = stamp; c < kc; c++) {
                    const uint32_t r = ring_row[c & (SLIP_CB_RING - 1)];
#pragma unroll
                    for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) if (rr[q] == r) hit = 1;
                }
            }
            if (slip_ballot(hit) && lane == 0) cb[18] = 1u;
        }
        slip_block_sync();
        SLIP_CT(2);                                  /* 2: rows against the known pivots (and rho after a resynchronisation) */
        /* (c) the columns of the batch, one after the other, on LDS only: wave 0 prepares a column (rows against the pivots of
         *     this batch, capacity checks, the candidates' positions), all waves multiply its candidates, wave 0 searches and
         *     issues stage 1 */
        int nbc = 0, rej = -1;
        for (int i = 0; i < nb; i++) {
            const int j = kc + i, col = P.q[j];
            uint32_t *cb = base + i * SLIP_CBW;
            const uint32_t *cands = cb + 32, *rows = cb + 32 + 6 * SLIP_PKG_CANDS;
            uint32_t *pk = P.pkg + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t)(cb[20] < (uint32_t) P.nworkers ? cb[20] : 0u) * SLIP_MBOX_WORDS;     /* the worker's mailbox */
            const int nrows = (int) cb[16], ncand = (int) cb[1], stamp0 = (int) cb[15];
            const SlipPiv M = *Mrec;
            const int lm = slip_abs(M.len), brho = M.bits, slot = (lm + 3) >> 1;
            const int slotw = (lm + 5) & ~1;
            const int nS = (int) cb[2], nB = (int) cb[6];
            const uint32_t nUc_all = cb[3];
            const uint64_t U_l = (uint64_t) cb[4];
            const int64_t Lnz_ = sv64[SV_LNZ / 2], Lnl_ = sv64[SV_LNL / 2], Unz_ = sv64[SV_UNZ / 2], Unl_ = sv64[SV_UNL / 2];
            const int nA = lm > 2 ? nS : 0;
            const int maxc = (int) cb[9] - SLIP_PP_BIAS + brho;
            const int maxub_all = maxc > (int) cb[10] ? maxc : (int) cb[10];
            const uint64_t L_b = (uint64_t) cb[5] + (uint64_t) nB * (uint64_t)((brho + 63) >> 6) + (lm <= 2 ? 2ull * (uint64_t) nS : 0ull);
            const uint64_t preserve = (uint64_t)((maxub_all + 63) >> 6) + 1;
            const uint64_t Lb_total = (uint64_t) nA * (uint64_t) slot + preserve + L_b;
            const uint64_t Ub_total = U_l + preserve;
            const int nLc = nrows - (int) nUc_all;
            if (wave == 0) {
                int reject = (int) cb[18];
                if (!reject && i > 0) {
                    /* ... and against the pivots of this batch */
                    int hit = 0;
                    for (int q = 0; q < SLIP_PKG_NROWMAX / SLIP_WAVE; q++) {
                        const uint32_t rq = lane + 64 * q < nrows ? rows[lane + 64 * q] : 0xFFFFFFFFu;
                        for (int c = kc; c < j; c++) if (ring_row[c & (SLIP_CB_RING - 1)] == rq) hit = 1;
                    }
                    if (slip_ballot(hit)) reject = 1;
                }
                if (!reject) {
                    const bool A_ok = lm + 2 <= P.xcap && lm + 2 <= 256;
                    if (lm > wcap || slotw > SLIP_CB_SLOTW || (lm > 2 && !A_ok)) reject = 2;
                    if (nB > 0) {
                        const int Wn = (((int) cb[7] - SLIP_PP_BIAS + brho + 31) >> 5) + (((int) cb[8] + 31) >> 5) + 1;
                        if (Wn > P.wcap || Wn > P.xcap || Wn > P.invcap) reject = 2;
                    }
                    if (Lnz_ + nLc > P.Lcap_nz || Lnl_ + (int64_t) Lb_total > P.Lcap_nl) reject = 2;
                    if (Unz_ + (int) nUc_all + 1 > P.Ucap_nz || Unl_ + (int64_t) Ub_total > P.Ucap_nl) reject = 2;
                    if (P.limb_cap > 0 && (int)((maxub_all + 63) >> 6) > P.limb_cap) reject = 2;
                }
                if (!reject) {
                    /* products of a one-limb pivot: in the lane */
                    if (lm <= 2 && lane < ncand) {
                        const uint64_t xv = (uint64_t) cands[6 * lane + 1] | ((uint64_t) cands[6 * lane + 2] << 32);
                        const slip_u128 y = (slip_u128) xv * M.lo;
                        const int yb = slip_bits128(y), yl = (yb + 31) >> 5;
                        dig_t *sl = stage + lane * slotw;
                        sl[0] = (uint32_t) y; sl[1] = (uint32_t)(y >> 32); sl[2] = (uint32_t)(y >> 64); sl[3] = (uint32_t)(y >> 96);
                        const uint64_t top = yb ? (uint64_t)((y << (128 - yb)) >> 64) : 0ull;
                        uint64_t key = ((uint64_t) yb << 40) | (top >> 24);
                        if (kind == 1) key = ~key;
                        ck0[lane] = (uint32_t) key; ck1[lane] = (uint32_t)(key >> 32); clen[lane] = (uint32_t) yl;
                    }
                }
                if (lane == 0) sv[C_ST] = reject;
            } else if (wave == 1) {
                /* meanwhile, the second wave: the candidates' positions (pinv as the reference has it at column j) -- the value the
                 * worker read at frontier stamp0, or where the LAST swap since then that displaced the row put it.  Lanes look
                 * at the swaps, the candidates' rows come from the lanes that hold them. */
                const uint32_t myrow = lane < ncand ? rows[cands[6 * lane]] : 0xFFFFFFFFu;
                int last = -1;
                for (int e0 = stamp0; e0 < j; e0 += SLIP_WAVE) {
                    const int e = e0 + lane;
                    const uint32_t d = e < j ? ring_disp[e & (SLIP_CB_RING - 1)] : 0xFFFFFFFEu;
                    for (int c = 0; c < ncand; c++) {
                        const uint64_t m = slip_ballot(d == slip_readlane(myrow, c));
                        if (m && lane == c) last = e0 + 63 - slip_clz64(m);
                    }
                }
                if (lane < ncand) cpos[lane] = last >= 0 ? ring_opos[last & (SLIP_CB_RING - 1)] : cands[6 * lane + 4];
                /* the row at position j (the one the pivot changes places with): as loaded at the start of the batch, or the row a
                 * swap of this batch displaced to j */
                int intermed2 = (int) cb[17];
                {
                    const int e = kc + lane;
                    const uint64_t m = slip_ballot(e < j && (int) ring_opos[e & (SLIP_CB_RING - 1)] == j);
                    if (m) intermed2 = (int) ring_disp[(kc + 63 - slip_clz64(m)) & (SLIP_CB_RING - 1)];
                }
                if (lane == 0) sv[C_IM2] = intermed2;
            }
            SLIP_CT(10);
            slip_block_sync();
            SLIP_CT(11);
            if (sv[C_ST]) { rej = j; break; }
            /* products of a long pivot: one wave multiply per candidate with rho[j-1] in registers, the waves side by side */
            if (lm > 2) {
                const int Dm = (lm + 2 + 63) >> 6;
                for (int c = wave; c < ncand; c += nw) {
                    const uint32_t a0 = cands[6 * c + 1], a1 = cands[6 * c + 2]; const int nd = (int)((cands[6 * c + 3] >> 12) & 3u);
                    uint64_t key; int len;
                    if (Dm <= 1) slip_commit_mul<1>(wr_load<1>(Ms, lm), a0, a1, nd, stage + c * slotw, kind, &key, &len);
                    else if (Dm == 2) slip_commit_mul<2>(wr_load<2>(Ms, lm), a0, a1, nd, stage + c * slotw, kind, &key, &len);
                    else if (Dm == 3) slip_commit_mul<3>(wr_load<3>(Ms, lm), a0, a1, nd, stage + c * slotw, kind, &key, &len);
                    else slip_commit_mul<4>(wr_load<4>(Ms, lm), a0, a1, nd, stage + c * slotw, kind, &key, &len);
                    if (lane == 0) { ck0[c] = (uint32_t) key; ck1[c] = (uint32_t)(key >> 32); clen[c] = (uint32_t) len; }
                }
            }
            SLIP_CT(12);
            slip_block_sync();
            SLIP_CT(13);
            if (wave == 0) {
                const int intermed2 = sv[C_IM2];
                /* the search: (bit length, leading bits) keys; ties compared exactly, then by position (slip_get_smallest_pivot.c:79) */
                const uint64_t mykey = lane < ncand ? ((uint64_t) ck0[lane] | ((uint64_t) ck1[lane] << 32)) : ~0ull;
                const uint32_t mh = slip_wave_min_u32((uint32_t)(mykey >> 32));
                const uint32_t ml = slip_wave_min_u32((uint32_t)(mykey >> 32) == mh ? (uint32_t) mykey : 0xFFFFFFFFu);
                const uint64_t mk = ((uint64_t) mh << 32) | ml;
                const int kbits = (int)((kind == 0 ? mk : ~mk) >> 40);
                uint64_t tie = slip_ballot(lane < ncand && mykey == mk);
                int bc = -1;
                while (tie) {
                    const int l = slip_ctz64(tie); tie &= tie - 1;
                    if (bc < 0) { bc = l; continue; }
                    int cmp = 0;
                    if (kbits > 40) cmp = slip_cmp_mag(stage + bc * slotw, 0, stage + l * slotw, 0, (int) clen[l]);
                    if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0) || (cmp == 0 && cpos[l] < cpos[bc])) bc = l;
                }
                int est = bc < 0 ? SLIPDEV_INTERNAL : 0;
                if (bc < 0) bc = 0;
                /* the diagonal preference (slip_get_pivot.c:68-76, 89-118, 126-146); the worker listed the diagonal row when it is
                 * a nonzero non-pivotal row of the pattern */
                const int diag_t = (int) cb[13] - 1;
                if (!est && (scheme == 1 || scheme == 3 || scheme == 4) && diag_t >= 0 && (int) cands[6 * bc] != diag_t) {
                    const uint64_t dm = slip_ballot(lane < ncand && (int) cands[6 * lane] == diag_t);
                    const int dc = dm ? slip_ctz64(dm) : -1;
                    if (dc < 0) est = -1;                       /* not among the candidates it sent: the worker decides */
                    else if (scheme == 1 || P.tol_mode == 0) bc = dc;
                    else {
                        const uint64_t kb_ = (uint64_t) ck0[bc] | ((uint64_t) ck1[bc] << 32), kd_ = (uint64_t) ck0[dc] | ((uint64_t) ck1[dc] << 32);
                        const int kb = (int)((kind == 0 ? kb_ : ~kb_) >> 40), kd = (int)((kind == 0 ? kd_ : ~kd_) >> 40);
                        const int te0 = P.tol_e;
                        const int bnum_ = (scheme == 3 ? kb : kd) + (te0 < 0 ? -te0 : 0), bden_ = (scheme == 3 ? kd : kb) + (te0 > 0 ? te0 : 0);
                        int take = 0;
                        if (bnum_ < 52 + bden_) take = 0;
                        else if (bnum_ > 53 + bden_) take = 1;
                        else {
                            const int cn = scheme == 3 ? bc : dc, cd = scheme == 3 ? dc : bc;
                            const int ln = (int) clen[cn], ldn = (int) clen[cd];
                            if (ldn + 2 > wcap) est = -1;
                            else {
                                const int tk = slip_tol_compare_out(P.tol_m, P.tol_e, stage + cn * slotw, ln, stage + cd * slotw, ldn, b0, b1, b2, wcap);
                                if (tk < 0) est = -1; else take = tk;
                            }
                        }
                        if (take) bc = dc;
                    }
                }
                if (est) { if (est > 0 && lane == 0) slip_raise_stop(st, 0, SLIPDEV_INTERNAL); }
                else {
                    /* stage 1, issued and not waited for: the pivot's digits written through, its record, the swap and its log,
                     * the column pointers, the outcome for the worker */
                    const int e_pivrow = (int) rows[cands[6 * bc]], e_pivpos = (int) cpos[bc];
                    const uint32_t ax = cands[6 * bc + 3];
                    const int lp_ = (int) clen[bc];
                    const int neg = (int)((ax >> 14) & 1u) ^ (M.len < 0);
                    uint64_t key = (uint64_t) ck0[bc] | ((uint64_t) ck1[bc] << 32);
                    if (kind == 1) key = ~key;
                    const int pbits = (int)(key >> 40);
                    const int64_t poff = lm > 2 ? Lnl_ + (int64_t)(ax & 0x3FFu) * slot : Lnl_ + (int64_t) nA * slot;
                    const uint64_t plimbs = (uint64_t)((lp_ + 1) >> 1);
                    const uint64_t lalloc = (uint64_t) nA * (uint64_t) slot + (lm > 2 ? 0ull : plimbs);
                    const dig_t *src = stage + bc * slotw;
                    const int z = slip_publish_digits((dig_t *)(P.Llimbs + poff), src, 0, lp_);
                    {
                        SlipPiv pr; pr.off = poff; pr.len = neg ? -lp_ : lp_; pr.bits = pbits; pr.ctz = z; pr.invlen = 0;
                        pr.lo = *(const uint64_t *) src; pr.inv64 = 0; pr.pad = 0;
                        if (lp_ <= 2) pr.inv64 = slip_inv64(pr.lo >> z);
                        const int64_t nUnz = Unz_ + (int) nUc_all + 1, nLnz = Lnz_ + nLc;
                        const int64_t nUnl = Unl_ + (int64_t)(U_l + plimbs), nLnl = Lnl_ + (int64_t) Lb_total;
                        /* every lane computes the same values; lane q issues store q: two store instructions instead of twenty-odd */
                        {
                            uint64_t *a8 = (uint64_t *) 0; uint64_t v8 = 0;
                            uint64_t *pw = (uint64_t *) &P.piv[j];
                            switch (lane) {
                                case 0: a8 = pw; v8 = (uint64_t) pr.off; break;
                                case 1: a8 = pw + 1; v8 = (uint64_t)(uint32_t) pr.len | ((uint64_t)(uint32_t) pr.bits << 32); break;
                                case 2: a8 = pw + 2; v8 = pr.lo; break;
                                case 3: a8 = pw + 3; v8 = (uint64_t)(uint32_t) pr.ctz; break;
                                case 4: a8 = pw + 4; v8 = pr.inv64; break;
                                case 5: a8 = (uint64_t *) &P.Up[j + 1]; v8 = (uint64_t) nUnz; break;
                                case 6: a8 = (uint64_t *) &P.Lp[j + 1]; v8 = (uint64_t) nLnz; break;
                                case 7: a8 = (uint64_t *) &P.Uo[j + 1]; v8 = (uint64_t) nUnl; break;
                                case 8: a8 = (uint64_t *) &P.Lo[j + 1]; v8 = (uint64_t) nLnl; break;
                                case 9: a8 = (uint64_t *)(pk + SLIP_PKG_OUT + 6); v8 = (uint64_t) poff; break;
                                case 10: a8 = (uint64_t *)(pk + SLIP_PKG_OUT + 8); v8 = lalloc; break;
                                default: break;
                            }
                            if (a8) slip_st_u64(a8, v8);
                            uint32_t *a4 = (uint32_t *) 0; uint32_t v4 = 0;
                            switch (lane) {
                                case 0: a4 = (uint32_t *) &P.row_perm[j]; v4 = (uint32_t) e_pivrow; break;
                                case 1: a4 = (uint32_t *) &P.row_perm[e_pivpos]; v4 = (uint32_t) intermed2; break;
                                case 2: a4 = (uint32_t *) &P.pinv[e_pivrow]; v4 = (uint32_t) j; break;
                                case 3: a4 = (uint32_t *) &P.pinv[intermed2]; v4 = (uint32_t) e_pivpos; break;
                                case 4: a4 = (uint32_t *) &P.sw_row[j]; v4 = (uint32_t) intermed2; break;
                                case 5: a4 = (uint32_t *) &P.sw_pos[j]; v4 = (uint32_t) e_pivpos; break;
                                case 6: a4 = pk + SLIP_PKG_OUT + 1; v4 = (uint32_t) e_pivrow; break;
                                case 7: a4 = pk + SLIP_PKG_OUT + 2; v4 = (uint32_t) e_pivpos; break;
                                case 8: a4 = pk + SLIP_PKG_OUT + 3; v4 = (uint32_t)(neg ? -lp_ : lp_); break;
                                case 9: a4 = pk + SLIP_PKG_OUT + 4; v4 = (uint32_t) pbits; break;
                                default: break;
                            }
                            if (a4) slip_st_u32(a4, v4);
                        }
                        if (lane == 0) {
                            *Mrec = pr;
                            sv64[SV_LNZ / 2] = nLnz; sv64[SV_LNL / 2] = nLnl; sv64[SV_UNZ / 2] = nUnz; sv64[SV_UNL / 2] = nUnl;
                            ring_row[j & (SLIP_CB_RING - 1)] = (uint32_t) e_pivrow; ring_disp[j & (SLIP_CB_RING - 1)] = (uint32_t) intermed2;
                            ring_opos[j & (SLIP_CB_RING - 1)] = (uint32_t) e_pivpos;
                            if (j + 1 - sv[C_RING0] > SLIP_CB_RING) sv[C_RING0] = j + 1 - SLIP_CB_RING;
                            sv[C_LASTPR] = e_pivrow;
                        }
                    }
                    /* rho[j] for the next column: LDS to LDS */
                    if (lp_ <= wcap) for (int c = lane; c < ((lp_ + 1) & ~1); c += SLIP_WAVE) Ms[c] = src[c];
                }
                if (lane == 0) sv[C_ST] = est;
            }
            SLIP_CT(14);
            slip_block_sync();
            SLIP_CT(15);
            if (sv[C_ST]) { rej = j; break; }
            nbc = i + 1;
        }
        SLIP_CT(3);                                  /* 3: the serial part */
        /* (d) everything issued above has left; then the verdicts and the frontier */
        if (wave == 0) {
            slip_vm_drain();
            if (lane < nbc) slip_st_u32(P.pkg + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t)(base + lane * SLIP_CBW)[20] * SLIP_MBOX_WORDS + SLIP_PKG_OUT, (hver[lane] << 24) | (uint32_t)(kc + lane + 1));
#ifdef SLIP_PROFILE_PHASES
            if (lane < nbc) P.dbg[18 * (int64_t) P.n + 6 * (int64_t)(kc + lane) + 2] = (int32_t) slip_realtime();  /* time line 2: committed by the committer */
#endif
            if (lane == 0) {
                if (rej >= 0) {
                    const uint32_t rv = hver[rej - kc];
                    const uint32_t rw = (base + (rej - kc) * SLIP_CBW)[20];
                    if (rw < (uint32_t) P.nworkers) slip_st_u32(P.pkg + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) rw * SLIP_MBOX_WORDS + SLIP_PKG_OUT, (uint32_t)(-(int32_t)((rv << 24) | (uint32_t)(rej + 1))));
                    sv[C_REJ] = rej; sv[C_REJV] = (int32_t) rv;
                }
                if (nbc > 0) {
                    slip_st_frontier(st, kc + nbc, sv[C_LASTPR]);
                    slip_agent_add_u64(&st->c_short, (unsigned long long) nbc | ((unsigned long long) nbc << 32));      /* high word: by the committer */
                    sv[C_K] = kc + nbc; sv[C_HAVE] = 1;
                }
            }
        }
        SLIP_CT(4);                                  /* 4: the drain */
#ifdef SLIP_PROFILE_COMMIT
        if (tid == 0) { tacc_[5] += 1; tacc_[6] += (unsigned long long) nbc; tacc_[7] += rej >= 0; tacc_[8] += (unsigned long long) nb; }
#endif
        slip_block_sync();
    }
#ifdef SLIP_PROFILE_COMMIT
    if (tid == 0) for (int q = 0; q < 24; q++) st->prof[q] = tacc_[q];      /* (the workers' own slots are added on top: read the committer's with workers that do not stamp) */
#endif
}

#endif /* SLIP_REF_LU_PIPE_COMMIT_H */
