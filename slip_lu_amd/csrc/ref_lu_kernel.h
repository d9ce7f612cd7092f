/* ref_lu_kernel.h -- the left-looking REF sparse LU column loop on one gfx950 workgroup.
 *
 * Replaces, on the device, the reference's hot path
 *   SLIP_LU/Source/SLIP_LU_factorize.c:190-264      (column loop, L/U split)
 *   SLIP_LU/Source/slip_REF_triangular_solve.c:65-265 (reach + history/IPGE sweep)
 *   SLIP_LU/Source/slip_reach.c, slip_dfs.c, slip_sort_xi.c (pattern, order)
 *   SLIP_LU/Source/slip_get_pivot.c:30-183 and the smallest/largest/nonzero searches
 * with an MI355X-first formulation rather than a translation:
 *
 *  - No DFS and no sort.  Every edge of G(L) goes from a pivotal position to a
 *    strictly larger one, so the pattern is discovered by ONE ascending sweep
 *    over a bitmap indexed by pivot position (LDS): the next source is the next
 *    set bit below k; streaming that source's L column marks its rows.  The
 *    bitmap, read in order, IS the reference's sorted pattern xi[top..n).
 *  - One wavefront per IPGE update (target nonzero); the rows of a source's L
 *    column are dealt round-robin to the waves of the workgroup; indices and
 *    limbs stream coalesced from HBM, the big-integer intermediates live in LDS.
 *  - Exact divisions by pivots are 2-adic (wave_bigint.h); pivot inverses are
 *    computed lazily per pivot and cached in HBM.
 *  - Pivot search, row-permutation swap and the append of U(:,k), L(:,k) to the
 *    limb slabs happen on the device; the host only launches and grows buffers.
 *
 * Values are sign-magnitude: a signed digit count (32-bit digits) plus the
 * magnitude; slabs hold whole 64-bit limbs (odd digit counts are zero padded).
 */
#ifndef SLIP_REF_LU_KERNEL_H
#define SLIP_REF_LU_KERNEL_H

#include "wave_bigint.h"

/* status of a launch (SlipDev.status) */
enum {
    SLIPDEV_OK = 0,          /* reached k_stop                                      */
    SLIPDEV_SINGULAR = 1,    /* no eligible nonzero pivot in column status_k        */
    SLIPDEV_GROW_L = 2,      /* L slab / index arrays full; column status_k not done */
    SLIPDEV_GROW_U = 3,
    SLIPDEV_GROW_X = 4,      /* a value needs more than xcap / wcap / invcap digits  */
    SLIPDEV_WINDOW_END = 5   /* column status_k holds a value above limb_cap         */
};

typedef struct SlipDev {
    int32_t n, pivot_scheme, limb_cap, tol_mode;    /* tol_mode 0: tol <= 0          */
    uint64_t tol_m; int32_t tol_e, pad0;            /* tol = tol_m * 2^tol_e         */
    int32_t k_next, k_stop, status, status_k;
    /* A (CSC, duplicate-free columns) and the column order */
    const int64_t *Ap; const int32_t *Ai; const int32_t *Alen; const int64_t *Aoff;
    const uint64_t *Alimbs; const int32_t *q;
    /* row permutation and history */
    int32_t *pinv, *row_perm, *h;
    /* dense scatter vector x: row i at xd[i*xcap], signed digit count xlen[i] */
    uint32_t *xd; int32_t *xlen; int32_t xcap, invcap;
    /* pivots: rho[k] is the pivot entry of L(:,k) in the L slab */
    int64_t *rho_off; int32_t *rho_len, *rho_bits, *rho_ctz;
    uint32_t *invd; int32_t *invlen;
    /* factors under construction */
    int64_t *Lp; int32_t *Li, *Llen; int64_t *Loff; uint64_t *Llimbs; int64_t Lcap_nz, Lcap_nl, Lnz, Lnl;
    int64_t *Up; int32_t *Ui, *Ulen; int64_t *Uoff; uint64_t *Ulimbs; int64_t Ucap_nz, Ucap_nl, Unz, Unl;
    /* pattern of the current column (pivot positions, ascending) */
    int32_t *pat;
    /* scratch: 3 buffers of wcap digits per wave; LDS unless it does not fit */
    uint32_t *gscratch; int32_t wcap, scratch_in_lds;
    uint32_t *gbitmap; int32_t bm_words, bitmap_in_lds;
    /* algorithmic counters, committed columns only */
    unsigned long long c_upd, c_read, c_write, c_src, c_streamed, c_maxdig;
    /* diagnostic builds only (-DSLIP_PROFILE_PHASES): shader cycles per phase, thread 0 */
    unsigned long long prof[12];
} SlipDev;

#if defined(SLIP_PROFILE_PHASES) && !defined(SLIP_EMULATE)
#define SLIP_STAMP(slot) do { if (tid == 0) { unsigned long long now_ = clock64(); S->prof[slot] += now_ - t_prev_; t_prev_ = now_; } } while (0)
#define SLIP_STAMP_INIT() unsigned long long t_prev_ = clock64()
#else
#define SLIP_STAMP(slot) do { } while (0)
#define SLIP_STAMP_INIT() do { } while (0)
#endif

/* LDS layout in 32-bit words */
#define SLIP_LDS_VARS      0        /* 64 words of workgroup-shared scalars */
#define SLIP_LDS_SCAN      64       /* 64 words: per-wave partial sums (u64) */
#define SLIP_LDS_BITMAP    128

enum { SV_BEST = 0 /* 16 */, SV_ERR = 16, SV_PIVROW = 17, SV_PIVT = 18, SV_NU = 19, SV_NL = 20,
       SV_MAXDIG = 21, SV_DIAGOK = 22 };

/* ------------------------------------------------------------------ */
SLIP_DEV uint64_t slip_shfl_up_u64(uint64_t v, int d)
{
    int l = slip_lane(), s = l - d;
    return slip_shfl_u64(v, s < 0 ? l : s);
}

/* exclusive prefix sum of v over the workgroup's threads; *total = sum */
SLIP_DEV uint64_t slip_block_scan(uint64_t v, uint64_t *tmp, uint64_t *total)
{
    const int lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    uint64_t inc = v;
    for (int d = 1; d < SLIP_WAVE; d <<= 1) {
        uint64_t t = slip_shfl_up_u64(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == SLIP_WAVE - 1) tmp[wave] = inc;
    slip_block_sync();
    uint64_t base = 0, tot = 0;
    for (int w = 0; w < nw; w++) { uint64_t t = tmp[w]; if (w < wave) base += t; tot += t; }
    slip_block_sync();
    *total = tot;
    return base + inc - v;
}

/* next set bit of the position bitmap in [from, limit), or -1 (wave-cooperative) */
SLIP_DEV int slip_bitmap_next(const uint32_t *bm, int from, int limit)
{
    const int lane = slip_lane();
    if (from >= limit) return -1;
    int w = from >> 5;
    const int wend = (limit + 31) >> 5;
    int first = 1;
    while (w < wend) {
        int idx = w + lane;
        uint32_t word = idx < wend ? bm[idx] : 0u;
        if (first && lane == 0) word &= 0xFFFFFFFFu << (from & 31);
        if (idx == wend - 1 && (limit & 31)) word &= (1u << (limit & 31)) - 1u;
        uint64_t nz = slip_ballot(word != 0);
        if (nz) {
            int t = slip_ctz64(nz);
            uint32_t wv = slip_shfl_u32(word, t);
            return (w + t) * 32 + slip_ctz32(wv);
        }
        w += SLIP_WAVE; first = 0;
    }
    return -1;
}

SLIP_DEV int slip_sgn(int32_t slen) { return (slen > 0) - (slen < 0); }
SLIP_DEV int slip_abs(int32_t v) { return v < 0 ? -v : v; }

SLIP_DEV const dig_t *slip_rho(const SlipDev *S, int p) { return (const dig_t *)(S->Llimbs + S->rho_off[p]); }

/* make the cached inverse of pivot p's odd part valid modulo B^want.
 * b0,b1,b2: this wave's scratch.  Returns 0, or 1 if want exceeds the cache row. */
SLIP_DEV int slip_ensure_inv(SlipDev *S, int p, int want, dig_t *b0, dig_t *b1, dig_t *b2)
{
    /* another wave may publish a longer inverse at any time: one lane reads, all agree */
    int have = (int) slip_shfl_u32((uint32_t) *(volatile int32_t *) &S->invlen[p], 0);
    if (have >= want) return 0;
    if (want > S->invcap || want > S->wcap) return 1;
    int target = 2 * have > want ? 2 * have : want;
    if (target > S->invcap) target = S->invcap;
    if (target > S->wcap) target = S->wcap;
    const int ld = slip_abs(S->rho_len[p]);
    const int z = S->rho_ctz[p];
    int lodd = ld - (z >> 5);
    if (lodd > target) lodd = target;
    wb_copy_shr(b0, slip_rho(S, p), ld, z, lodd);
    dig_t *inv = S->invd + (int64_t) p * S->invcap;
    wb_inv_extend(inv, have, target, b0, lodd, b1, b2);
    slip_fence_block();
    if (slip_lane() == 0) slip_atomic_max_i32(&S->invlen[p], target);
    slip_wave_sync();
    return 0;
}

/* store a W-digit result (normalising) as row i of x; returns 1 if it does not fit */
SLIP_DEV int slip_store_x(SlipDev *S, int i, const dig_t *q, int W, int sign)
{
    const int lane = slip_lane();
    int len = wb_len(q, W);
    if (len > S->xcap) return 1;
    dig_t *X = S->xd + (int64_t) i * S->xcap;
    for (int c = lane; c < len; c += SLIP_WAVE) X[c] = q[c];
    if (lane == 0) S->xlen[i] = sign < 0 ? -len : len;
    slip_wave_sync();
    return 0;
}

/* History update of row r (slip_REF_triangular_solve.c:139-149, 248-257):
 *     x[r] <- x[r] * rho[pm] / rho[pd]      (pd < 0: no division)           */
SLIP_DEV int slip_history(SlipDev *S, int r, int pm, int pd, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int32_t xl = S->xlen[r];
    const int lx = slip_abs(xl);
    dig_t *X = S->xd + (int64_t) r * S->xcap;
    const int lm = slip_abs(S->rho_len[pm]);
    int sign = slip_sgn(xl) * slip_sgn(S->rho_len[pm]);
    int bq = wb_bits(X, lx) + S->rho_bits[pm];
    if (pd < 0) {
        int W = (bq + 31) >> 5;
        if (W > S->wcap) return 1;
        wb_mul_lo(b0, X, lx, slip_rho(S, pm), lm, W);
        return slip_store_x(S, r, b0, W, sign);
    }
    bq -= S->rho_bits[pd] - 1;
    const int W = (bq + 31) >> 5, zh = S->rho_ctz[pd], W2 = W + ((zh + 31) >> 5);
    if (W2 > S->wcap) return 1;
    if (slip_ensure_inv(S, pd, W, b0, b1, b2)) return 1;
    wb_mul_lo(b0, X, lx, slip_rho(S, pm), lm, W2);
    wb_copy_shr(b1, b0, W2, zh, W);
    wb_mul_lo(b2, b1, W, S->invd + (int64_t) pd * S->invcap, W, W);
    return slip_store_x(S, r, b2, W, sign * slip_sgn(S->rho_len[pd]));
}

/* One IPGE update (slip_REF_triangular_solve.c:156-241) of target row i by
 * source row j (pivot position jn) through the L entry m:
 *     x[i] <- ( hist(x[i]) * rho[jn] - L_m * x[j] ) / rho[jn-1]
 * hist() being the history update to level jn-1 when h[i] < jn-1.
 * One wavefront; everything modulo B^W (see wave_bigint.h).  Returns 1 when a
 * buffer is too small. */
SLIP_DEV int slip_ipge(SlipDev *S, int i, int j, int jn, int64_t m, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int32_t xl = S->xlen[i];
    const int lx = slip_abs(xl), sx = slip_sgn(xl);
    dig_t *X = S->xd + (int64_t) i * S->xcap;
    const int32_t rl = S->rho_len[jn];
    const int lr = slip_abs(rl), sr = slip_sgn(rl), br = S->rho_bits[jn];
    const int32_t ml = S->Llen[m];
    const int ll = slip_abs(ml), sl = slip_sgn(ml);
    const dig_t *Lm = (const dig_t *)(S->Llimbs + S->Loff[m]);
    const int32_t jl = S->xlen[j];
    const int lj = slip_abs(jl), sj = slip_sgn(jl);
    const dig_t *Xj = S->xd + (int64_t) j * S->xcap;
    const int hi = S->h[i];
    const int has_d = jn >= 1;
    int ld = 0, sd = 1, bd = 0, zd = 0;
    if (has_d) { const int32_t dl = S->rho_len[jn - 1]; ld = slip_abs(dl); sd = slip_sgn(dl); bd = S->rho_bits[jn - 1]; zd = S->rho_ctz[jn - 1]; }
    const int hist = lx != 0 && has_d && hi < jn - 1;
    const int hdiv = hist && hi > -1;
    int bh = 0, zh = 0, sh = 1;
    if (hdiv) { bh = S->rho_bits[hi]; zh = S->rho_ctz[hi]; sh = slip_sgn(S->rho_len[hi]); }

    /* bit bounds -> working widths */
    const int bx = wb_bits(X, lx);
    const int bxp = !lx ? 0 : (!hist ? bx : (hdiv ? bx + bd - bh + 1 : bx + bd));
    const int b1b = lx ? bxp + br : 0, b2b = wb_bits(Lm, ll) + wb_bits(Xj, lj);
    const int bnum = (b1b > b2b ? b1b : b2b) + 1;
    const int bq = has_d ? bnum - bd + 1 : bnum;
    const int W = (bq + 1 + 31) >> 5;                       /* + sign bit */
    const int W1 = W + (has_d ? ((zd + 31) >> 5) : 0);
    const int W2 = W1 + (hdiv ? ((zh + 31) >> 5) : 0);
    if (W2 > S->wcap) return 1;
    if (hdiv && slip_ensure_inv(S, hi, W1, b0, b1, b2)) return 1;
    if (has_d && slip_ensure_inv(S, jn - 1, W, b0, b1, b2)) return 1;

    /* P1 = hist(x_i) * rho_jn  -> b1, sign s1 */
    int s1 = sx * sr;
    if (lx == 0) {
        /* nothing: handled below */
    } else if (!hist) {
        wb_mul_lo(b1, X, lx, slip_rho(S, jn), lr, W1);
    } else if (!hdiv) {
        wb_mul_lo(b0, X, lx, slip_rho(S, jn - 1), ld, W1);
        wb_mul_lo(b1, b0, W1, slip_rho(S, jn), lr, W1);
        s1 *= sd;
    } else {
        wb_mul_lo(b0, X, lx, slip_rho(S, jn - 1), ld, W2);
        wb_copy_shr(b1, b0, W2, zh, W1);
        wb_mul_lo(b0, b1, W1, S->invd + (int64_t) hi * S->invcap, W1, W1);
        wb_mul_lo(b1, b0, W1, slip_rho(S, jn), lr, W1);
        s1 *= sd * sh;
    }
    /* P2 = L_m * x_j -> b2, sign s2 */
    const int s2 = sl * sj;
    wb_mul_lo(b2, Lm, ll, Xj, lj, W1);
    /* T = s1*P1 - s2*P2 = sT * (P1 -/+ P2)  -> b1 */
    int sT;
    if (lx == 0)       { wb_addsub(b1, (const dig_t *) 0, 0, b2, W1, W1, 1, 0u); sT = s2; }
    else if (s1 == s2) { wb_addsub(b1, b1, W1, b2, W1, W1, 1); sT = s1; }
    else               { wb_addsub(b1, b1, W1, b2, W1, W1, 0); sT = s1; }
    /* exact division by rho[jn-1] */
    dig_t *Q = b1;
    if (has_d) {
        wb_copy_shr(b0, b1, W1, zd, W);
        wb_mul_lo(b2, b0, W, S->invd + (int64_t)(jn - 1) * S->invcap, W, W);
        Q = b2; sT *= sd;
    }
    /* two's complement -> sign-magnitude */
    if (Q[W - 1] >> 31) {
        wb_addsub(Q, (const dig_t *) 0, 0, Q, W, W, 1, 0u);
        sT = -sT;
    }
    int rc = slip_store_x(S, i, Q, W, sT);
    if (slip_lane() == 0) S->h[i] = jn;
    return rc;
}

/* is |a| * 2^sa >= |b| * 2^sb ?  a, b normalised; scratch b0, b1 of wcap digits */
SLIP_DEV int slip_ge_shifted(const dig_t *a, int la, int sa, const dig_t *b, int lb, int sb,
                             dig_t *b0, dig_t *b1, int wcap, int *err)
{
    int ba = wb_bits(a, la) + sa, bb = wb_bits(b, lb) + sb;
    if (ba != bb) return ba > bb;
    int W = (ba + 31) >> 5;
    if (W > wcap) { *err = 1; return 0; }
    wb_copy_shl(b0, a, la, sa, W);
    wb_copy_shl(b1, b, lb, sb, W);
    return wb_cmp(b0, W, b1, W) >= 0;
}

/* ------------------------------------------------------------------ */
/* one column; returns a SLIPDEV_* status (0 = committed)              */
/* ------------------------------------------------------------------ */
SLIP_DEV int slip_do_column(SlipDev *S, const int k, uint32_t *lds,
                            unsigned long long *t_read, unsigned long long *t_upd,
                            unsigned long long *t_src, unsigned long long *t_str)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    const int n = S->n, col = S->q[k];
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint32_t *bm = S->bitmap_in_lds ? lds + SLIP_LDS_BITMAP : S->gbitmap;
    const int wcap = S->wcap;
    dig_t *b0 = S->scratch_in_lds ? lds + SLIP_LDS_BITMAP + (S->bitmap_in_lds ? S->bm_words : 0) + (int64_t) wave * 3 * wcap
                                  : S->gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    unsigned long long c_read = 0, c_upd = 0, c_src = 0, c_str = 0;

    SLIP_STAMP_INIT();
    /* ---- phase 0: clear the pattern bitmap ---- */
    for (int w = tid; w < S->bm_words; w += T) bm[w] = 0;
    if (tid == 0) { sv[SV_ERR] = 0; }
    slip_block_sync();

    /* ---- phase 1: scatter A(:,col) into x (slip_REF_triangular_solve.c:105-119) ---- */
    for (int64_t p = S->Ap[col] + tid; p < S->Ap[col + 1]; p += T) {
        const int row = S->Ai[p];
        const int pos = S->pinv[row];
        slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
        const int32_t al = S->Alen[p];
        const int la = slip_abs(al);
        const dig_t *src = (const dig_t *)(S->Alimbs + S->Aoff[p]);
        dig_t *X = S->xd + (int64_t) row * S->xcap;
        if (la > S->xcap) sv[SV_ERR] = 1;
        else for (int c = 0; c < la; c++) X[c] = src[c];
        S->xlen[row] = al;
        S->h[row] = -1;
        c_read += 4 + 8 * (unsigned long long)((la + 1) >> 1);
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;
    SLIP_STAMP(0);

    /* ---- phase 2: ascending sweep over the pivotal part of the pattern ---- */
    int cur = -1;
    for (;;) {
        slip_block_sync();
        const int jn = slip_bitmap_next(bm, cur + 1, k);
        if (jn < 0) break;
        cur = jn;
        const int j = S->row_perm[jn];
        if (wave == 0) {
            /* bring x[j] to its final value: history update to level jn-1 */
            if (S->xlen[j] != 0 && S->h[j] < jn - 1)
                if (slip_history(S, j, jn - 1, S->h[j], b0, b1, b2)) sv[SV_ERR] = 1;
        }
        slip_block_sync();
        const int32_t xjl = S->xlen[j];
        const int64_t m0 = S->Lp[jn], m1 = S->Lp[jn + 1];
        if (xjl != 0 && tid == 0) {
            c_src++;
            c_read += 8ull * ((slip_abs(S->rho_len[jn]) + 1) >> 1)
                    + (jn >= 1 ? 8ull * ((slip_abs(S->rho_len[jn - 1]) + 1) >> 1) : 0ull);
        }
        for (int64_t m = m0 + wave; m < m1; m += nw) {
            const int i = S->Li[m];
            const int inew = S->pinv[i];
            const int32_t ml = S->Llen[m];
            /* structural discovery (what the reference's DFS does) */
            int fresh = 0;
            if (lane == 0) {
                uint32_t bit = 1u << (inew & 31);
                uint32_t old = slip_atomic_or_u32(&bm[inew >> 5], bit);
                fresh = !(old & bit);
                if (fresh) { S->xlen[i] = 0; S->h[i] = -1; }
                if (xjl != 0) { c_str++; c_read += 4 + 8ull * ((slip_abs(ml) + 1) >> 1); }
            }
            slip_wave_sync();
            if (xjl != 0 && inew > jn && ml != 0) {
                if (lane == 0) c_upd++;
                if (slip_ipge(S, i, j, jn, m, b0, b1, b2)) { if (lane == 0) sv[SV_ERR] = 1; }
            }
        }
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;
    SLIP_STAMP(1);

    /* ---- phase 3: read the bitmap in order = the sorted pattern (slip_sort_xi.c) ---- */
    const int nwords = S->bm_words;
    const int per = (nwords + T - 1) / T;
    int w0 = tid * per, w1 = w0 + per; if (w1 > nwords) w1 = nwords;
    uint64_t cnt = 0;
    for (int w = w0; w < w1; w++) {
        uint32_t word = bm[w];
        uint32_t below;
        if ((w + 1) * 32 <= k) below = word;
        else if (w * 32 >= k) below = 0;
        else below = word & ((1u << (k - w * 32)) - 1u);
        cnt += ((uint64_t) slip_popc32(word) << 32) | (uint64_t) slip_popc32(below);
    }
    uint64_t tot;
    uint64_t ex = slip_block_scan(cnt, scan_tmp, &tot);
    {
        int o = (int)(ex >> 32);
        for (int w = w0; w < w1; w++) {
            uint32_t word = bm[w];
            while (word) { int b = slip_ctz32(word); word &= word - 1; S->pat[o++] = w * 32 + b; }
        }
    }
    const int npat = (int)(tot >> 32), nU = (int)(tot & 0xFFFFFFFFu), nL = npat - nU;
    slip_block_sync();
    SLIP_STAMP(2);

    /* ---- phase 4: history update of the non-pivotal rows to level k-1 (:248-257) ---- */
    if (k >= 1) {
        for (int t = wave; t < nL; t += nw) {
            const int r = S->row_perm[S->pat[nU + t]];
            if (S->xlen[r] != 0 && S->h[r] < k - 1)
                if (slip_history(S, r, k - 1, S->h[r], b0, b1, b2)) { if (lane == 0) sv[SV_ERR] = 1; }
        }
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;
    SLIP_STAMP(3);

    /* ---- phase 5: column-window cap, then the pivot search ---- */
    {
        int mx = 0;
        for (int t = tid; t < npat; t += T) { int l = slip_abs(S->xlen[S->row_perm[S->pat[t]]]); if (l > mx) mx = l; }
        uint64_t dummy;
        /* max via a scan-free reduction: reuse scan_tmp through atomics on LDS var */
        if (tid == 0) sv[SV_MAXDIG] = 0;
        slip_block_sync();
        if (mx > 0) slip_atomic_max_i32((int32_t *) &sv[SV_MAXDIG], mx);
        slip_block_sync();
        (void) dummy;
    }
    const int maxdig = sv[SV_MAXDIG];
    if (S->limb_cap > 0 && ((maxdig + 1) >> 1) > S->limb_cap) return SLIPDEV_WINDOW_END;

    /* kind of search: 0 smallest, 1 largest, 2 first nonzero (slip_get_pivot.c:58-155) */
    const int scheme = S->pivot_scheme;
    const int kind = (scheme == 2) ? 2 : ((scheme == 4 || scheme == 5) ? 1 : 0);
    int best = -1;                                   /* index t into the L part */
    for (int t = wave; t < nL; t += nw) {
        const int r = S->row_perm[S->pat[nU + t]];
        const int lr_ = slip_abs(S->xlen[r]);
        if (lr_ == 0) continue;
        if (best < 0) { best = t; continue; }
        if (kind == 2) continue;
        const int rb = S->row_perm[S->pat[nU + best]];
        const int c = wb_cmp(S->xd + (int64_t) rb * S->xcap, slip_abs(S->xlen[rb]), S->xd + (int64_t) r * S->xcap, lr_);
        if ((kind == 0 && c > 0) || (kind == 1 && c < 0)) best = t;
    }
    if (lane == 0) sv[SV_BEST + wave] = best;
    slip_block_sync();
    /* every wave reduces the per-wave candidates identically (ties -> smaller t) */
    best = -1;
    for (int w = 0; w < nw; w++) {
        const int t = sv[SV_BEST + w];
        if (t < 0) continue;
        if (best < 0) { best = t; continue; }
        if (kind == 2) { if (t < best) best = t; continue; }
        const int rb = S->row_perm[S->pat[nU + best]], r = S->row_perm[S->pat[nU + t]];
        const int c = wb_cmp(S->xd + (int64_t) rb * S->xcap, slip_abs(S->xlen[rb]), S->xd + (int64_t) r * S->xcap, slip_abs(S->xlen[r]));
        if ((kind == 0 && c > 0) || (kind == 1 && c < 0) || (c == 0 && t < best)) best = t;
    }
    if (best < 0) return SLIPDEV_SINGULAR;
    int pivrow = S->row_perm[S->pat[nU + best]];
    /* the diagonal preference (slip_get_pivot.c:68-76, 89-118, 126-146) */
    if (scheme == 1 || scheme == 3 || scheme == 4) {
        const int pc = S->pinv[col];
        const int diag_ok = pc >= k && ((bm[pc >> 5] >> (pc & 31)) & 1u) && S->xlen[col] != 0;
        if (diag_ok && pivrow != col) {
            int take = 0, err = 0;
            if (scheme == 1) take = 1;
            else if (S->tol_mode == 0) take = 1;
            else {
                const dig_t *num, *den; int ln, ldn;
                const dig_t *xp = S->xd + (int64_t) pivrow * S->xcap, *xc = S->xd + (int64_t) col * S->xcap;
                const int lp_ = slip_abs(S->xlen[pivrow]), lc_ = slip_abs(S->xlen[col]);
                if (scheme == 3) { num = xp; ln = lp_; den = xc; ldn = lc_; }   /* |small| / |diag| >= tol */
                else             { num = xc; ln = lc_; den = xp; ldn = lp_; }   /* |diag| / |large| >= tol */
                /* num >= tol_m * 2^tol_e * den */
                const int Wm = ldn + 2;
                if (Wm > wcap) err = 1;
                else {
                    if (lane == 0) { b2[0] = (uint32_t) S->tol_m; b2[1] = (uint32_t)(S->tol_m >> 32); }
                    slip_wave_sync();
                    wb_mul_lo(b0, b2, 2, den, ldn, Wm);
                    int lm_ = wb_len(b0, Wm);
                    /* copy product out of b0 because slip_ge_shifted uses b0,b1 */
                    for (int c = lane; c < lm_; c += SLIP_WAVE) b2[c] = b0[c];
                    slip_wave_sync();
                    const int te = S->tol_e;
                    take = slip_ge_shifted(num, ln, te < 0 ? -te : 0, b2, lm_, te > 0 ? te : 0, b0, b1, wcap, &err);
                }
            }
            if (err) return SLIPDEV_GROW_X;
            if (take) pivrow = col;
        }
    }
    const int pivpos = S->pinv[pivrow];               /* pre-swap position, >= k */
    SLIP_STAMP(4);

    /* ---- phase 6: append U(:,k) and L(:,k) (SLIP_LU_factorize.c:226-263) ---- */
    /* U(:,k): pattern rows below k in order, then the pivot.  L(:,k): rows at or above k in order. */
    const int nUe = nU + 1;
    uint64_t limbsU = 0, limbsL = 0;
    for (int t = tid; t < nUe; t += T) {
        const int r = t < nU ? S->row_perm[S->pat[t]] : pivrow;
        limbsU += (uint64_t)((slip_abs(S->xlen[r]) + 1) >> 1);
    }
    for (int t = tid; t < nL; t += T) limbsL += (uint64_t)((slip_abs(S->xlen[S->row_perm[S->pat[nU + t]]]) + 1) >> 1);
    uint64_t totU, totL;
    (void) slip_block_scan(limbsU, scan_tmp, &totU);
    (void) slip_block_scan(limbsL, scan_tmp, &totL);
    if (S->Unz + nUe > S->Ucap_nz || S->Unl + (int64_t) totU > S->Ucap_nl) return SLIPDEV_GROW_U;
    if (S->Lnz + nL > S->Lcap_nz || S->Lnl + (int64_t) totL > S->Lcap_nl) return SLIPDEV_GROW_L;

    /* offsets: tiles of T entries, running base */
    {
        int64_t base = S->Unl;
        for (int t0 = 0; t0 < nUe; t0 += T) {
            const int t = t0 + tid;
            int r = -1; uint64_t l = 0;
            if (t < nUe) { r = t < nU ? S->row_perm[S->pat[t]] : pivrow; l = (uint64_t)((slip_abs(S->xlen[r]) + 1) >> 1); }
            uint64_t tt; uint64_t e = slip_block_scan(l, scan_tmp, &tt);
            if (t < nUe) { const int64_t at = S->Unz + t; S->Ui[at] = r; S->Ulen[at] = S->xlen[r]; S->Uoff[at] = base + (int64_t) e; }
            base += (int64_t) tt;
        }
        base = S->Lnl;
        for (int t0 = 0; t0 < nL; t0 += T) {
            const int t = t0 + tid;
            int r = -1; uint64_t l = 0;
            if (t < nL) { r = S->row_perm[S->pat[nU + t]]; l = (uint64_t)((slip_abs(S->xlen[r]) + 1) >> 1); }
            uint64_t tt; uint64_t e = slip_block_scan(l, scan_tmp, &tt);
            if (t < nL) { const int64_t at = S->Lnz + t; S->Li[at] = r; S->Llen[at] = S->xlen[r]; S->Loff[at] = base + (int64_t) e; }
            base += (int64_t) tt;
        }
    }
    slip_block_sync();
    SLIP_STAMP(5);
    /* limbs: one wave per entry, coalesced */
    for (int t = wave; t < nUe + nL; t += nw) {
        const int isU = t < nUe;
        const int64_t at = isU ? S->Unz + t : S->Lnz + (t - nUe);
        const int r = isU ? S->Ui[at] : S->Li[at];
        dig_t *dst = isU ? (dig_t *)(S->Ulimbs + S->Uoff[at]) : (dig_t *)(S->Llimbs + S->Loff[at]);
        wb_copy_pad(dst, S->xd + (int64_t) r * S->xcap, slip_abs(S->xlen[r]));
    }
    slip_block_sync();
    SLIP_STAMP(6);
    /* pivot bookkeeping (slip_get_pivot.c:164-182); wave 0 */
    if (wave == 0) {
        const int64_t pat_at = S->Lnz + (int64_t) best;       /* valid only when pivrow was the searched best */
        int64_t at = pat_at;
        if (S->Li[at] != pivrow) {                            /* diagonal override: find it in L(:,k) */
            int found = -1;
            for (int t0 = 0; t0 < nL && found < 0; t0 += SLIP_WAVE) {
                int t = t0 + lane;
                uint64_t hit = slip_ballot(t < nL && S->Li[S->Lnz + t] == pivrow);
                if (hit) found = t0 + slip_ctz64(hit);
            }
            at = S->Lnz + found;
        }
        const int lp_ = slip_abs(S->Llen[at]);
        const dig_t *pv = (const dig_t *)(S->Llimbs + S->Loff[at]);
        const int z = wb_ctz(pv, lp_);
        const int bits = wb_bits(pv, lp_);
        if (lane == 0) {
            S->rho_off[k] = S->Loff[at]; S->rho_len[k] = S->Llen[at];
            S->rho_bits[k] = bits; S->rho_ctz[k] = z; S->invlen[k] = 0;
            const int intermed = pivpos, intermed2 = S->row_perm[k];
            S->row_perm[k] = pivrow; S->row_perm[intermed] = intermed2;
            S->pinv[pivrow] = k; S->pinv[intermed2] = intermed;
            S->Unz += nUe; S->Unl += (int64_t) totU; S->Lnz += nL; S->Lnl += (int64_t) totL;
            S->Up[k + 1] = S->Unz; S->Lp[k + 1] = S->Lnz;
            S->c_write += 4ull * (unsigned long long)(nUe + nL) + 8ull * (totU + totL) + 8ull * ((lp_ + 1) >> 1);
            if ((unsigned long long) maxdig > S->c_maxdig) S->c_maxdig = (unsigned long long) maxdig;
        }
    }
    *t_read += c_read; *t_upd += c_upd; *t_src += c_src; *t_str += c_str;
    slip_block_sync();
    SLIP_STAMP(7);
    return SLIPDEV_OK;
}

/* the kernel body: columns [k_next, k_stop) on ONE workgroup */
SLIP_DEV void slip_factor_columns(SlipDev *S, uint32_t *lds)
{
    unsigned long long t_read = 0, t_upd = 0, t_src = 0, t_str = 0;
    int k = S->k_next;
    const int k_stop = S->k_stop;
    int status = SLIPDEV_OK;
    slip_block_sync();
    for (; k < k_stop; k++) {
        status = slip_do_column(S, k, lds, &t_read, &t_upd, &t_src, &t_str);
        if (status != SLIPDEV_OK) break;
    }
    slip_block_sync();
    /* per-thread counters -> totals */
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint64_t a, b, c, d;
    (void) slip_block_scan(t_read, scan_tmp, &a);
    (void) slip_block_scan(t_upd, scan_tmp, &b);
    (void) slip_block_scan(t_src, scan_tmp, &c);
    (void) slip_block_scan(t_str, scan_tmp, &d);
    if (slip_tid() == 0) {
        S->c_read += a; S->c_upd += b; S->c_src += c; S->c_streamed += d;
        S->k_next = k; S->status = status; S->status_k = k;
    }
}

#endif /* SLIP_REF_LU_KERNEL_H */
