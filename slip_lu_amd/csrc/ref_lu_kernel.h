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
 *  - A source's L column is streamed coalesced, one entry per LANE: index,
 *    packed entry record, row state.  Updates whose operands fit one 64-bit
 *    limb are finished right there in the lane with 128-bit modular
 *    arithmetic; the others are queued in LDS and done one per WAVEFRONT with
 *    the limb-parallel primitives of wave_bigint.h (intermediates in LDS).
 *  - Exact divisions by pivots are 2-adic; pivot inverses are computed lazily
 *    per pivot and cached in HBM (the 64-bit inverse eagerly, in the pivot record).
 *  - Pivot search, row-permutation swap and the append of U(:,k), L(:,k) to the
 *    limb slabs happen on the device; the host only launches and grows buffers.
 *
 * Values are sign-magnitude: a signed digit count (32-bit digits) plus the
 * magnitude; all stores are padded to whole 64-bit limbs.
 */
#ifndef SLIP_REF_LU_KERNEL_H
#define SLIP_REF_LU_KERNEL_H

#include "wave_bigint.h"
#include "wave_bigint_reg.h"

typedef unsigned __int128 slip_u128;

/* status of a launch (SlipState.status) */
enum {
    SLIPDEV_OK = 0,          /* reached k_stop                                      */
    SLIPDEV_SINGULAR = 1,    /* no eligible nonzero pivot in column status_k        */
    SLIPDEV_GROW_L = 2,      /* L slab / index arrays full; column status_k not done */
    SLIPDEV_GROW_U = 3,
    SLIPDEV_GROW_X = 4,      /* a value needs more than xcap / wcap / invcap digits  */
    SLIPDEV_WINDOW_END = 5,  /* column status_k holds a value above limb_cap         */
    SLIPDEV_INTERNAL = 6     /* helper workgroups did not answer / inconsistent batch */
};

/* state of one row of the dense scatter vector x */
typedef struct { int32_t len, h, bits, pad; } SlipRow;            /* signed digits, history, bit length */
/* one stored entry of L or U: where its limbs are, how long, how many bits */
typedef struct { int64_t off; int32_t len, bits; } SlipEnt;       /* off in 64-bit limbs */
/* one pivot rho[k] (= the pivot entry of L(:,k)) */
typedef struct {
    int64_t off; int32_t len, bits;          /* limbs in the L slab */
    uint64_t lo; int32_t ctz, invlen;        /* low limb; trailing zero bits; cached inverse digits */
    uint64_t inv64, pad;                     /* inverse of the odd part modulo 2^64 (one-limb pivots) */
} SlipPiv;

/* immutable during a launch: passed by value (kernarg -> scalar loads) */
typedef struct SlipParams {
    int32_t n, pivot_scheme, limb_cap, tol_mode;    /* tol_mode 0: tol <= 0          */
    uint64_t tol_m; int32_t tol_e, k_stop;          /* tol = tol_m * 2^tol_e         */
    const int64_t *Ap; const int32_t *Ai; const int32_t *Alen; const int64_t *Aoff;
    const uint64_t *Alimbs; const int32_t *q;
    int32_t *pinv, *row_perm;
    SlipRow *xrow; uint32_t *xd;                    /* row i's digits at xd[i*xcap]  */
    SlipPiv *piv; uint32_t *invd;                   /* pivot p's inverse at invd[p*invcap] */
    int32_t xcap, invcap, wcap, bm_words;
    int64_t *Lp; int32_t *Li; SlipEnt *Le; uint64_t *Llimbs; int64_t Lcap_nz, Lcap_nl;
    int64_t *Up; int32_t *Ui; SlipEnt *Ue; uint64_t *Ulimbs; int64_t Ucap_nz, Ucap_nl;
    int32_t *pat;                                   /* pattern of the column: pivot positions, ascending */
    uint32_t *gscratch, *gbitmap;                   /* used when LDS does not hold them */
    /* helper workgroups (blocks 1..nhelpers): multi-limb updates of one source are farmed out */
    struct SlipBatch *batch; uint32_t *batch_items;
    int32_t nhelpers, fork_min;                     /* fork_min: queue length from which a batch is published */
    int32_t seq0, pad1;                             /* hand-off generation at launch (SlipState.seq)          */
    int32_t bitmap_in_lds, scratch_in_lds;          /* where the bitmap / wave scratch live (generic kernel)  */
    int32_t *dbg;                                   /* 4 words per workgroup: hand-off diagnostics            */
    const struct SlipParams *self;                  /* device copy of this struct: what out-of-line routines read */
} SlipParams;

/* one published batch of wave-level work (HBM; handed over with agent-scope release/acquire).
 * The three polled words live in cache lines of their own and are only ever touched with
 * agent-scope atomics (sc1): a plain store to the same line would leave a private copy in the
 * writer's XCD L2 that its own polls could then be served from. */
typedef struct SlipBatch {
    int32_t seq;  int32_t pad0[31];                 /* generation, bumped by the master to publish   */
    int32_t done; int32_t pad1[31];                 /* helper workgroups finished with this batch    */
    int32_t err;  int32_t pad2[31];                 /* first error a helper met (1 grow, 8 internal) */
    /* payload: plain stores before the release, plain loads after the acquire */
    int32_t kind, nitems;                           /* 0: leave; 1: IPGE updates (m-m0, i) pairs; 2: history rows */
    int32_t j, jn, k, stamp;                        /* stamp = the generation this payload belongs to */
    int64_t m0;
    int32_t stale_seen, pad4[23];                   /* diagnostics: payload re-reads a helper needed */
} SlipBatch;

/* arguments of the REF triangular solves (SLIP_LU_solve.c:41-86) on resident factors */
typedef struct SlipSolveArgs {
    int32_t nrhs, pad;
    const int32_t *blen; const int64_t *boff; const uint64_t *blimbs;   /* dense b, entry (c,i) at c*n+i: signed digits, limb offset */
    int32_t *olen; int64_t *ooff; uint64_t *olimbs; int64_t ocap;       /* numerators over det = rho[n-1], by pivot position        */
} SlipSolveArgs;

/* mutable across launches */
typedef struct SlipState {
    int32_t k_next, status, status_k;
    int32_t seq;                                    /* batch generation: monotonic across launches */
    int64_t Lnz, Lnl, Unz, Unl;                     /* Lnl counts allocated limbs (a direct row may leave one limb unused) */
    int64_t Lnl_exact;                              /* limbs actually stored in L                                          */
    int64_t out_used;                               /* solve: limbs of the output slab in use                              */
    int32_t solve_next, pad32;                      /* solve: next right-hand side                                         */
    unsigned long long c_upd, c_read, c_write, c_src, c_streamed, c_maxdig;
    unsigned long long prof[20];                    /* -DSLIP_PROFILE_PHASES builds only */
} SlipState;

#if defined(SLIP_PROFILE_PHASES) && !defined(SLIP_EMULATE)
#define SLIP_STAMP(slot) do { if (tid == 0) { unsigned long long now_ = clock64(); st->prof[slot] += now_ - t_prev_; t_prev_ = now_; } } while (0)
#define SLIP_PROFILING 1
#define SLIP_STAMP_INIT() unsigned long long t_prev_ = clock64()
#else
#define SLIP_STAMP(slot) do { } while (0)
#define SLIP_STAMP_INIT() do { } while (0)
#endif

/* LDS layout in 32-bit words */
#define SLIP_LDS_VARS      0        /* 64 words of workgroup-shared scalars          */
#define SLIP_LDS_SCAN      64       /* 128 words: per-wave partials of scans/reductions */
#define SLIP_LDS_WORK      192      /* work lists: 2 x SLIP_WORK_CAP (m, i) pairs, or 1 x rows + 1 x 4-word row records */
#define SLIP_WORK_CAP      1024
#define SLIP_WORK_WORDS    (6 * SLIP_WORK_CAP)
#define SLIP_LDS_TAB       (SLIP_LDS_WORK + SLIP_WORK_WORDS)   /* column table: row, len, bits, slab offset per pattern entry */
#define SLIP_TAB_CAP       1024
#define SLIP_PAT_CAP       1024     /* a pattern of at most this many entries stays in LDS               */
#define SLIP_LDS_PAT       (SLIP_LDS_TAB + 4 * SLIP_TAB_CAP)
#define SLIP_LDS_ROWS      (SLIP_LDS_PAT + SLIP_PAT_CAP)     /* row id of every pattern entry (same cap)        */
#define SLIP_LDS_DIROFF    (SLIP_LDS_ROWS + SLIP_PAT_CAP)    /* slab offset of a row multiplied straight into L */
#define SLIP_LDS_KEYS      (SLIP_LDS_DIROFF + SLIP_PAT_CAP)  /* leading 64 bits of a row multiplied straight into L */
#define SLIP_LDS_TODO      (SLIP_LDS_KEYS + 2 * SLIP_PAT_CAP) /* batch planning: (pivot, digits) requests         */
#define SLIP_LDS_BITMAP    (SLIP_LDS_TODO + 2 * SLIP_WORK_CAP)

enum { SV_ERR = 0, SV_CNT0 = 1 /* 3 rotating work counters */, SV_MAXDIG = 4, SV_GEN = 5, SV_LISTN = 6, SV_TMP = 7,
       SV_LNZ = 8 /* int64 slots from here */, SV_LNL = 10, SV_UNZ = 12, SV_UNL = 14,
       SV_LALLOC = 16 /* limbs of the L slab handed out to this column's direct rows */, SV_LEXACT = 18, SV_LNLX = 20,
       SV_PLAN_D = 24 /* batch planning: digits wanted of 1/rho[jn-1] */, SV_PLAN_R = 25 /* ... of 1/rho[jn] */,
       SV_PLAN_N = 26 /* other (pivot, digits) requests listed in the table area */ };

SLIP_DEV int slip_sgn(int32_t slen) { return (slen > 0) - (slen < 0); }
SLIP_DEV int slip_abs(int32_t v) { return v < 0 ? -v : v; }
SLIP_DEV int slip_limbs(int32_t slen) { return (slip_abs(slen) + 1) >> 1; }

SLIP_DEV uint64_t slip_shfl_up_u64(uint64_t v, int d)
{
    int l = slip_lane(), s = l - d;
    return slip_shfl_u64(v, s < 0 ? l : s);
}

/* maximum over the lanes of a wave (all lanes call) */
SLIP_DEV int slip_wave_max_i32(int v)
{
    const int lane = slip_lane();
    for (int d = 32; d >= 1; d >>= 1) { const int o = (int) slip_shfl_u32((uint32_t) v, lane ^ d); if (o > v) v = o; }
    return v;
}

/* exclusive prefix sums of two values over the workgroup's threads; totals returned */
SLIP_DEV void slip_block_scan2(uint64_t a, uint64_t b, uint64_t *tmp, uint64_t *ea, uint64_t *eb,
                               uint64_t *ta, uint64_t *tb)
{
    const int lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    uint64_t ia = a, ib = b;
    for (int d = 1; d < SLIP_WAVE; d <<= 1) {
        uint64_t x = slip_shfl_up_u64(ia, d), y = slip_shfl_up_u64(ib, d);
        if (lane >= d) { ia += x; ib += y; }
    }
    if (lane == SLIP_WAVE - 1) { tmp[2 * wave] = ia; tmp[2 * wave + 1] = ib; }
    slip_block_sync();
    uint64_t ba = 0, bb = 0, sa = 0, sb = 0;
    for (int w = 0; w < nw; w++) {
        uint64_t x = tmp[2 * w], y = tmp[2 * w + 1];
        if (w < wave) { ba += x; bb += y; }
        sa += x; sb += y;
    }
    slip_block_sync();
    *ea = ba + ia - a; *eb = bb + ib - b; *ta = sa; *tb = sb;
}

/* workgroup minimum of a 64-bit key (all threads get it) */
/* the same for two counts whose totals stay below 2^32: both travel in one 64-bit word (half the shuffles) */
SLIP_DEV void slip_block_scan2_small(uint32_t a, uint32_t b, uint64_t *tmp, uint32_t *ea, uint32_t *eb, uint32_t *ta, uint32_t *tb)
{
    const int lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    const uint64_t v = (uint64_t) a | ((uint64_t) b << 32);
    uint64_t iv = v;
    for (int d = 1; d < SLIP_WAVE; d <<= 1) {
        const uint64_t x = slip_shfl_up_u64(iv, d);
        if (lane >= d) iv += x;
    }
    if (lane == SLIP_WAVE - 1) tmp[wave] = iv;
    slip_block_sync();
    uint64_t before = 0, total = 0;
    for (int w = 0; w < nw; w++) { const uint64_t x = tmp[w]; if (w < wave) before += x; total += x; }
    slip_block_sync();
    const uint64_t ex = before + iv - v;
    *ea = (uint32_t) ex; *eb = (uint32_t)(ex >> 32); *ta = (uint32_t) total; *tb = (uint32_t)(total >> 32);
}

SLIP_DEV uint64_t slip_block_min_u64(uint64_t v, uint64_t *tmp)
{
    const int lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    for (int d = 32; d >= 1; d >>= 1) {
        uint64_t o = slip_shfl_u64(v, lane ^ d);
        if (o < v) v = o;
    }
    if (lane == 0) tmp[wave] = v;
    slip_block_sync();
    uint64_t r = tmp[0];
    for (int w = 1; w < nw; w++) { uint64_t t = tmp[w]; if (t < r) r = t; }
    slip_block_sync();
    return r;
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

SLIP_DEV const dig_t *slip_piv_digits(const SlipParams &P, const SlipPiv &pv) { return (const dig_t *)(P.Llimbs + pv.off); }

SLIP_DEV SlipPiv slip_piv_none(void)
{
    SlipPiv p; p.off = 0; p.len = 0; p.bits = 0; p.lo = 1; p.ctz = 0; p.invlen = 0; p.inv64 = 1; p.pad = 0;
    return p;
}

/* ------------------------------------------------------------------ */
/* in-lane arithmetic for one-limb operands (results up to 127 bits)   */
/* ------------------------------------------------------------------ */
SLIP_DEV int slip_bits128(slip_u128 v)
{
    uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t) v;
    return hi ? 128 - slip_clz64(hi) : (lo ? 64 - slip_clz64(lo) : 0);
}
SLIP_DEV uint64_t slip_inv64(uint64_t d)            /* d odd */
{
    uint64_t x = d;
    x *= 2 - d * x; x *= 2 - d * x; x *= 2 - d * x; x *= 2 - d * x; x *= 2 - d * x;
    return x;
}
/* v / d exactly, v < 2^128, d a one-limb divisor with ctz z and 64-bit inverse of its odd part */
SLIP_DEV slip_u128 slip_divexact128(slip_u128 v, uint64_t d, int z, uint64_t inv64)
{
    const uint64_t dodd = d >> z;
    slip_u128 inv = (slip_u128) inv64;
    inv = inv * ((slip_u128) 2 - (slip_u128) dodd * inv);          /* 128-bit inverse by one Newton step */
    return (v >> z) * inv;
}
/* the 64-bit magnitude of a value of at most 2 digits, read as one aligned limb */
SLIP_DEV uint64_t slip_limb0(const dig_t *p) { return *(const uint64_t *) p; }

/* store a value of at most 4 digits (magnitude mag, sign sgn) as row i */
SLIP_DEV void slip_store_small(const SlipParams &P, int i, slip_u128 mag, int sgn, int h)
{
    const int bits = slip_bits128(mag), len = (bits + 31) >> 5;
    uint64_t *X = (uint64_t *)(P.xd + (int64_t) i * P.xcap);
    X[0] = (uint64_t) mag;
    if (len > 2) X[1] = (uint64_t)(mag >> 64);
    SlipRow r; r.len = sgn < 0 ? -len : len; r.h = h; r.bits = bits; r.pad = 0;
    P.xrow[i] = r;
}

/* ------------------------------------------------------------------ */
/* wave-level pieces                                                    */
/* ------------------------------------------------------------------ */
/* make the cached inverse of pivot p's odd part valid modulo B^want (b0,b1,b2: wave scratch) */
SLIP_DEV int slip_ensure_inv(const SlipParams &P, int p, int want, dig_t *b0, dig_t *b1, dig_t *b2, int publish)
{
    /* another wave may publish a longer inverse at any time: one lane reads, all agree */
    int have = (int) slip_shfl_u32((uint32_t) *(volatile int32_t *) &P.piv[p].invlen, 0);
    if (have >= want) return 0;
    if (!publish) return 2;                   /* a helper never writes the shared cache: the master prepared it */
    if (want > P.invcap || want > P.wcap) return 1;
    int target = 2 * have > want ? 2 * have : want;
    if (target > P.invcap) target = P.invcap;
    if (target > P.wcap) target = P.wcap;
    const SlipPiv pv = P.piv[p];
    const int ld = slip_abs(pv.len), z = pv.ctz;
    int lodd = ld - (z >> 5);
    if (lodd > target) lodd = target;
    wb_copy_shr(b0, slip_piv_digits(P, pv), ld, z, lodd);
    dig_t *inv = P.invd + (int64_t) p * P.invcap;
    wb_inv_extend(inv, have, target, b0, lodd, b1, b2);
    slip_fence_block();
    if (slip_lane() == 0) slip_atomic_max_i32(&P.piv[p].invlen, target);
    slip_wave_sync();
    return 0;
}

/* store a W-digit result (normalising, zero padded to whole limbs) as row i; 1 if it does not fit */
SLIP_DEV int slip_store_x(const SlipParams &P, int i, const dig_t *q, int W, int sign, int h)
{
    const int lane = slip_lane();
    const int len = wb_len(q, W);
    if (len > P.xcap) return 1;
    dig_t *X = P.xd + (int64_t) i * P.xcap;
    const int lw = (len + 1) & ~1;
    for (int c = lane; c < lw; c += SLIP_WAVE) X[c] = c < len ? q[c] : 0u;
    if (lane == 0) {
        SlipRow r; r.len = sign < 0 ? -len : len; r.h = h; r.pad = 0;
        r.bits = len ? 32 * len - slip_clz32(q[len - 1]) : 0;
        P.xrow[i] = r;
    }
    slip_wave_sync();
    return 0;
}

/* ---- register-resident versions (operands of at most 64*D digits; wave_bigint_reg.h) ---- */

/* cached inverse of pivot p's odd part to `want` digits, register Newton; b0: scratch for wide shifts */
template <int D> SLIP_DEV int slip_ensure_inv_reg(const SlipParams &P, int p, int want, dig_t *b0, int publish)
{
    int have = (int) slip_shfl_u32((uint32_t) *(volatile int32_t *) &P.piv[p].invlen, 0);
    if (have >= want) return 0;
    if (!publish) return 2;
    if (want > P.invcap) return 1;
    int target = 2 * have > want ? 2 * have : want;
    if (target > P.invcap) target = P.invcap;
    if (target > 64 * D) target = 64 * D;
    const SlipPiv pv = P.piv[p];
    const int ld = slip_abs(pv.len);
    WR<D> dodd = wr_shr<D>(wr_load<D>(slip_piv_digits(P, pv), ld), pv.ctz, b0);
    if (ld > 64 * D) {               /* the shift must see the digits above the register window */
        wb_copy_shr(b0, slip_piv_digits(P, pv), ld, pv.ctz, 64 * D);
        dodd = wr_load<D>(b0, 64 * D);
    }
    dig_t *inv = P.invd + (int64_t) p * P.invcap;
    WR<D> V = wr_inv_extend<D>(wr_load<D>(inv, have), have, target, dodd);
    wr_store<D>(inv, V, target);
    slip_fence_block();
    if (slip_lane() == 0) slip_atomic_max_i32(&P.piv[p].invlen, target);
    slip_wave_sync();
    return 0;
}

/* store the low W digits of q (normalising, padded to whole limbs) as row i */
template <int D> SLIP_DEV int slip_store_x_reg(const SlipParams &P, int i, const WR<D> &q, int sign, int h)
{
    const int len = wr_len<D>(q);
    if (len > P.xcap) return 1;
    dig_t *X = P.xd + (int64_t) i * P.xcap;
    wr_store<D>(X, q, (len + 1) & ~1);                 /* digits above len are zero in q */
    if (slip_lane() == 0) {
        SlipRow r; r.len = sign < 0 ? -len : len; r.h = h; r.pad = 0; r.bits = 0;
        P.xrow[i] = r;
    }
    const uint32_t top = len ? wr_digit<D>(q, len - 1) : 0u;
    if (slip_lane() == 0) P.xrow[i].bits = len ? 32 * len - slip_clz32(top) : 0;
    slip_wave_sync();
    return 0;
}

/* make the cached inverse of pivot p cover `want` digits, with whichever arithmetic fits the width */
SLIP_DEV int slip_ensure_inv_any(const SlipParams &P, int p, int want, dig_t *b0, dig_t *b1, dig_t *b2)
{
    if (want <= 64)  return slip_ensure_inv_reg<1>(P, p, want, b0, 1);
    if (want <= 128) return slip_ensure_inv_reg<2>(P, p, want, b0, 1);
    if (want <= 192) return slip_ensure_inv_reg<3>(P, p, want, b0, 1);
    if (want <= 256) return slip_ensure_inv_reg<4>(P, p, want, b0, 1);
    return slip_ensure_inv(P, p, want, b0, b1, b2, 1);
}

template <int D> SLIP_DEV int slip_history_wave_reg(const SlipParams &P, int r, int pm, int pd, dig_t *b0, int mode, int publish)
{
    const SlipRow xr = P.xrow[r];
    const int lx = slip_abs(xr.len);
    const SlipPiv m = P.piv[pm];
    const int lm = slip_abs(m.len);
    int sign = slip_sgn(xr.len) * slip_sgn(m.len);
    if (pd >= 0) {
        const SlipPiv d = P.piv[pd];
        const int W = (xr.bits + m.bits - d.bits + 1 + 31) >> 5;
        { const int e = slip_ensure_inv_reg<D>(P, pd, W, b0, publish); if (e) return e; }
        if (mode == 1) return 0;
        WR<D> X = wr_load<D>(P.xd + (int64_t) r * P.xcap, lx), M = wr_load<D>(slip_piv_digits(P, m), lm);
        WR<D> Y = lx <= lm ? wr_mul<D>(X, lx < 64 * D ? lx : 64 * D, M) : wr_mul<D>(M, lm < 64 * D ? lm : 64 * D, X);
        Y = wr_mask<D>(wr_shr<D>(Y, d.ctz, b0), W);
        WR<D> I = wr_load<D>(P.invd + (int64_t) pd * P.invcap, W);
        Y = wr_mask<D>(wr_mul<D>(I, W, Y), W);
        return slip_store_x_reg<D>(P, r, Y, sign * slip_sgn(d.len), xr.h);
    }
    if (mode == 1) return 0;
    WR<D> X = wr_load<D>(P.xd + (int64_t) r * P.xcap, lx), M = wr_load<D>(slip_piv_digits(P, m), lm);
    WR<D> Y = lx <= lm ? wr_mul<D>(X, lx, M) : wr_mul<D>(M, lm, X);
    return slip_store_x_reg<D>(P, r, Y, sign, xr.h);
}

/* fuse_k >= 0: row i is non-pivotal and this is the column's last source, so the history update to level
 * fuse_k-1 (x * rho[fuse_k-1] / rho[jn], slip_REF_triangular_solve.c:248-257) is applied in the same pass;
 * Wf = digits of that result. */
template <int D> SLIP_DEV int slip_ipge_wave_reg(const SlipParams &P, int i, int j, int jn, int64_t m, dig_t *b0,
                                                 int W, int W1, int hist, int hdiv, int mode, int publish, int fuse_k, int Wf)
{
    const SlipRow xi = P.xrow[i], xj = P.xrow[j];
    const SlipEnt le = P.Le[m];
    const SlipPiv R = P.piv[jn];
    const int has_d = jn >= 1;
    const int lx = slip_abs(xi.len);
    SlipPiv Dv = slip_piv_none();
    if (has_d) Dv = P.piv[jn - 1];
    if (hdiv) { const int e = slip_ensure_inv_reg<D>(P, xi.h, W1, b0, publish); if (e) return e; }
    if (has_d) { const int e = slip_ensure_inv_reg<D>(P, jn - 1, W, b0, publish); if (e) return e; }
    if (fuse_k >= 0) { const int e = slip_ensure_inv_reg<D>(P, jn, Wf, b0, publish); if (e) return e; }
    if (mode == 1) return 0;
    const int CAP = 64 * D;
    const int lr = slip_abs(R.len) < W1 ? slip_abs(R.len) : W1;
    WR<D> Rr = wr_load<D>(slip_piv_digits(P, R), lr);
    /* P1 = hist(x_i) * rho_jn  (mod B^W1), sign s1 */
    int s1 = slip_sgn(xi.len) * slip_sgn(R.len);
    WR<D> P1 = wr_zero<D>();
    if (lx) {
        const int lxe = lx < CAP ? lx : CAP;
        WR<D> Y = wr_load<D>(P.xd + (int64_t) i * P.xcap, lxe);
        int ly = lxe;
        if (hist) {
            const int ld = slip_abs(Dv.len) < CAP ? slip_abs(Dv.len) : CAP;
            WR<D> Dd = wr_load<D>(slip_piv_digits(P, Dv), ld);
            Y = ly <= ld ? wr_mul<D>(Y, ly, Dd) : wr_mul<D>(Dd, ld, Y);
            s1 *= slip_sgn(Dv.len);
            ly = W1;
            if (hdiv) {
                const SlipPiv H = P.piv[xi.h];
                Y = wr_mask<D>(wr_shr<D>(Y, H.ctz, b0), W1);
                WR<D> IH = wr_load<D>(P.invd + (int64_t) xi.h * P.invcap, W1);
                Y = wr_mul<D>(IH, W1, Y);
                s1 *= slip_sgn(H.len);
            }
            Y = wr_mask<D>(Y, W1);
        }
        P1 = ly <= lr ? wr_mul<D>(Y, ly, Rr) : wr_mul<D>(Rr, lr, Y);
    }
    /* P2 = L_m * x_j, sign s2 */
    const int ll = slip_abs(le.len) < W1 ? slip_abs(le.len) : W1, lj = slip_abs(xj.len) < W1 ? slip_abs(xj.len) : W1;
    WR<D> Lm = wr_load<D>((const dig_t *)(P.Llimbs + le.off), ll), Xj = wr_load<D>(P.xd + (int64_t) j * P.xcap, lj);
    WR<D> P2 = ll <= lj ? wr_mul<D>(Lm, ll, Xj) : wr_mul<D>(Xj, lj, Lm);
    const int s2 = slip_sgn(le.len) * slip_sgn(xj.len);
    /* T = s1*P1 - s2*P2 (mod B^W1) */
    int sT;
    WR<D> T;
    if (!lx)           { T = wr_addsub<D>(wr_zero<D>(), P2, 1); sT = s2; }
    else if (s1 == s2) { T = wr_addsub<D>(P1, P2, 1); sT = s1; }
    else               { T = wr_addsub<D>(P1, P2, 0); sT = s1; }
    T = wr_mask<D>(T, W1);
    if (has_d) {
        T = wr_mask<D>(wr_shr<D>(T, Dv.ctz, b0), W);
        WR<D> ID = wr_load<D>(P.invd + (int64_t)(jn - 1) * P.invcap, W);
        T = wr_mask<D>(wr_mul<D>(ID, W, T), W);
        sT *= slip_sgn(Dv.len);
    }
    if (wr_digit<D>(T, W - 1) >> 31) { T = wr_mask<D>(wr_addsub<D>(wr_zero<D>(), T, 1), W); sT = -sT; }
    if (fuse_k >= 0) {
        const int lt = wr_len<D>(T);
        if (lt > 0) {
            const SlipPiv Mk = P.piv[fuse_k - 1];
            const int lmk = slip_abs(Mk.len) < CAP ? slip_abs(Mk.len) : CAP;
            WR<D> Mr = wr_load<D>(slip_piv_digits(P, Mk), lmk);
            WR<D> Y = lt <= lmk ? wr_mul<D>(T, lt, Mr) : wr_mul<D>(Mr, lmk, T);
            Y = wr_mask<D>(wr_shr<D>(Y, R.ctz, b0), Wf);
            WR<D> IR = wr_load<D>(P.invd + (int64_t) jn * P.invcap, Wf);
            Y = wr_mask<D>(wr_mul<D>(IR, Wf, Y), Wf);
            return slip_store_x_reg<D>(P, i, Y, sT * slip_sgn(Mk.len) * slip_sgn(R.len), fuse_k - 1);
        }
    }
    return slip_store_x_reg<D>(P, i, T, sT, jn);
}

/* History update of row r (slip_REF_triangular_solve.c:139-149, 248-257), one wavefront:
 *     x[r] <- x[r] * rho[pm] / rho[pd]      (pd < 0: no division); the history tag is kept */
/* mode 0: do it; mode 1: only make sure the shared inverse cache covers it.  publish: may extend the cache. */
SLIP_DEV int slip_history_wave(const SlipParams &P, int r, int pm, int pd, dig_t *b0, dig_t *b1, dig_t *b2,
                               int mode = 0, int publish = 1)
{
    const SlipRow xr = P.xrow[r];
    const int lx = slip_abs(xr.len);
    const dig_t *X = P.xd + (int64_t) r * P.xcap;
    const SlipPiv m = P.piv[pm];
    const int lm = slip_abs(m.len);
    int sign = slip_sgn(xr.len) * slip_sgn(m.len);
    int bq = xr.bits + m.bits;
    {
        /* widths: W digits of result; the shifted product needs W + ceil(ctz/32) */
        int Wn = (bq + 31) >> 5;
        if (pd >= 0) { const SlipPiv d0 = P.piv[pd]; Wn = ((bq - d0.bits + 1 + 31) >> 5) + ((d0.ctz + 31) >> 5); }
        if (Wn > P.wcap) return 1;
        if (Wn <= 64)  return slip_history_wave_reg<1>(P, r, pm, pd, b0, mode, publish);
        if (Wn <= 128) return slip_history_wave_reg<2>(P, r, pm, pd, b0, mode, publish);
        if (Wn <= 192) return slip_history_wave_reg<3>(P, r, pm, pd, b0, mode, publish);
        if (Wn <= 256) return slip_history_wave_reg<4>(P, r, pm, pd, b0, mode, publish);
    }
    if (pd < 0) {
        const int W = (bq + 31) >> 5;
        if (W > P.wcap) return 1;
        if (mode == 1) return 0;
        wb_mul_lo(b0, X, lx, slip_piv_digits(P, m), lm, W);
        return slip_store_x(P, r, b0, W, sign, xr.h);
    }
    const SlipPiv d = P.piv[pd];
    bq -= d.bits - 1;
    const int W = (bq + 31) >> 5, zh = d.ctz, W2 = W + ((zh + 31) >> 5);
    if (W2 > P.wcap) return 1;
    { const int e = slip_ensure_inv(P, pd, W, b0, b1, b2, publish); if (e) return e; }
    if (mode == 1) return 0;
    wb_mul_lo(b0, X, lx, slip_piv_digits(P, m), lm, W2);
    wb_copy_shr(b1, b0, W2, zh, W);
    wb_mul_lo(b2, b1, W, P.invd + (int64_t) pd * P.invcap, W, W);
    return slip_store_x(P, r, b2, W, sign * slip_sgn(d.len), xr.h);
}

/* One IPGE update (slip_REF_triangular_solve.c:156-241) of target row i by source row j
 * (pivot position jn) through the L entry m, one wavefront, everything modulo B^W:
 *     x[i] <- ( hist(x[i]) * rho[jn] - L_m * x[j] ) / rho[jn-1]                          */
struct SlipIpgePlan { int W, W1, W2, hist, hdiv, fk, Wf, Wall, has_d; };

/* bit bounds -> working widths of one IPGE update; scalar code, shared by the wave that performs the update
 * and by the lane that plans the inverse cache for a batch (the two must agree digit for digit) */
SLIP_DEV SlipIpgePlan slip_ipge_plan(const SlipParams &P, int i, int j, int jn, int64_t m, int fuse_k)
{
    const SlipRow xi = P.xrow[i], xj = P.xrow[j];
    const SlipEnt le = P.Le[m];
    const SlipPiv R = P.piv[jn];
    SlipIpgePlan pl;
    const int lx = slip_abs(xi.len), hi = xi.h, br = R.bits;
    pl.has_d = jn >= 1;
    int bd = 0, zd = 0;
    if (pl.has_d) { const SlipPiv D = P.piv[jn - 1]; bd = D.bits; zd = D.ctz; }
    pl.hist = lx != 0 && pl.has_d && hi < jn - 1;
    pl.hdiv = pl.hist && hi > -1;
    int bh = 0, zh = 0;
    if (pl.hdiv) { const SlipPiv H = P.piv[hi]; bh = H.bits; zh = H.ctz; }
    const int bxp = !lx ? 0 : (!pl.hist ? xi.bits : (pl.hdiv ? xi.bits + bd - bh + 1 : xi.bits + bd));
    const int b1b = lx ? bxp + br : 0, b2b = le.bits + xj.bits;
    const int bnum = (b1b > b2b ? b1b : b2b) + 1;
    const int bq = pl.has_d ? bnum - bd + 1 : bnum;
    pl.W = (bq + 1 + 31) >> 5;                       /* + sign bit */
    pl.W1 = pl.W + (pl.has_d ? ((zd + 31) >> 5) : 0);
    pl.W2 = pl.W1 + (pl.hdiv ? ((zh + 31) >> 5) : 0);
    /* last source of the column and a non-pivotal row: fold the history update to level fuse_k-1 in */
    pl.fk = -1; pl.Wf = 0; pl.Wall = pl.W2;
    if (pl.W2 <= P.wcap && fuse_k >= 1 && jn < fuse_k - 1 && P.pinv[i] >= fuse_k) {
        const int bf = bq + P.piv[fuse_k - 1].bits - br + 1;
        int Wf = (bf + 31) >> 5; if (Wf < 1) Wf = 1;
        const int need = Wf + ((R.ctz + 31) >> 5);
        if (need <= 256 && pl.W2 <= 256 && Wf <= P.xcap && Wf <= P.invcap) { pl.fk = fuse_k; pl.Wf = Wf; if (need > pl.Wall) pl.Wall = need; }
    }
    return pl;
}

SLIP_DEV int slip_ipge_wave(const SlipParams &P, int i, int j, int jn, int64_t m, dig_t *b0, dig_t *b1, dig_t *b2,
                            int mode = 0, int publish = 1, int fuse_k = -1)
{
    const SlipRow xi = P.xrow[i], xj = P.xrow[j];
    const SlipEnt le = P.Le[m];
    const SlipPiv R = P.piv[jn];
    const int lx = slip_abs(xi.len), sx = slip_sgn(xi.len);
    const dig_t *X = P.xd + (int64_t) i * P.xcap;
    const int lr = slip_abs(R.len), sr = slip_sgn(R.len);
    const int ll = slip_abs(le.len), sl = slip_sgn(le.len);
    const dig_t *Lm = (const dig_t *)(P.Llimbs + le.off);
    const int lj = slip_abs(xj.len), sj = slip_sgn(xj.len);
    const dig_t *Xj = P.xd + (int64_t) j * P.xcap;
    const int hi = xi.h;
    const SlipIpgePlan pl = slip_ipge_plan(P, i, j, jn, m, fuse_k);
    const int has_d = pl.has_d, hist = pl.hist, hdiv = pl.hdiv, W = pl.W, W1 = pl.W1, W2 = pl.W2;
    const int fk = pl.fk, Wf = pl.Wf, Wall = pl.Wall;
    SlipPiv D = slip_piv_none();
    if (has_d) D = P.piv[jn - 1];
    const int ld = slip_abs(D.len), sd = has_d ? slip_sgn(D.len) : 1, zd = D.ctz;
    int zh = 0, sh = 1;
    if (hdiv) { const SlipPiv H = P.piv[hi]; zh = H.ctz; sh = slip_sgn(H.len); }
    if (W2 > P.wcap) return 1;
    /* operands that fit 256 digits stay in registers */
    if (Wall <= 64)  return slip_ipge_wave_reg<1>(P, i, j, jn, m, b0, W, W1, hist, hdiv, mode, publish, fk, Wf);
    if (Wall <= 128) return slip_ipge_wave_reg<2>(P, i, j, jn, m, b0, W, W1, hist, hdiv, mode, publish, fk, Wf);
    if (Wall <= 192) return slip_ipge_wave_reg<3>(P, i, j, jn, m, b0, W, W1, hist, hdiv, mode, publish, fk, Wf);
    if (Wall <= 256) return slip_ipge_wave_reg<4>(P, i, j, jn, m, b0, W, W1, hist, hdiv, mode, publish, fk, Wf);
    if (hdiv) { const int e = slip_ensure_inv(P, hi, W1, b0, b1, b2, publish); if (e) return e; }
    if (has_d) { const int e = slip_ensure_inv(P, jn - 1, W, b0, b1, b2, publish); if (e) return e; }
    if (mode == 1) return 0;

    /* P1 = hist(x_i) * rho_jn  -> b1, sign s1 */
    int s1 = sx * sr;
    if (lx == 0) {
        /* handled below */
    } else if (!hist) {
        wb_mul_lo(b1, X, lx, slip_piv_digits(P, R), lr, W1);
    } else if (!hdiv) {
        wb_mul_lo(b0, X, lx, slip_piv_digits(P, D), ld, W1);
        wb_mul_lo(b1, b0, W1, slip_piv_digits(P, R), lr, W1);
        s1 *= sd;
    } else {
        wb_mul_lo(b0, X, lx, slip_piv_digits(P, D), ld, W2);
        wb_copy_shr(b1, b0, W2, zh, W1);
        wb_mul_lo(b0, b1, W1, P.invd + (int64_t) hi * P.invcap, W1, W1);
        wb_mul_lo(b1, b0, W1, slip_piv_digits(P, R), lr, W1);
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
        wb_mul_lo(b2, b0, W, P.invd + (int64_t)(jn - 1) * P.invcap, W, W);
        Q = b2; sT *= sd;
    }
    /* two's complement -> sign-magnitude */
    if (Q[W - 1] >> 31) {
        wb_addsub(Q, (const dig_t *) 0, 0, Q, W, W, 1, 0u);
        sT = -sT;
    }
    return slip_store_x(P, i, Q, W, sT, jn);
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

/* in-lane history update  x * rho[pm] / rho[pd]  when everything is one limb; 0 if not applicable */
SLIP_DEV int slip_history_small(const SlipParams &P, const SlipRow &xr, uint64_t xv, int pm, int pd,
                                slip_u128 *out, int *osgn)
{
    const SlipPiv m = P.piv[pm];
    if (slip_abs(xr.len) > 2 || slip_abs(m.len) > 2) return 0;
    slip_u128 y = (slip_u128) xv * m.lo;
    int s = slip_sgn(xr.len) * slip_sgn(m.len);
    if (pd >= 0) {
        const SlipPiv d = P.piv[pd];
        if (slip_abs(d.len) > 2) return 0;
        y = slip_divexact128(y, d.lo, d.ctz, d.inv64);
        s *= slip_sgn(d.len);
    }
    *out = y; *osgn = s;
    return 1;
}

/* left-aligned leading 64 bits of a normalised l-digit magnitude */
SLIP_DEV uint64_t slip_top64(const dig_t *X, int l)
{
    uint64_t top = ((uint64_t) X[l - 1] << 32) | (l >= 2 ? X[l - 2] : 0u);
    const int sh = slip_clz32(X[l - 1]);
    if (sh) top = (top << sh) | (l >= 3 ? (uint64_t)(X[l - 3] >> (32 - sh)) : 0ull);
    return top;
}

/* rows[t] (one-limb values, never updated: h < 0) times the long pivot M: the pivot's digits stay in
 * registers, every wave takes rows in turn (slip_REF_triangular_solve.c:248-257 for untouched rows) */
/* rec3: word 3 of the row's record = (pattern index << 3) | (negative << 2) | digits of the one-limb value;
 * ctab != null (the master's own LDS): also enter the row into the column table and its leading 64 bits into the key
 * list, so that the pivot search does not read back through memory what this CU has just produced */
template <int D> SLIP_DEV void slip_mul_row_finish(const SlipParams &P, const SlipPiv &M, const WR<D> &Y, int r, uint32_t rec3, int64_t off,
                                                   uint32_t slot_off, uint32_t *ctab, uint32_t *ckeys)
{
    const int len = wr_len<D>(Y);
    wr_store<D>((dig_t *)(P.Llimbs + off), Y, (len + 1) & ~1);
    const uint32_t d1 = len ? wr_digit<D>(Y, len - 1) : 0u;
    const int neg = (int)((rec3 >> 2) & 1u) ^ (M.len < 0);
    const int32_t slen = neg ? -len : len;
    const int bits = len ? 32 * len - slip_clz32(d1) : 0;
    if (ctab) {
        const uint32_t d2 = len >= 2 ? wr_digit<D>(Y, len - 2) : 0u, d3 = len >= 3 ? wr_digit<D>(Y, len - 3) : 0u;
        if (slip_lane() == 0) {
            uint64_t top = ((uint64_t) d1 << 32) | d2;
            const int sh = len ? slip_clz32(d1) : 0;
            if (sh) top = (top << sh) | (uint64_t)(d3 >> (32 - sh));
            const int pidx = (int)(rec3 >> 3);
            ctab[0 * SLIP_TAB_CAP + pidx] = (uint32_t) r; ctab[1 * SLIP_TAB_CAP + pidx] = (uint32_t) slen; ctab[2 * SLIP_TAB_CAP + pidx] = (uint32_t) bits;
            ctab[3 * SLIP_TAB_CAP + pidx] = 0x80000000u | slot_off;
            ckeys[2 * pidx] = (uint32_t) top; ckeys[2 * pidx + 1] = (uint32_t)(top >> 32);
        }
    }
    if (slip_lane() == 0) {
        SlipRow nr; nr.len = slen; nr.h = -1;
        nr.pad = 1;                                          /* the value lives in the L slab ... */
        nr.bits = bits;
        P.xrow[r] = nr;
        *(int64_t *)(P.xd + (int64_t) r * P.xcap) = off;     /* ... at this limb offset */
    }
}

template <int D> SLIP_DEV int slip_mul_rows_reg(const SlipParams &P, const SlipPiv &M, const dig_t *Md, const uint32_t *recs,
                                                int first, int stride, int nrows, int64_t slab_base, uint32_t *ctab, uint32_t *ckeys)
{
    const int lane = slip_lane();
    const WR<D> Mr = wr_load<D>(Md, slip_abs(M.len));
    /* record written by the classifying lane: row, low/high digit of the one-limb value, signed length,
     * and the slot of the L slab reserved for the product (these rows ARE L(:,k): no second copy) */
    int t = first;
    /* pairs of one-digit rows (|a| < 2^32, the common case): two carry chains side by side */
    for (; t + stride < nrows; t += 2 * stride) {
        const int u = t + stride;
        const uint32_t w0 = recs[5 * t + 3], w1 = recs[5 * u + 3];
        if ((w0 & 3u) != 1u || (w1 & 3u) != 1u) break;
        WR<D> Y0, Y1;
        wr_mul_digit2<D>(recs[5 * t + 1], recs[5 * u + 1], Mr, Y0, Y1);
        slip_mul_row_finish<D>(P, M, Y0, (int) recs[5 * t], w0, slab_base + (int64_t) recs[5 * t + 4], recs[5 * t + 4], ctab, ckeys);
        slip_mul_row_finish<D>(P, M, Y1, (int) recs[5 * u], w1, slab_base + (int64_t) recs[5 * u + 4], recs[5 * u + 4], ctab, ckeys);
    }
    for (; t < nrows; t += stride) {
        const uint32_t w = recs[5 * t + 3];
        WR<D> Y;
        if ((w & 3u) == 1u) Y = wr_mul_digit<D>(recs[5 * t + 1], Mr);      /* |a| < 2^32 */
        else {
            WR<D> A = wr_zero<D>();
            if (lane == 0) A.d[0] = recs[5 * t + 1];
            if (lane == 1) A.d[0] = recs[5 * t + 2];
            Y = wr_mul<D>(A, (int)(w & 3u), Mr);
        }
        slip_mul_row_finish<D>(P, M, Y, (int) recs[5 * t], w, slab_base + (int64_t) recs[5 * t + 4], recs[5 * t + 4], ctab, ckeys);
    }
    return 0;
}

/* rows [first, first+stride, ...) of a column's one-limb-times-pivot list (5-word records), pivot rho[k-1];
 * Md: staged copy of the pivot's digits, or null to read them from the L slab */
SLIP_DEV int slip_mul_rows_any(const SlipParams &P, int k, const dig_t *Md, const uint32_t *recs, int first, int stride, int nrows,
                               int64_t slab_base, uint32_t *ctab = (uint32_t *) 0, uint32_t *ckeys = (uint32_t *) 0)
{
    const SlipPiv M = P.piv[k - 1];
    if (!Md) Md = slip_piv_digits(P, M);
    const int Dm = (slip_abs(M.len) + 2 + 63) >> 6;
    if (Dm <= 1) return slip_mul_rows_reg<1>(P, M, Md, recs, first, stride, nrows, slab_base, ctab, ckeys);
    if (Dm == 2) return slip_mul_rows_reg<2>(P, M, Md, recs, first, stride, nrows, slab_base, ctab, ckeys);
    if (Dm == 3) return slip_mul_rows_reg<3>(P, M, Md, recs, first, stride, nrows, slab_base, ctab, ckeys);
    return slip_mul_rows_reg<4>(P, M, Md, recs, first, stride, nrows, slab_base, ctab, ckeys);
}

/* ------------------------------------------------------------------ */
/* wave-level work queues: local drain, or fork over the helper workgroups */
/* ------------------------------------------------------------------ */
/* Spins are bounded by ITERATION counts (each iteration sleeps ~1 us): differences of clock64() are
 * not usable for this -- the counter was observed to jump between two reads of one wave. */
#define SLIP_SPIN_LIMIT 30000000ull            /* ~30 s: the master gives up on helpers that do not answer   */
#define SLIP_IDLE_LIMIT 1000000000ull          /* ~15 min: helpers idle for as long as the column loop runs */

/* ---- REF triangular solves (SLIP_LU_solve.c:41-86): the two extra wave-level operations ---- */

/* x[r] <- x[r] / rho[p], exact (slip_back_sub.c:43: divide by the diagonal of U = the pivot) */
SLIP_DEV int slip_divexact_wave(const SlipParams &P, int r, int p, dig_t *b0, dig_t *b1, dig_t *b2, int mode, int publish)
{
    const SlipRow xr = P.xrow[r];
    const int lx = slip_abs(xr.len);
    const SlipPiv d = P.piv[p];
    const int bq = xr.bits - d.bits + 1;
    const int W = bq > 0 ? (bq + 31) >> 5 : 1, zh = d.ctz, W2 = W + ((zh + 31) >> 5);
    if (W2 > P.wcap) return 1;
    { const int e = slip_ensure_inv(P, p, W, b0, b1, b2, publish); if (e) return e; }
    if (mode == 1) return 0;
    wb_copy_shr(b1, P.xd + (int64_t) r * P.xcap, lx, zh, W);
    wb_mul_lo(b2, b1, W, P.invd + (int64_t) p * P.invcap, W, W);
    return slip_store_x(P, r, b2, W, slip_sgn(xr.len) * slip_sgn(d.len), xr.h);
}

/* x[i] <- x[i] - U_m * x[j]   (slip_back_sub.c:44-50) */
SLIP_DEV int slip_submul_wave(const SlipParams &P, int i, int j, int64_t m, dig_t *b0, dig_t *b1, dig_t *b2, int mode)
{
    if (mode == 1) return 0;
    const SlipRow xi = P.xrow[i], xj = P.xrow[j];
    const SlipEnt ue = P.Ue[m];
    const int lx = slip_abs(xi.len), sx = slip_sgn(xi.len);
    const int bt = (xi.bits > ue.bits + xj.bits ? xi.bits : ue.bits + xj.bits) + 1;
    const int W = (bt + 1 + 31) >> 5;                       /* + sign bit */
    if (W > P.wcap) return 1;
    wb_mul_lo(b2, (const dig_t *)(P.Ulimbs + ue.off), slip_abs(ue.len), P.xd + (int64_t) j * P.xcap, slip_abs(xj.len), W);
    const int s2 = slip_sgn(ue.len) * slip_sgn(xj.len);
    if (lx == 0) return slip_store_x(P, i, b2, W, -s2, xi.h);
    int sT = sx;
    wb_addsub(b1, P.xd + (int64_t) i * P.xcap, lx, b2, W, W, sx == s2);       /* |x| -/+ |U x_j| */
    if (sx == s2 && (b1[W - 1] >> 31)) { wb_addsub(b1, (const dig_t *) 0, 0, b1, W, W, 1, 0u); sT = -sT; }
    return slip_store_x(P, i, b1, W, sT, xi.h);
}

/* kind 1: k is the fuse level (-1: none) of the IPGE updates; kind 2: k is the column of the history rows;
 * kind 3: one-limb rows times rho[k-1] straight into the L slab (5-word records, m0 = slab base) */
SLIP_DEV int slip_run_item(const SlipParams &P, int kind, int j, int jn, int k, int64_t m0, const uint32_t *items, int t,
                           dig_t *b0, dig_t *b1, dig_t *b2, int mode, int publish)
{
    if (kind == 1) return slip_ipge_wave(P, (int) items[2 * t + 1], j, jn, m0 + (int64_t) items[2 * t], b0, b1, b2, mode, publish, k);
    if (kind == 5) return slip_submul_wave(P, (int) items[2 * t + 1], j, m0 + (int64_t) items[2 * t], b0, b1, b2, mode);
    if (kind == 3) return mode == 1 ? 0 : slip_mul_rows_any(P, k, (const dig_t *) 0, items, t, 0x40000000, t + 1, m0);   /* m0 = slab base */
    const int r = (int) items[t];
    if (kind == 4) return slip_history_wave(P, r, k - 1, -1, b0, b1, b2, mode, publish);      /* x * rho[k-1] */
    return slip_history_wave(P, r, k - 1, P.xrow[r].h, b0, b1, b2, mode, publish);
}

/* ---- out-of-line entry points (one copy each; they read the parameters from the device copy) ---- */
SLIP_DEVN int slip_run_item_out(const SlipParams *Pg, int kind, int j, int jn, int k, int64_t m0, const uint32_t *items, int t,
                                dig_t *b0, dig_t *b1, dig_t *b2, int mode, int publish)
{
    return slip_run_item(*Pg, kind, j, jn, k, m0, items, t, b0, b1, b2, mode, publish);
}
SLIP_DEVN int slip_history_wave_out(const SlipParams *Pg, int r, int pm, int pd, dig_t *b0, dig_t *b1, dig_t *b2)
{
    return slip_history_wave(*Pg, r, pm, pd, b0, b1, b2);
}
SLIP_DEVN int slip_ensure_inv_out(const SlipParams *Pg, int p, int want, dig_t *b0, dig_t *b1, dig_t *b2)
{
    return slip_ensure_inv_any(*Pg, p, want, b0, b1, b2);
}
SLIP_DEVN int slip_divexact_out(const SlipParams *Pg, int r, int p, dig_t *b0, dig_t *b1, dig_t *b2)
{
    return slip_divexact_wave(*Pg, r, p, b0, b1, b2, 0, 1);
}
/* the exact tolerance test of the diagonal preference: |num| * 2^(-te) >= tol_m * |den|  (slip_get_pivot.c:89-118) */
SLIP_DEVN int slip_tol_compare_out(uint64_t tol_m, int te, const dig_t *num, int ln, const dig_t *den, int ldn,
                                   dig_t *b0, dig_t *b1, dig_t *b2, int wcap)     /* 1 / 0, or -1: scratch too small */
{
    int err_ = 0, *err = &err_;
    const int Wm = ldn + 2;
    if (slip_lane() == 0) { b2[0] = (uint32_t) tol_m; b2[1] = (uint32_t)(tol_m >> 32); }
    slip_wave_sync();
    wb_mul_lo(b0, b2, 2, den, ldn, Wm);
    const int lm_ = wb_len(b0, Wm);
    for (int c = slip_lane(); c < lm_; c += SLIP_WAVE) b2[c] = b0[c];
    slip_wave_sync();
    const int ge = slip_ge_shifted(num, ln, te < 0 ? -te : 0, b2, lm_, te > 0 ? te : 0, b0, b1, wcap, err);
    return err_ ? -1 : ge;
}

/* planning helper, called by every lane of a wave: the lanes with a request (pivot h to W digits) that name the
 * same pivot reduce to one list entry carrying the largest width */
SLIP_DEV void slip_plan_push(volatile int32_t *sv, uint32_t *todo, int has, int h, int W)
{
    const int lane = slip_lane();
    uint64_t pending = slip_ballot(has);
    while (pending) {
        const int l0 = slip_ctz64(pending);
        const int hh = (int) slip_shfl_u32((uint32_t) h, l0);
        const int mine = has && h == hh;
        uint32_t w = mine ? (uint32_t) W : 0u;
        for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = slip_shfl_u32(w, lane ^ d); if (o > w) w = o; }
        if (lane == l0) {
            const int at = slip_atomic_add_i32((int32_t *) &sv[SV_PLAN_N], 1);
            todo[2 * at] = (uint32_t) hh; todo[2 * at + 1] = w;
        }
        pending &= ~slip_ballot(mine);
    }
}

/* Queue processing in two halves so that the caller can overlap its own work with the helpers:
 *   slip_drain_begin: (fork only) prepare the shared inverse cache, publish the batch; returns 1 if forked
 *   slip_drain_end:   this workgroup's share (or the whole queue when not forked), wait for the helpers
 * Both are called by all threads after a workgroup barrier; _end returns after a workgroup barrier with
 * every item done and visible.  Errors land in sv[SV_ERR]. */
SLIP_DEV int slip_drain_begin(const SlipParams &P, uint32_t *lds, int kind, int j, int jn, int k, int64_t m0, int nq,
                              const uint32_t *wl, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    const int fork = P.fork_min > 0 && nq >= P.fork_min;
    if (!fork) return 0;
    /* 1. the shared inverse cache must cover the batch before other CUs read it.  One LANE per item works out
     *    which cached inverses the item divides by and to how many digits (the same width rules the performing
     *    wave applies): 1/rho[jn-1] and 1/rho[jn] are common to a source's updates and reduce to two maxima,
     *    the history divisors that are not long enough yet go on a list; then one wave per inverse extends it. */
    uint32_t *todo = lds + SLIP_LDS_TODO;
    if (tid == 0) { sv[SV_PLAN_D] = 0; sv[SV_PLAN_R] = 0; sv[SV_PLAN_N] = 0; }
    slip_block_sync();
    if (kind == 1 || kind == 2)
        for (int t0 = 0; t0 < nq; t0 += T) {
            const int t = t0 + tid;
            int has = 0, h = 0, Wh = 0, wantD = 0, wantR = 0;
            if (t < nq && kind == 1) {
                const int i = (int) wl[2 * t + 1];
                const SlipIpgePlan pl = slip_ipge_plan(P, i, j, jn, m0 + (int64_t) wl[2 * t], k);
                if (pl.W2 <= P.wcap) {                                     /* else the item itself reports the short buffer */
                    if (pl.has_d) wantD = pl.W;
                    if (pl.fk >= 0) wantR = pl.Wf;
                    if (pl.hdiv) { h = P.xrow[i].h; Wh = pl.W1; has = *(volatile int32_t *) &P.piv[h].invlen < Wh; }
                }
            } else if (t < nq) {
                const SlipRow xr = P.xrow[(int) wl[t]];
                if (xr.h >= 0) {
                    h = xr.h;
                    Wh = (xr.bits + P.piv[k - 1].bits - P.piv[h].bits + 1 + 31) >> 5;
                    has = *(volatile int32_t *) &P.piv[h].invlen < Wh;
                }
            }
            wantD = slip_wave_max_i32(wantD); wantR = slip_wave_max_i32(wantR);    /* one LDS atomic per wave, not per item */
            if (lane == 0 && wantD > 0) slip_atomic_max_i32((int32_t *) &sv[SV_PLAN_D], wantD);
            if (lane == 0 && wantR > 0) slip_atomic_max_i32((int32_t *) &sv[SV_PLAN_R], wantR);
            slip_plan_push(sv, todo, has, h, Wh);
        }
    slip_block_sync();
    {
        const int wd = sv[SV_PLAN_D], wr = sv[SV_PLAN_R], ntodo = sv[SV_PLAN_N];
        int e = 0;
        if (wd > 0 && wave == 0) e = slip_ensure_inv_out(P.self, jn - 1, wd, b0, b1, b2);
        if (!e && wr > 0 && wave == 1 % nw) e = slip_ensure_inv_out(P.self, jn, wr, b0, b1, b2);
        for (int t = (wave + nw - (2 % nw)) % nw; !e && t < ntodo; t += nw)
            e = slip_ensure_inv_out(P.self, (int) todo[2 * t], (int) todo[2 * t + 1], b0, b1, b2);
        if (e && lane == 0) sv[SV_ERR] = e;
    }
    slip_block_sync();
    if (sv[SV_ERR]) return 0;                 /* _end will see the error and do nothing */
    /* 2. publish: items and descriptor to HBM, agent-scope release, bump the generation */
    SlipBatch *B = P.batch;
    const int nwords = (kind == 1 || kind == 5) ? 2 * nq : (kind == 3 ? 5 * nq : nq);
    for (int t = tid; t < nwords; t += T) P.batch_items[t] = wl[t];
    if (tid == 0) {
        B->kind = kind; B->nitems = nq; B->j = j; B->jn = jn; B->k = k; B->m0 = m0; B->stamp = sv[SV_GEN] + 1;
        slip_agent_store_i32(&B->err, 0); slip_agent_store_i32(&B->done, 0);
    }
    slip_vm_drain();
    slip_block_sync();
    if (tid == 0 && P.nhelpers > 0) {
        slip_agent_release();
        const int g = sv[SV_GEN] + 1;
        sv[SV_GEN] = g;
        slip_agent_store_i32(&B->seq, g);
    }
    return 1;
}

SLIP_DEV void slip_drain_end(const SlipParams &P, uint32_t *lds, int forked, int kind, int j, int jn, int k, int64_t m0, int nq,
                             const uint32_t *wl, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int tid = slip_tid(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    if (!forked) {
        if (!sv[SV_ERR])
            for (int t = wave; t < nq; t += nw) {
                const int e = slip_run_item_out(P.self, kind, j, jn, k, m0, wl, t, b0, b1, b2, 0, 1);
                if (e && lane == 0) sv[SV_ERR] = e;
            }
        slip_block_sync();
        return;
    }
    SlipBatch *B = P.batch;
    const int H = P.nhelpers;
    /* 3. with helpers present the master keeps its waves free for its own work (it is the critical path);
     *    without helpers (tests) it does the published batch itself */
    if (H == 0)
        for (int t = wave; t < nq; t += nw) {
            const int e = slip_run_item_out(P.self, kind, j, jn, k, m0, wl, t, b0, b1, b2, 0, 0);
            if (e && lane == 0) sv[SV_ERR] = e == 2 ? 7 : e;
        }
    /* 4. wait for the helpers, then acquire what they wrote */
    slip_vm_drain();
    slip_block_sync();
    if (H > 0) {
        if (tid == 0) {
            unsigned long long spins = 0;
            int ok = 1;
            while (slip_agent_load_i32(&B->done) < H) {
                slip_sleep();
                if (++spins > SLIP_SPIN_LIMIT) { ok = 0; break; }
            }
            slip_agent_acquire();
            const int he = slip_agent_load_i32(&B->err);
            if (!ok) { sv[SV_ERR] = 6; P.dbg[0] = sv[SV_GEN]; P.dbg[1] = slip_agent_load_i32(&B->done); P.dbg[2] = kind; P.dbg[3] = nq; }
            else if (he && sv[SV_ERR] < 6) sv[SV_ERR] = he;
        }
        slip_block_sync();
    }
}

SLIP_DEV void slip_drain(const SlipParams &P, uint32_t *lds, int kind, int j, int jn, int k, int64_t m0, int nq,
                         const uint32_t *wl, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int forked = slip_drain_begin(P, lds, kind, j, jn, k, m0, nq, wl, b0, b1, b2);
    slip_drain_end(P, lds, forked, kind, j, jn, k, m0, nq, wl, b0, b1, b2);
}

/* helper workgroups: wait for batches, do their share, report */
SLIP_DEV void slip_helper_loop(const SlipParams &P, const SlipState *st, uint32_t *lds, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int tid = slip_tid(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    SlipBatch *B = P.batch;
    const int H = P.nhelpers;
    int seen = P.seq0;                       /* the generation the previous launch ended with */
    (void) st;
    for (;;) {
        if (tid == 0) {
            unsigned long long spins = 0;
            int s, gave_up = 0;
            P.dbg[4 * slip_block() + 1] = 1;                 /* polling for a generation newer than `seen` */
            /* only a NEWER generation is a batch (an old value could be a stale copy of the word) */
            while ((int32_t)((s = slip_agent_load_i32(&B->seq)) - seen) <= 0) {
                slip_sleep();
                if (++spins > SLIP_IDLE_LIMIT) { gave_up = 1; break; }
            }
            slip_agent_acquire();
            /* the payload must carry the same generation; if not, this CU still sees old lines: acquire again */
            while (!gave_up && *(volatile int32_t *) &B->stamp != s) {
                slip_agent_add_i32(&B->stale_seen, 1);
                slip_sleep();
                slip_agent_acquire();
                if (++spins > SLIP_IDLE_LIMIT) gave_up = 1;
            }
            sv[SV_GEN] = s; sv[SV_ERR] = 0; sv[SV_TMP] = gave_up;
        }
        slip_block_sync();
        const int s = sv[SV_GEN];
        if (sv[SV_TMP]) { if (tid == 0) P.dbg[4 * slip_block() + 1] = 5; return; }   /* idle limit: nobody talks to us */
        seen = s;
        const int kind = B->kind, nitems = B->nitems, j = B->j, jn = B->jn, k = B->k;
        if (tid == 0) { P.dbg[4 * slip_block()] = s; P.dbg[4 * slip_block() + 1] = 2; P.dbg[4 * slip_block() + 2] = kind; P.dbg[4 * slip_block() + 3] = nitems; }
        if (kind == 0) { if (tid == 0) P.dbg[4 * slip_block() + 1] = 4; return; }     /* the column loop has ended */
        const int64_t m0 = B->m0;
        /* item t -> workgroup t mod H first: a short batch spreads one wave per CU (a wave alone on its SIMD
         * multiplies at full rate) instead of filling the first few workgroups */
        for (int t = (slip_block() - 1) + wave * H; t < nitems; t += nw * H) {
            const int e = slip_run_item_out(P.self, kind, j, jn, k, m0, P.batch_items, t, b0, b1, b2, 0, 0);
            if (e && lane == 0) sv[SV_ERR] = e == 2 ? 8 : 1;      /* 1: a buffer is too small (the host grows it) */
        }
        slip_vm_drain();
        slip_block_sync();
        if (tid == 0) {
            if (sv[SV_ERR]) slip_agent_store_i32(&B->err, sv[SV_ERR]);
            slip_agent_release();
            slip_agent_add_i32(&B->done, 1);
            P.dbg[4 * slip_block() + 1] = 3;
        }
        slip_block_sync();
    }
}

/* ------------------------------------------------------------------ */
/* The ascending sweep over the pivotal positions < k of the pattern (slip_REF_triangular_solve.c:124-241):
 * the bitmap already holds the scattered rows.  Used for a column of the factorisation (k = column)
 * and, with k = n and the right-hand side scattered instead of A(:,col), as the REF forward
 * substitution (slip_forward_sub.c:61-158 is the same recurrence over all positions).
 * Called by all threads; the caller syncs and checks sv[SV_ERR] afterwards. */
template <bool FAST>
SLIP_DEV void slip_sweep(const SlipParams &P, const int k, uint32_t *lds, uint32_t *bm, dig_t *b0, dig_t *b1, dig_t *b2,
                         unsigned long long &c_read, unsigned long long &c_upd, unsigned long long &c_src, unsigned long long &c_str)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint32_t *work = lds + SLIP_LDS_WORK;
        int cur = -1, step = 0;
        int pj = -1, pjn = -1;                       /* the source whose queued (wave) updates are pending */
        int dj = -1, dys = 1, dh = -1;               /* finalised one-limb source value not yet written back */
        slip_u128 dy = 0;
        for (;; step++) {
            slip_block_sync();
            /* write back the previous source's finalised value: every thread has read the old one by now */
            if (dj >= 0) { if (tid == 0) slip_store_small(P, dj, dy, dys, dh); dj = -1; }
            /* queued multi-limb updates of the previous source: one wavefront each */
            const int nq = sv[SV_CNT0 + (step + 2) % 3];
            if (tid == 0) sv[SV_CNT0 + (step + 1) % 3] = 0;
            /* the bitmap is complete for the previous source, so the next one is already known; if there is
             * none, the queued updates of non-pivotal rows fold their history update to level k-1 in */
            const int jn = slip_bitmap_next(bm, cur + 1, k);
            if (nq > 0) {
                slip_block_sync();
                slip_drain(P, lds, 1, pj, pjn, jn < 0 ? k : -1, P.Lp[pjn], nq, work + ((step + 1) & 1) * 2 * SLIP_WORK_CAP, b0, b1, b2);
                if (sv[SV_ERR]) break;
            }
            if (jn < 0) break;
            cur = jn;
            const int j = P.row_perm[jn];
            SlipRow xj = P.xrow[j];
            uint64_t xjv = xj.len != 0 ? slip_limb0(P.xd + (int64_t) j * P.xcap) : 0;
            /* bring x[j] to its final value: history update to level jn-1 (:139-149) */
            if (xj.len != 0 && xj.h < jn - 1) {
                slip_u128 y = 0; int ys = 1;
                if (slip_history_small(P, xj, xjv, jn - 1, xj.h, &y, &ys)) {
                    /* every thread derives the same value in registers; thread 0 stores it one step later */
                    dj = j; dy = y; dys = ys; dh = xj.h;
                    const int yb = slip_bits128(y), yl = (yb + 31) >> 5;
                    xj.len = ys < 0 ? -yl : yl; xj.bits = yb; xjv = (uint64_t) y;
                } else {
                    slip_block_sync();
                    if (wave == 0) {
                        if (slip_history_wave_out(P.self, j, jn - 1, xj.h, b0, b1, b2)) { if (lane == 0) sv[SV_ERR] = 1; }
                    }
                    slip_block_sync();
                    xj = P.xrow[j];
                    xjv = slip_limb0(P.xd + (int64_t) j * P.xcap);
                }
            }
            const int src_nz = xj.len != 0;
            const int64_t m0 = P.Lp[jn], m1 = P.Lp[jn + 1];
            const SlipPiv R = P.piv[jn];
            SlipPiv D = slip_piv_none();
            if (jn >= 1) D = P.piv[jn - 1];
            const int src_small = slip_abs(xj.len) <= 2 && slip_abs(R.len) <= 2 && slip_abs(D.len) <= 2;
            if (src_nz && tid == 0) {
                c_src++;
                c_read += 8ull * slip_limbs(R.len) + (jn >= 1 ? 8ull * slip_limbs(D.len) : 0ull);
            }
            uint32_t *wl = work + (step & 1) * 2 * SLIP_WORK_CAP;
            volatile int32_t *wcnt = &sv[SV_CNT0 + step % 3];
            /* stream L(:,jn): one entry per lane, SLIP_WORK_CAP entries per pass */
            for (int64_t mb = m0; mb < m1; mb += SLIP_WORK_CAP) {
                if (mb > m0) {
                    /* long column: drain the queue of the previous pass before refilling it */
                    slip_block_sync();
                    if (dj >= 0) {                      /* the wave path reads x[j] from memory */
                        if (tid == 0) slip_store_small(P, dj, dy, dys, dh);
                        dj = -1;
                        slip_block_sync();
                    }
                    const int nq2 = *wcnt;
                    slip_drain(P, lds, 1, j, jn, -1, m0, nq2, wl, b0, b1, b2);
                    if (tid == 0) *wcnt = 0;
                    slip_block_sync();
                }
                const int64_t me = mb + SLIP_WORK_CAP < m1 ? mb + SLIP_WORK_CAP : m1;
                for (int64_t mm = mb; mm < me; mm += T) {
                    const int64_t m = mm + tid;
                    int queue = 0, qi = 0;                   /* this lane's update goes to the wave-item queue */
                    if (m < me) do {
                    const int i = P.Li[m];
                    const SlipEnt le = P.Le[m];
                    const int inew = P.pinv[i];
                    /* structural discovery (what the reference's DFS does) */
                    const uint32_t bit = 1u << (inew & 31);
                    const uint32_t old = slip_atomic_or_u32(&bm[inew >> 5], bit);
                    SlipRow xi;
                    if (!(old & bit)) { xi.len = 0; xi.h = -1; xi.bits = 0; xi.pad = 0; P.xrow[i] = xi; }
                    else xi = P.xrow[i];
                    if (!src_nz) break;
                    c_str++; c_read += 4 + 8ull * slip_limbs(le.len);
                    if (inew <= jn || le.len == 0) break;
                    c_upd++;
                    /* ---- one-limb operands: finish the update in this lane ---- */
                    int done = 0;
                    if (src_small && slip_abs(le.len) <= 2 && slip_abs(xi.len) <= 2) {
                        const int lx = xi.len != 0, has_d = jn >= 1;
                        const int hist = lx && has_d && xi.h < jn - 1, hdiv = hist && xi.h > -1;
                        SlipPiv H = slip_piv_none();
                        if (hdiv) H = P.piv[xi.h];
                        const int bxp = !lx ? 0 : (!hist ? xi.bits : (hdiv ? xi.bits + D.bits - H.bits + 1 : xi.bits + D.bits));
                        const int b1b = lx ? bxp + R.bits : 0, b2b = le.bits + xj.bits;
                        const int bnum = (b1b > b2b ? b1b : b2b) + 1;
                        if (bnum <= 126 && (!hdiv || slip_abs(H.len) <= 2)) {
                            slip_u128 y = 0; int s1 = slip_sgn(xi.len) * slip_sgn(R.len);
                            if (lx) {
                                y = (slip_u128) slip_limb0(P.xd + (int64_t) i * P.xcap);
                                if (hist) { y *= D.lo; s1 *= slip_sgn(D.len); }
                                if (hdiv) { y = slip_divexact128(y, H.lo, H.ctz, H.inv64); s1 *= slip_sgn(H.len); }
                                y *= R.lo;
                            }
                            const slip_u128 p2 = (slip_u128) slip_limb0((const dig_t *)(P.Llimbs + le.off)) * xjv;
                            const int s2 = slip_sgn(le.len) * slip_sgn(xj.len);
                            slip_u128 mag; int sT;
                            if (!lx) { mag = p2; sT = -s2; }
                            else if (s1 == s2) { if (y >= p2) { mag = y - p2; sT = s1; } else { mag = p2 - y; sT = -s1; } }
                            else { mag = y + p2; sT = s1; }
                            if (has_d) { mag = slip_divexact128(mag, D.lo, D.ctz, D.inv64); sT *= slip_sgn(D.len); }
                            slip_store_small(P, i, mag, sT, jn);
                            done = 1;
                        }
                    }
                    if (!done) { queue = 1; qi = i; }
                    } while (0);
                    /* queue slots per wave: one LDS atomic per wave instead of one per update on the same counter */
                    const uint64_t qm = slip_ballot(queue);
                    int qbase = 0;
                    if (lane == 0 && qm) qbase = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(qm));
                    qbase = (int) slip_shfl_u32((uint32_t) qbase, 0);
                    if (queue) {
                        const int at = qbase + slip_popc64(qm & ((1ull << lane) - 1ull));
                        wl[2 * at] = (uint32_t)(m - m0); wl[2 * at + 1] = (uint32_t) qi;
                    }
                }
            }
            pj = j; pjn = jn;
        }
}

/* bitmap -> pattern: P.pat[0..npat) = the set pivot positions in ascending order (what slip_sort_xi.c
 * produces); *nU_out = how many of them are below k.  Called by all threads; ends before a barrier. */
SLIP_DEV void slip_pattern(const SlipParams &P, uint32_t *lds, const uint32_t *bm, int k, int *npat_out, int *nU_out)
{
    const int tid = slip_tid(), T = slip_nthreads();
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    const int nwords = P.bm_words;
    const int per = (nwords + T - 1) / T;
    int w0 = tid * per, w1 = w0 + per;
    if (w0 > nwords) w0 = nwords;
    if (w1 > nwords) w1 = nwords;
    uint64_t cntA = 0, cntU = 0;
    for (int w = w0; w < w1; w++) {
        const uint32_t word = bm[w];
        uint32_t below;
        if ((w + 1) * 32 <= k) below = word;
        else if (w * 32 >= k) below = 0;
        else below = word & ((1u << (k - w * 32)) - 1u);
        cntA += (uint64_t) slip_popc32(word); cntU += (uint64_t) slip_popc32(below);
    }
    uint32_t exA, exU, totA, totU_;              /* at most n < 2^31 set bits */
    slip_block_scan2_small((uint32_t) cntA, (uint32_t) cntU, scan_tmp, &exA, &exU, &totA, &totU_);
    {
        /* short patterns never leave the CU (the readers pick the same place by npat) */
        int o = (int) exA;
        int32_t *patl = (int32_t *)(lds + SLIP_LDS_PAT);
        const bool in_lds = totA <= SLIP_PAT_CAP;
        for (int w = w0; w < w1; w++) {
            uint32_t word = bm[w];
            while (word) {
                int b = slip_ctz32(word); word &= word - 1;
                if (in_lds) patl[o++] = w * 32 + b; else P.pat[o++] = w * 32 + b;
            }
        }
    }
    /* the rows behind the positions, gathered once for the phases that follow (one parallel round of loads) */
    if (totA <= SLIP_PAT_CAP) {
        int32_t *patl = (int32_t *)(lds + SLIP_LDS_PAT), *rowl = (int32_t *)(lds + SLIP_LDS_ROWS);
        slip_block_sync();
        for (int t = tid; t < (int) totA; t += T) rowl[t] = P.row_perm[patl[t]];
    }
    *npat_out = (int) totA; *nU_out = (int) totU_;
}

/* ------------------------------------------------------------------ */
/* one column; returns a SLIPDEV_* status (0 = committed)              */
/* ------------------------------------------------------------------ */
/* FAST: bitmap and wave scratch both in LDS (addresses provably LDS, ds_* instructions);
 * otherwise the generic build picks either place at run time (flat addressing). */
template <bool FAST>
SLIP_DEV int slip_do_column(const SlipParams &P, SlipState *st, const int k, uint32_t *lds,
                            unsigned long long *t_read, unsigned long long *t_upd,
                            unsigned long long *t_src, unsigned long long *t_str)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    const int col = P.q[k];
    const int pc_col = P.pinv[col];                     /* position of the "diagonal" row: fixed until this column's swap */
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    volatile int64_t *sv64 = (volatile int64_t *)(lds + SLIP_LDS_VARS);
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint32_t *work = lds + SLIP_LDS_WORK;
    uint32_t *bm = BM_LDS ? lds + SLIP_LDS_BITMAP : P.gbitmap;
    const int wcap = P.wcap;
    dig_t *b0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                        : P.gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    unsigned long long c_read = 0, c_upd = 0, c_src = 0, c_str = 0;
    SLIP_STAMP_INIT();

    /* ---- phase 0: clear the pattern bitmap ---- */
    for (int w = tid; w < P.bm_words; w += T) bm[w] = 0;
    if (tid == 0) { sv[SV_ERR] = 0; sv[SV_CNT0] = 0; sv[SV_CNT0 + 1] = 0; sv[SV_CNT0 + 2] = 0; sv[SV_MAXDIG] = 0; sv64[SV_LALLOC / 2] = 0; sv64[SV_LEXACT / 2] = 0; }
    slip_block_sync();

    /* ---- phase 1: scatter A(:,col) into x (slip_REF_triangular_solve.c:105-119) ---- */
    for (int64_t p = P.Ap[col] + tid; p < P.Ap[col + 1]; p += T) {
        const int row = P.Ai[p];
        const int pos = P.pinv[row];
        slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
        const int32_t al = P.Alen[p];
        const int la = slip_abs(al);
        const dig_t *src = (const dig_t *)(P.Alimbs + P.Aoff[p]);
        dig_t *X = P.xd + (int64_t) row * P.xcap;
        SlipRow r; r.len = al; r.h = -1; r.pad = 0; r.bits = 0;
        if (la > P.xcap) sv[SV_ERR] = 1;
        else {
            const int lw = (la + 1) & ~1;
            for (int c = 0; c < lw; c++) X[c] = c < la ? src[c] : 0u;
            r.bits = la ? 32 * la - slip_clz32(src[la - 1]) : 0;
        }
        P.xrow[row] = r;
        c_read += 4 + 8 * (unsigned long long)((la + 1) >> 1);
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;
    SLIP_STAMP(0);

    /* ---- phase 2: ascending sweep over the pivotal part of the pattern ---- */
    slip_sweep<FAST>(P, k, lds, bm, b0, b1, b2, c_read, c_upd, c_src, c_str);
    slip_block_sync();
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;
    SLIP_STAMP(1);

    /* ---- phase 3: read the bitmap in order = the sorted pattern (slip_sort_xi.c) ---- */
    int npat_, nU_;
    slip_pattern(P, lds, bm, k, &npat_, &nU_);
    const int npat = npat_, nU = nU_, nL = npat - nU;
    const int32_t *patl = (const int32_t *)(lds + SLIP_LDS_PAT);
    auto pat_at = [&](int t) -> int { return npat <= SLIP_PAT_CAP ? patl[t] : P.pat[t]; };
    const int32_t *rowl = (const int32_t *)(lds + SLIP_LDS_ROWS);
    uint32_t *diroff = lds + SLIP_LDS_DIROFF;
    auto row_at = [&](int t) -> int { return npat <= SLIP_PAT_CAP ? rowl[t] : P.row_perm[P.pat[t]]; };
    slip_block_sync();
    SLIP_STAMP(2);

    /* ---- phase 4: history update of the non-pivotal rows to level k-1 (:248-257) ---- */
    /* one-limb rows finished by a lane enter the column table (and the key list of the pivot search) from that lane's
     * registers (diroff[] = 0x7FFFFFFF); for rows multiplied straight into the L slab diroff[] holds the slab offset;
     * all-ones: neither */
    int prefill_ok = k >= 1 && npat <= SLIP_PAT_CAP;
    uint32_t *ctab = lds + SLIP_LDS_TAB, *ckeys = lds + SLIP_LDS_KEYS;
    if (k >= 1) {
        volatile int32_t *wcnt = &sv[SV_CNT0], *wcnt2 = &sv[SV_CNT0 + 1];
        if (tid == 0) { *wcnt = 0; *wcnt2 = 0; }
        uint32_t *wl2 = work + SLIP_WORK_CAP;              /* 5-word records, SLIP_WORK_CAP of them */
        /* rho[k-1] is the multiplier of every row: stage its digits once (LDS when it fits) */
        const SlipPiv M = P.piv[k - 1];
        const int lm = slip_abs(M.len);
        const dig_t *Mg = slip_piv_digits(P, M);
        dig_t *Ms = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + nw * 3 * wcap : (dig_t *) 0;
        const dig_t *Md = Mg;
        if (SCR_LDS && lm <= wcap) { for (int c = tid; c < lm; c += T) Ms[c] = Mg[c]; Md = Ms; }
        slip_block_sync();
        SLIP_STAMP(8);
        uint32_t *wl = work;
        const unsigned long long slot = (unsigned long long)((lm + 3) >> 1);
        for (int t0 = 0; t0 < nL; t0 += SLIP_WORK_CAP) {
            const int te = t0 + SLIP_WORK_CAP < nL ? t0 + SLIP_WORK_CAP : nL;
            const unsigned long long chunk_base = (unsigned long long) sv64[SV_LALLOC / 2];
            for (int tb = t0; tb < te; tb += T) {
                /* every lane classifies its row; list slots are then handed out per WAVE (one LDS atomic per wave and list
                 * instead of one per row on the same counter) */
                const int t = tb + tid;
                int cls = 0, r = 0;                          /* 1: one limb times the long pivot, 2: wave item (division) */
                SlipRow xr; xr.len = 0; xr.h = 0; xr.bits = 0; xr.pad = 0;
                uint64_t xv = 0;
                if (t < te) {
                r = row_at(nU + t);
                xr = P.xrow[r];
                if (npat <= SLIP_PAT_CAP) diroff[nU + t] = 0xFFFFFFFFu;
                }
                if (t < te && !(xr.len == 0 || xr.h >= k - 1)) {
                int done = 0;
                if (slip_abs(xr.len) <= 2) {
                    xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                    slip_u128 y = 0; int ys = 1;
                    if (slip_history_small(P, xr, xv, k - 1, xr.h, &y, &ys)) {
                        slip_store_small(P, r, y, ys, xr.h);
                        if (npat <= SLIP_PAT_CAP) {          /* the lane has the value: table entry and pivot key from registers */
                            const int yb = slip_bits128(y), yl = (yb + 31) >> 5, pidx = nU + t;
                            ctab[0 * SLIP_TAB_CAP + pidx] = (uint32_t) r; ctab[1 * SLIP_TAB_CAP + pidx] = (uint32_t)(ys < 0 ? -yl : yl);
                            ctab[2 * SLIP_TAB_CAP + pidx] = (uint32_t) yb; ctab[3 * SLIP_TAB_CAP + pidx] = 0u;
                            const uint64_t top = yb ? (uint64_t)((y << (128 - yb)) >> 64) : 0ull;
                            ckeys[2 * pidx] = (uint32_t) top; ckeys[2 * pidx + 1] = (uint32_t)(top >> 32);
                            diroff[pidx] = 0x7FFFFFFFu;      /* entered, value in its x row */
                        }
                        done = 1;
                    } else if (xr.h < 0 && lm + 2 <= P.xcap && lm + 2 <= 256) {
                        /* one limb times a long pivot, no division: wave path with the pivot in registers */
                        /* every such row gets a slot of (lm+3)/2 limbs in the L slab (the product has at most lm+2 digits) */
                        cls = 1;
                        done = 1;
                    } else if (xr.h < 0 && lm + 2 <= P.xcap) {
                        /* beyond 256 digits: this lane walks the pivot's digits */
                        dig_t *X = P.xd + (int64_t) r * P.xcap;
                        const uint64_t a0 = xv & 0xFFFFFFFFu, a1 = xv >> 32;
                        uint64_t carry = 0;              /* < 2^64 */
                        for (int c = 0; c < lm; c++) {
                            const uint64_t d = Md[c];
                            const uint64_t lo = a0 * d + (carry & 0xFFFFFFFFu);          /* < 2^64 */
                            X[c] = (uint32_t) lo;
                            carry = a1 * d + (carry >> 32) + (lo >> 32);                 /* < 2^64 */
                        }
                        int len = lm;
                        if (carry) { X[len++] = (uint32_t) carry; if (carry >> 32) X[len++] = (uint32_t)(carry >> 32); }
                        if (len & 1) X[len] = 0;
                        SlipRow nr; nr.len = (slip_sgn(xr.len) * slip_sgn(M.len)) < 0 ? -len : len; nr.h = xr.h; nr.pad = 0;
                        nr.bits = 32 * len - slip_clz32(X[len - 1]);
                        P.xrow[r] = nr;
                        done = 1;
                    }
                }
                if (!done) cls = 2;
                }
                const uint64_t m1 = slip_ballot(cls == 1), m2 = slip_ballot(cls == 2);
                int base1 = 0, base2 = 0;
                if (lane == 0) {
                    if (m1) base1 = slip_atomic_add_i32((int32_t *) wcnt2, slip_popc64(m1));
                    if (m2) base2 = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(m2));
                }
                base1 = (int) slip_shfl_u32((uint32_t) base1, 0); base2 = (int) slip_shfl_u32((uint32_t) base2, 0);
                const uint64_t below = (1ull << lane) - 1ull;
                if (cls == 1) {
                    const int at = base1 + slip_popc64(m1 & below);
                    wl2[5 * at] = (uint32_t) r; wl2[5 * at + 1] = (uint32_t) xv; wl2[5 * at + 2] = (uint32_t)(xv >> 32);
                    wl2[5 * at + 3] = ((uint32_t)(nU + t) << 3) | (xr.len < 0 ? 4u : 0u) | (uint32_t) slip_abs(xr.len);
                    wl2[5 * at + 4] = (uint32_t)(chunk_base + (unsigned long long) at * slot);
                    if (npat <= SLIP_PAT_CAP) diroff[nU + t] = wl2[5 * at + 4];
                } else if (cls == 2) {
                    wl[base2 + slip_popc64(m2 & below)] = (uint32_t) r;
                }
            }
            slip_block_sync();
            SLIP_STAMP(9);
            /* the slots handed out above must exist before anything is written into them */
            const unsigned long long lalloc_now = chunk_base + (unsigned long long) *wcnt2 * slot;
            if (sv64[SV_LNL / 2] + (int64_t) lalloc_now > P.Lcap_nl) return SLIPDEV_GROW_L;
            const int nq = *wcnt;
#ifdef SLIP_PROFILING
            if (tid == 0) st->prof[11] += (unsigned long long) nq;
#endif
            /* the history rows that need a division go to the helpers first; the one-limb-times-pivot rows
             * are multiplied here meanwhile */
            const int forked = slip_drain_begin(P, lds, 2, 0, 0, k, 0, nq, wl, b0, b1, b2);
            const int n2 = *wcnt2;
            const int64_t sb = sv64[SV_LNL / 2];
            /* a long list of one-limb rows goes to the helpers as well (one row per wave over 63 CUs) unless they
             * are busy with this column's division rows; a short one is multiplied here */
            /* (a hand-off costs ~10 us, the price of about 120 rows done here) */
            const int fork3 = !forked && P.fork_min > 0 && n2 >= 10 * P.fork_min;
            if (fork3) prefill_ok = 0;                       /* helpers cannot write this CU's LDS */
            if (n2 > 0 && !fork3) {
                const int e = slip_mul_rows_any(P, k, Md, wl2, wave, nw, n2, sb);     /* inline: once per column */
                if (e && lane == 0) sv[SV_ERR] = 1;
            }
            slip_drain_end(P, lds, forked, 2, 0, 0, k, 0, nq, wl, b0, b1, b2);
            if (fork3 && !sv[SV_ERR]) slip_drain(P, lds, 3, 0, 0, k, sb, n2, wl2, b0, b1, b2);
            SLIP_STAMP(10);
            if (tid == 0) { *wcnt = 0; *wcnt2 = 0; sv64[SV_LALLOC / 2] = (int64_t) lalloc_now; }
            slip_block_sync();
        }
    }
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;
    SLIP_STAMP(3);

    /* ---- phase 5: column-window cap, then the pivot search ---- */
    /* column table in LDS (four arrays of SLIP_TAB_CAP words: row, signed length, bit length, flag/offset per pattern
     * entry -- struct-of-arrays so that consecutive lanes hit consecutive banks): read once, used by
     * the cap test, the pivot search, the offsets and the copy below */
    uint32_t *tab = lds + SLIP_LDS_TAB;
    const bool use_tab = npat <= SLIP_TAB_CAP;
    const int64_t Lnl0 = sv64[SV_LNL / 2];              /* L slab cursor at the start of this column */
    auto ent_row  = [&](int t) -> int { return use_tab ? (int) tab[0 * SLIP_TAB_CAP + t] : row_at(t); };
    auto ent_len  = [&](int t) -> int32_t { return use_tab ? (int32_t) tab[1 * SLIP_TAB_CAP + t] : P.xrow[row_at(t)].len; };
    auto ent_bits = [&](int t) -> int { return use_tab ? (int) tab[2 * SLIP_TAB_CAP + t] : P.xrow[row_at(t)].bits; };
    /* where the digits of a row are: its x row, or (rows multiplied straight into L) the slab */
    auto row_digits = [&](int r) -> const dig_t * {
        const dig_t *X = P.xd + (int64_t) r * P.xcap;
        return P.xrow[r].pad ? (const dig_t *)(P.Llimbs + *(const int64_t *) X) : X;
    };
    auto ent_digits = [&](int t) -> const dig_t * {
        if (use_tab) return (tab[3 * SLIP_TAB_CAP + t] >> 31) ? (const dig_t *)(P.Llimbs + Lnl0 + (int64_t)(tab[3 * SLIP_TAB_CAP + t] & 0x7FFFFFFFu))
                                                  : P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + t] * P.xcap;
        return row_digits(row_at(t));
    };
    {
        int mx = 0;
        for (int t = tid; t < npat; t += T) {
            if (prefill_ok && t >= nU && diroff[t] == 0x7FFFFFFFu) {     /* entered by the lane that produced the value */
                const int l = slip_abs((int32_t) tab[1 * SLIP_TAB_CAP + t]);
                if (l > mx) mx = l;
                continue;
            }
            const int r = row_at(t);
            const SlipRow xr = P.xrow[r];
            const int l = slip_abs(xr.len);
            if (l > mx) mx = l;
            if (use_tab) {
                tab[0 * SLIP_TAB_CAP + t] = (uint32_t) r; tab[1 * SLIP_TAB_CAP + t] = (uint32_t) xr.len; tab[2 * SLIP_TAB_CAP + t] = (uint32_t) xr.bits;
                tab[3 * SLIP_TAB_CAP + t] = xr.pad ? (0x80000000u | diroff[t]) : 0u;      /* use_tab implies the pattern is in LDS */
            }
        }
        mx = slip_wave_max_i32(mx);
        if (mx > 0 && lane == 0) slip_atomic_max_i32((int32_t *) &sv[SV_MAXDIG], mx);
        slip_block_sync();
    }
    const int maxdig = sv[SV_MAXDIG];
    if (P.limb_cap > 0 && ((maxdig + 1) >> 1) > P.limb_cap) return SLIPDEV_WINDOW_END;
    SLIP_STAMP(12);                                       /* column table built */

    /* kind of search: 0 smallest, 1 largest, 2 first nonzero (slip_get_pivot.c:58-155).
     * Lanes order the candidates by (bit length, leading 64 bits); the candidates that tie on
     * that key are compared exactly, ties resolved towards the earlier pattern position. */
    const int scheme = P.pivot_scheme;
    const int kind = (scheme == 2) ? 2 : ((scheme == 4 || scheme == 5) ? 1 : 0);
    int best = -1;
    if (kind != 2 && maxdig < (1 << 18)) {
        /* one pass: (bit length, leading 40 bits) packed into one key; the candidates that share the best key are
         * compared exactly, ties towards the earlier pattern position (slip_get_smallest_pivot.c:79) */
        auto key_of = [&](int t) -> uint64_t {
            const int32_t xl = ent_len(nU + t);
            if (xl == 0) return ~0ull;
            const bool pre = prefill_ok && diroff[nU + t] == 0x7FFFFFFFu;
            const uint64_t top = pre ? ((uint64_t) ckeys[2 * (nU + t)] | ((uint64_t) ckeys[2 * (nU + t) + 1] << 32))
                                     : slip_top64(ent_digits(nU + t), slip_abs(xl));
            const uint64_t v = ((uint64_t) ent_bits(nU + t) << 40) | (top >> 24);
            return kind == 0 ? v : ~v;
        };
        uint64_t mykey = ~0ull, k1 = ~0ull;
        for (int t = tid; t < nL; t += T) { const uint64_t c = key_of(t); if (t == tid) mykey = c; if (c < k1) k1 = c; }
        const uint64_t mk = slip_block_min_u64(k1, scan_tmp);
        if (mk == ~0ull) return SLIPDEV_SINGULAR;
        if (tid == 0) sv[SV_LISTN] = 0;
        slip_block_sync();
        for (int t = tid; t < nL; t += T) {
            const uint64_t c = t == tid ? mykey : key_of(t);
            if (c != mk) continue;
            const int at = slip_atomic_add_i32((int32_t *) &sv[SV_LISTN], 1);
            if (at < 2 * SLIP_WORK_CAP) work[at] = (uint32_t) t;
        }
        slip_block_sync();
        const int nc = sv[SV_LISTN];
        /* every wave performs the same reduction (wave-uniform, reads only) */
        const int listed = nc <= 2 * SLIP_WORK_CAP;
        const int kbits = (int)((kind == 0 ? mk : ~mk) >> 40);   /* at most 40 bits: equal keys are equal values */
        for (int c = 0; c < (listed ? nc : nL); c++) {
            const int t = listed ? (int) work[c] : c;
            if (!listed && key_of(t) != mk) continue;
            if (best < 0) { best = t; continue; }
            int cmp = 0;
            if (kbits > 40)
                cmp = wb_cmp(ent_digits(nU + best), slip_abs(ent_len(nU + best)), ent_digits(nU + t), slip_abs(ent_len(nU + t)));
            if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0) || (cmp == 0 && t < best)) best = t;
        }
        slip_block_sync();
    } else {
        uint64_t k1 = ~0ull;                                     /* (key, t) packed: smaller is better */
        for (int t = tid; t < nL; t += T) {
            if (ent_len(nU + t) == 0) continue;
            const int bits = ent_bits(nU + t);
            const uint64_t key = kind == 2 ? 0 : (kind == 0 ? (uint64_t) bits : (uint64_t)(0x7FFFFFFF - bits));
            const uint64_t c = (key << 32) | (uint32_t) t;
            if (c < k1) k1 = c;
        }
        k1 = slip_block_min_u64(k1, scan_tmp);
        if (k1 == ~0ull) return SLIPDEV_SINGULAR;
        if (kind == 2) best = (int)(k1 & 0xFFFFFFFFu);
        else {
            const uint32_t bkey = (uint32_t)(k1 >> 32);
            const int bbits = kind == 0 ? (int) bkey : 0x7FFFFFFF - (int) bkey;
            /* leading 64 bits of the candidates of the winning bit-length class; kept for the tie pass */
            uint64_t k2 = ~0ull;
            for (int t = tid; t < nL; t += T) {
                const int32_t xl = ent_len(nU + t);
                if (xl == 0 || ent_bits(nU + t) != bbits) continue;
                const uint64_t top = slip_top64(ent_digits(nU + t), slip_abs(xl));
                const uint64_t key = kind == 0 ? top : ~top;
                if (key < k2) k2 = key;
            }
            const uint64_t m2 = slip_block_min_u64(k2, scan_tmp);
            if (tid == 0) sv[SV_LISTN] = 0;
            slip_block_sync();
            for (int t = tid; t < nL; t += T) {
                const int32_t xl = ent_len(nU + t);
                if (xl == 0 || ent_bits(nU + t) != bbits) continue;
                const uint64_t top = slip_top64(ent_digits(nU + t), slip_abs(xl));
                if ((kind == 0 ? top : ~top) != m2) continue;
                const int at = slip_atomic_add_i32((int32_t *) &sv[SV_LISTN], 1);
                if (at < 2 * SLIP_WORK_CAP) work[at] = (uint32_t) t;
            }
            slip_block_sync();
            const int nc = sv[SV_LISTN];
            /* every wave performs the same reduction (wave-uniform, reads only) */
            if (nc <= 2 * SLIP_WORK_CAP) {
                for (int c = 0; c < nc; c++) {
                    const int t = (int) work[c];
                    if (best < 0) { best = t; continue; }
                    int cmp = 0;
                    if (bbits > 64)
                        cmp = wb_cmp(ent_digits(nU + best), slip_abs(ent_len(nU + best)), ent_digits(nU + t), slip_abs(ent_len(nU + t)));
                    if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0) || (cmp == 0 && t < best)) best = t;
                }
            } else {                                              /* too many ties for the list */
                for (int t = 0; t < nL; t++) {
                    const int32_t xl = ent_len(nU + t);
                    if (xl == 0 || ent_bits(nU + t) != bbits) continue;
                    if (best < 0) { best = t; continue; }
                    const int cmp = wb_cmp(ent_digits(nU + best), slip_abs(ent_len(nU + best)), ent_digits(nU + t), slip_abs(xl));
                    if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0)) best = t;
                }
            }
            slip_block_sync();
        }
    }
    SLIP_STAMP(13);                                       /* smallest / largest candidate known */
    int pivrow = ent_row(nU + best);
    /* the diagonal preference (slip_get_pivot.c:68-76, 89-118, 126-146) */
    if (scheme == 1 || scheme == 3 || scheme == 4) {
        const int pc = pc_col;
        const int diag_ok = pc >= k && ((bm[pc >> 5] >> (pc & 31)) & 1u) && P.xrow[col].len != 0;
        if (diag_ok && pivrow != col) {
            int take = 0, err = 0;
            if (scheme == 1) take = 1;
            else if (P.tol_mode == 0) take = 1;
            else {
                const dig_t *num, *den; int ln, ldn;
                const dig_t *xp = row_digits(pivrow), *xc = row_digits(col);
                const int lp_ = slip_abs(P.xrow[pivrow].len), lc_ = slip_abs(P.xrow[col].len);
                if (scheme == 3) { num = xp; ln = lp_; den = xc; ldn = lc_; }   /* |small| / |diag| >= tol */
                else             { num = xc; ln = lc_; den = xp; ldn = lp_; }   /* |diag| / |large| >= tol */
                const int Wm = ldn + 2;
                /* tol_m has exactly 53 bits, so tol_m*|den| has 52 or 53 bits more than |den|: most columns
                 * are decided by the bit lengths alone */
                const int te0 = P.tol_e;
                const int bnum_ = (scheme == 3 ? P.xrow[pivrow].bits : P.xrow[col].bits) + (te0 < 0 ? -te0 : 0);
                const int bden_ = (scheme == 3 ? P.xrow[col].bits : P.xrow[pivrow].bits) + (te0 > 0 ? te0 : 0);
                if (bnum_ < 52 + bden_) take = 0;
                else if (bnum_ > 53 + bden_) take = 1;
                else if (Wm > wcap) err = 1;
                else { const int tk = slip_tol_compare_out(P.tol_m, P.tol_e, num, ln, den, ldn, b0, b1, b2, wcap); if (tk < 0) err = 1; else take = tk; }
            }
            if (err) return SLIPDEV_GROW_X;
            if (take) pivrow = col;
        }
    }
    const int pivpos = pivrow == col ? pc_col : pat_at(nU + best);   /* pre-swap position (the pattern holds positions), >= k */
    SLIP_STAMP(4);

    /* ---- phase 6: append U(:,k) and L(:,k) (SLIP_LU_factorize.c:226-263) ---- */
    /* U(:,k): pattern rows below k in order, then the pivot.  L(:,k): rows at or above k in order. */
    const int nUe = nU + 1, nE = nUe + nL;
    const int64_t Lnz = sv64[SV_LNZ / 2], Lnl = sv64[SV_LNL / 2], Unz = sv64[SV_UNZ / 2], Unl = sv64[SV_UNL / 2];
    uint64_t baseU = 0, baseL = 0;
    /* pattern index of output entry e: U part, then the pivot (L position `pividx`), then the L part */
    int pividx = nU + best;
    if (pivrow != ent_row(nU + best)) {               /* the diagonal was preferred: find it in the L part */
        if (tid == 0) sv[SV_TMP] = -1;
        slip_block_sync();
        for (int t = tid; t < nL; t += T) if (ent_row(nU + t) == pivrow) sv[SV_TMP] = nU + t;
        slip_block_sync();
        pividx = sv[SV_TMP];
    }
    /* the pivot's digits also go to U(:,k): the wave that copies them starts the loads now, the scans below hide them */
    uint32_t pvd[4] = {0u, 0u, 0u, 0u};
    int pv_ok = 0;
    if (use_tab && wave == nU % nw) {
        const int lwp = (slip_abs((int32_t) tab[1 * SLIP_TAB_CAP + pividx]) + 1) & ~1;
        if (lwp <= 4 * SLIP_WAVE) {
            const dig_t *src = (tab[3 * SLIP_TAB_CAP + pividx] >> 31) ? (const dig_t *)(P.Llimbs + Lnl + (int64_t)(tab[3 * SLIP_TAB_CAP + pividx] & 0x7FFFFFFFu))
                                                          : P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + pividx] * P.xcap;
#pragma unroll
            for (int qd = 0; qd < 4; qd++) { const int c = SLIP_WAVE * qd + lane; pvd[qd] = c < lwp ? src[c] : 0u; }
            pv_ok = 1;
        }
    }
    /* rows multiplied straight into the L slab (phase 4) already own the first `lalloc` limbs behind Lnl;
     * the other L rows are copied behind them */
    const uint64_t lalloc = (uint64_t) sv64[SV_LALLOC / 2];
    uint64_t dirL = 0;                                   /* exact limbs of the direct rows (channel packed with U) */
    for (int e0 = 0; e0 < nE; e0 += T) {
        const int e = e0 + tid;
        int r = -1, pt = 0, direct = 0; uint64_t lu = 0, ll = 0; int32_t xl = 0; int xb = 0; int64_t doff = 0;
        if (e < nE) {
            pt = e < nU ? e : (e == nU ? pividx : e - 1);
            r = ent_row(pt); xl = ent_len(pt); xb = ent_bits(pt);
            if (e < nUe) lu = (uint64_t) slip_limbs(xl);
            else {
                if (use_tab) { direct = (int)(tab[3 * SLIP_TAB_CAP + pt] >> 31); doff = Lnl + (int64_t)(tab[3 * SLIP_TAB_CAP + pt] & 0x7FFFFFFFu); }
                else if (P.xrow[r].pad) { direct = 1; doff = *(const int64_t *)(P.xd + (int64_t) r * P.xcap); }
                if (!direct) ll = (uint64_t) slip_limbs(xl);
                else lu = (uint64_t) slip_limbs(xl) << 32;             /* summed in the high half of the U channel */
            }
        }
        uint64_t eu, el, tu, tl;
        slip_block_scan2(lu, ll, scan_tmp, &eu, &el, &tu, &tl);
        if (e < nE) {
            /* capacity is verified before anything is committed; these records are provisional */
            if (e < nUe) {
                const int64_t at = Unz + e;
                if (at < P.Ucap_nz) { P.Ui[at] = r; SlipEnt en; en.off = Unl + (int64_t)(baseU + (eu & 0xFFFFFFFFull)); en.len = xl; en.bits = xb; P.Ue[at] = en; }
                if (use_tab) work[e] = (uint32_t)(baseU + (eu & 0xFFFFFFFFull));     /* copy destination inside the U slab */
            } else {
                const int64_t at = Lnz + (e - nUe);
                const int64_t off = direct ? doff : Lnl + (int64_t)(lalloc + baseL + el);
                if (at < P.Lcap_nz) { P.Li[at] = r; SlipEnt en; en.len = xl; en.bits = xb; en.off = off; P.Le[at] = en; }
                if (use_tab && !direct) tab[3 * SLIP_TAB_CAP + pt] = (uint32_t)(off - Lnl);     /* copy destination, flag bit clear */
            }
        }
        baseU += tu & 0xFFFFFFFFull; dirL += tu >> 32; baseL += tl;
    }
    const uint64_t totU = baseU, totL = lalloc + baseL;         /* limbs of slab consumed by this column */
    const uint64_t totLexact = baseL + dirL;
    if (Unz + nUe > P.Ucap_nz || Unl + (int64_t) totU > P.Ucap_nl) return SLIPDEV_GROW_U;
    if (Lnz + nL > P.Lcap_nz || Lnl + (int64_t) totL > P.Lcap_nl) return SLIPDEV_GROW_L;
#ifdef SLIP_PROFILING
    slip_vm_drain();                 /* diagnostic build: charge the wait for this phase's own stores to this phase */
#endif
    slip_block_sync();
    SLIP_STAMP(5);
    /* limbs: one wave per entry, coalesced; x rows are stored padded to whole limbs and nobody reads
     * the slabs before the barrier below.  Rows that already live in the L slab are not copied. */
#ifdef SLIP_PROFILING
    for (int rep_ = 0; rep_ < 2; rep_++) {       /* diagnostic build: the (idempotent) copy twice, second pass on warm caches */
    if (rep_ == 1) SLIP_STAMP(14);
#endif
    /* the lanes of a wave look at its entries side by side and only the entries that really need a copy (U entries,
     * L rows that are not already in the slab) are then walked one after the other */
    for (int base = wave; base < nE; base += nw * SLIP_WAVE) {
    uint64_t todo_e;
    {
        const int el = base + nw * lane;
        int need = 0;
        if (el < nE) need = (use_tab && el >= nUe) ? !(tab[3 * SLIP_TAB_CAP + (el - 1)] >> 31) : 1;
        todo_e = slip_ballot(need);
    }
    while (todo_e) {
        const int e = base + nw * slip_ctz64(todo_e);
        todo_e &= todo_e - 1;
        const int isU = e < nUe;
        const dig_t *srcx; dig_t *dst; int32_t xl;
        if (use_tab && isU) {
            /* U(:,k): pivotal rows live in x; the pivot (last) may have been multiplied straight into the L slab */
            const int pt = e < nU ? e : pividx;
            xl = (int32_t) tab[1 * SLIP_TAB_CAP + pt];
            srcx = (tab[3 * SLIP_TAB_CAP + pt] >> 31) ? (const dig_t *)(P.Llimbs + Lnl + (int64_t)(tab[3 * SLIP_TAB_CAP + pt] & 0x7FFFFFFFu))
                                           : P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + pt] * P.xcap;
            dst = (dig_t *)(P.Ulimbs + Unl + (int64_t) work[e]);
            if (e == nU && pv_ok) {
                const int lwp = (slip_abs(xl) + 1) & ~1;
#pragma unroll
                for (int qd = 0; qd < 4; qd++) { const int c = SLIP_WAVE * qd + lane; if (c < lwp) dst[c] = pvd[qd]; }
                continue;
            }
        } else if (use_tab) {
            const int pt = e - 1;
            if (tab[3 * SLIP_TAB_CAP + pt] >> 31) continue;                 /* multiplied straight into the slab */
            xl = (int32_t) tab[1 * SLIP_TAB_CAP + pt];
            srcx = P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + pt] * P.xcap;
            dst = (dig_t *)(P.Llimbs + Lnl + (int64_t) tab[3 * SLIP_TAB_CAP + pt]);
        } else {
            const int64_t at = isU ? Unz + e : Lnz + (e - nUe);
            const int r = isU ? P.Ui[at] : P.Li[at];
            const SlipEnt en = isU ? P.Ue[at] : P.Le[at];
            xl = en.len;
            srcx = row_digits(r);
            dst = isU ? (dig_t *)(P.Ulimbs + en.off) : (dig_t *)(P.Llimbs + en.off);
            if (dst == srcx) continue;
        }
        const int lw = (slip_abs(xl) + 1) & ~1;
        for (int c = lane; c < lw; c += SLIP_WAVE) dst[c] = srcx[c];
    }
    }
#ifdef SLIP_PROFILING
    }
    SLIP_STAMP(15);                                       /* second pass of wave 0's share */
#endif
    slip_block_sync();
    SLIP_STAMP(6);
    /* pivot bookkeeping (slip_get_pivot.c:164-182); wave 0 */
    if (wave == 0) {
        /* position of the pivot inside L(:,k): `best` unless the diagonal was preferred */
        const int found = pividx - nU;
        const SlipEnt pe = P.Le[Lnz + found];
        const int lp_ = slip_abs(pe.len);
        const dig_t *pv = (const dig_t *)(P.Llimbs + pe.off);
        const int z = wb_ctz(pv, lp_);
        if (lane == 0) {
            SlipPiv pr; pr.off = pe.off; pr.len = pe.len; pr.bits = pe.bits; pr.ctz = z; pr.invlen = 0;
            pr.lo = *(const uint64_t *) pv; pr.inv64 = 0; pr.pad = 0;
            if (lp_ <= 2) pr.inv64 = slip_inv64(pr.lo >> z);
            P.piv[k] = pr;
            const int intermed = pivpos, intermed2 = P.row_perm[k];
            P.row_perm[k] = pivrow; P.row_perm[intermed] = intermed2;
            P.pinv[pivrow] = k; P.pinv[intermed2] = intermed;
            sv64[SV_UNZ / 2] = Unz + nUe; sv64[SV_UNL / 2] = Unl + (int64_t) totU;
            sv64[SV_LNZ / 2] = Lnz + nL;  sv64[SV_LNL / 2] = Lnl + (int64_t) totL;
            sv64[SV_LNLX / 2] = sv64[SV_LNLX / 2] + (int64_t) totLexact;
            P.Up[k + 1] = Unz + nUe; P.Lp[k + 1] = Lnz + nL;
            st->c_write += 4ull * (unsigned long long) nE + 8ull * (totU + totLexact) + 8ull * slip_limbs(pe.len);
            if ((unsigned long long) maxdig > st->c_maxdig) st->c_maxdig = (unsigned long long) maxdig;
        }
    }
    *t_read += c_read; *t_upd += c_upd; *t_src += c_src; *t_str += c_str;
    slip_block_sync();
    SLIP_STAMP(7);
    return SLIPDEV_OK;
}

/* the kernel body: columns [k_next, k_stop) on ONE workgroup */
template <bool FAST>
SLIP_DEV void slip_factor_columns(const SlipParams &P, SlipState *st, uint32_t *lds)
{
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    if (slip_block() != 0) {
        const int wcap = P.wcap, wave = slip_wave();
        dig_t *hb0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                             : P.gscratch + ((int64_t) slip_block() * slip_nwaves() + wave) * 3 * wcap;
        slip_helper_loop(P, st, lds, hb0, hb0 + wcap, hb0 + 2 * wcap);
        return;
    }
    unsigned long long t_read = 0, t_upd = 0, t_src = 0, t_str = 0;
    volatile int64_t *sv64 = (volatile int64_t *)(lds + SLIP_LDS_VARS);
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    if (slip_tid() == 0) sv[SV_GEN] = P.seq0;
    int k = st->k_next;
    if (slip_tid() == 0) {
        sv64[SV_LNZ / 2] = st->Lnz; sv64[SV_LNL / 2] = st->Lnl; sv64[SV_UNZ / 2] = st->Unz; sv64[SV_UNL / 2] = st->Unl;
        sv64[SV_LNLX / 2] = st->Lnl_exact;
    }
    int status = SLIPDEV_OK;
    slip_block_sync();
    for (; k < P.k_stop; k++) {
        status = slip_do_column<FAST>(P, st, k, lds, &t_read, &t_upd, &t_src, &t_str);
        if (status != SLIPDEV_OK) break;
    }
    slip_block_sync();
    /* release the helper workgroups */
    if (P.nhelpers > 0 && slip_tid() == 0) {
        P.batch->kind = 0; P.batch->stamp = sv[SV_GEN] + 1;
        slip_agent_release();
        sv[SV_GEN] = sv[SV_GEN] + 1;
        slip_agent_store_i32(&P.batch->seq, sv[SV_GEN]);
    }
    /* per-thread counters -> totals */
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint64_t e0, e1, a, b, c, d;
    slip_block_scan2(t_read, t_upd, scan_tmp, &e0, &e1, &a, &b);
    slip_block_scan2(t_src, t_str, scan_tmp, &e0, &e1, &c, &d);
    if (slip_tid() == 0) {
        st->c_read += a; st->c_upd += b; st->c_src += c; st->c_streamed += d;
        st->Lnz = sv64[SV_LNZ / 2]; st->Lnl = sv64[SV_LNL / 2]; st->Unz = sv64[SV_UNZ / 2]; st->Unl = sv64[SV_UNL / 2];
        st->Lnl_exact = sv64[SV_LNLX / 2];
        st->k_next = k; st->status = status; st->status_k = k; st->seq = sv[SV_GEN];
    }
}

/* ------------------------------------------------------------------ */
/* REF forward / back substitution for one right-hand side             */
/* ------------------------------------------------------------------ */
/* previous set bit of the bitmap strictly below `from`, or -1 (wave-cooperative) */
SLIP_DEV int slip_bitmap_prev(const uint32_t *bm, int from)
{
    const int lane = slip_lane();
    if (from <= 0) return -1;
    int whi = (from - 1) >> 5;                    /* highest word that can hold a candidate */
    int first = 1;
    while (whi >= 0) {
        const int idx = whi - lane;
        uint32_t word = idx >= 0 ? bm[idx] : 0u;
        if (first && lane == 0 && (from & 31)) word &= (1u << (from & 31)) - 1u;
        const uint64_t nz = slip_ballot(word != 0);
        if (nz) {
            const int t = slip_ctz64(nz);          /* lowest lane = highest word */
            const uint32_t wv = slip_shfl_u32(word, t);
            return (whi - t) * 32 + 31 - slip_clz32(wv);
        }
        whi -= SLIP_WAVE; first = 0;
    }
    return -1;
}

template <bool FAST>
SLIP_DEV int slip_solve_rhs(const SlipParams &P, SlipState *st, const SlipSolveArgs &A, const int c, uint32_t *lds)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    const int n = P.n;
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint32_t *work = lds + SLIP_LDS_WORK;
    uint32_t *bm = BM_LDS ? lds + SLIP_LDS_BITMAP : P.gbitmap;
    const int wcap = P.wcap;
    dig_t *b0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                        : P.gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    unsigned long long c_read = 0, c_upd = 0, c_src = 0, c_str = 0;

    /* b2[pinv[i]] = b[i]  (SLIP_LU_solve.c:68-75): rows keep their ids, the bitmap is indexed by position */
    for (int w = tid; w < P.bm_words; w += T) bm[w] = 0;
    if (tid == 0) { sv[SV_ERR] = 0; sv[SV_CNT0] = 0; sv[SV_CNT0 + 1] = 0; sv[SV_CNT0 + 2] = 0; }
    slip_block_sync();
    for (int i = tid; i < n; i += T) {
        const int32_t bl = A.blen[(int64_t) c * n + i];
        if (bl == 0) continue;
        const int pos = P.pinv[i], lb = slip_abs(bl);
        slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
        const dig_t *src = (const dig_t *)(A.blimbs + A.boff[(int64_t) c * n + i]);
        dig_t *X = P.xd + (int64_t) i * P.xcap;
        SlipRow r; r.len = bl; r.h = -1; r.pad = 0; r.bits = 0;
        if (lb > P.xcap) sv[SV_ERR] = 1;
        else {
            const int lw = (lb + 1) & ~1;
            for (int d = 0; d < lw; d++) X[d] = d < lb ? src[d] : 0u;
            r.bits = 32 * lb - slip_clz32(src[lb - 1]);
        }
        P.xrow[i] = r;
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;

    /* forward substitution = the sweep over ALL pivot positions (slip_forward_sub.c:61-158) */
    slip_sweep<FAST>(P, n, lds, bm, b0, b1, b2, c_read, c_upd, c_src, c_str);
    slip_block_sync();
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;

    /* x <- x * det (slip_array_mul.c:19), det = rho[n-1] */
    int npat, nUdummy;
    slip_pattern(P, lds, bm, n, &npat, &nUdummy);
    const int32_t *patl = (const int32_t *)(lds + SLIP_LDS_PAT);
    auto pat_at = [&](int t) -> int { return npat <= SLIP_PAT_CAP ? patl[t] : P.pat[t]; };
    slip_block_sync();
    {
        volatile int32_t *wcnt = &sv[SV_CNT0];
        if (tid == 0) *wcnt = 0;
        slip_block_sync();
        for (int t0 = 0; t0 < npat; t0 += SLIP_WORK_CAP) {
            const int te = t0 + SLIP_WORK_CAP < npat ? t0 + SLIP_WORK_CAP : npat;
            for (int tb = t0; tb < te; tb += T) {
                const int t = tb + tid;
                int queue = 0, r = 0;
                if (t < te) do {
                    r = P.row_perm[pat_at(t)];
                    const SlipRow xr = P.xrow[r];
                    if (xr.len == 0) break;
                    slip_u128 y = 0; int ys = 1; int done = 0;
                    if (slip_abs(xr.len) <= 2) {
                        const uint64_t xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                        if (slip_history_small(P, xr, xv, n - 1, -1, &y, &ys)) { slip_store_small(P, r, y, ys, xr.h); done = 1; }
                    }
                    if (!done) queue = 1;
                } while (0);
                /* queue slots per wave (one LDS atomic per wave, not per row on the same counter) */
                const uint64_t qm = slip_ballot(queue);
                int qbase = 0;
                if (lane == 0 && qm) qbase = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(qm));
                qbase = (int) slip_shfl_u32((uint32_t) qbase, 0);
                if (queue) work[qbase + slip_popc64(qm & ((1ull << lane) - 1ull))] = (uint32_t) r;
            }
            slip_block_sync();
            slip_drain(P, lds, 4, 0, 0, n, 0, *wcnt, work, b0, b1, b2);
            if (tid == 0) *wcnt = 0;
            slip_block_sync();
        }
    }
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;

    /* back substitution (slip_back_sub.c:36-52): positions descending; x_j /= U_jj (= rho_j, the last entry
     * of U(:,j)), then x_i -= U_ij x_j for the rows above */
    {
        volatile int32_t *wcnt = &sv[SV_CNT0];
        int cur = n;
        for (;;) {
            slip_block_sync();
            const int jp = slip_bitmap_prev(bm, cur);
            if (jp < 0) break;
            cur = jp;
            const int j = P.row_perm[jp];
            SlipRow xj = P.xrow[j];
            if (xj.len == 0) continue;
            const SlipPiv Dj = P.piv[jp];
            if (slip_abs(xj.len) <= 2 && slip_abs(Dj.len) <= 2) {
                const slip_u128 y = slip_divexact128((slip_u128) slip_limb0(P.xd + (int64_t) j * P.xcap), Dj.lo, Dj.ctz, Dj.inv64);
                slip_block_sync();                                 /* every thread has read the old value */
                if (tid == 0) slip_store_small(P, j, y, slip_sgn(xj.len) * slip_sgn(Dj.len), xj.h);
            } else {
                slip_block_sync();
                if (wave == 0) { const int e = slip_divexact_out(P.self, j, jp, b0, b1, b2); if (e && lane == 0) sv[SV_ERR] = e; }
            }
            slip_block_sync();
            if (sv[SV_ERR]) break;
            xj = P.xrow[j];
            const uint64_t xjv = slip_limb0(P.xd + (int64_t) j * P.xcap);
            const int64_t m0 = P.Up[jp], m1 = P.Up[jp + 1] - 1;   /* the pivot is the last entry */
            for (int64_t mb = m0; mb < m1; mb += SLIP_WORK_CAP) {
                const int64_t me = mb + SLIP_WORK_CAP < m1 ? mb + SLIP_WORK_CAP : m1;
                for (int64_t mm = mb; mm < me; mm += T) {
                    const int64_t m = mm + tid;
                    int queue = 0, qi = 0;
                    if (m < me) do {
                    const int i = P.Ui[m];
                    const SlipEnt ue = P.Ue[m];
                    const int pos = P.pinv[i];
                    const uint32_t bit = 1u << (pos & 31);
                    const uint32_t old = slip_atomic_or_u32(&bm[pos >> 5], bit);
                    SlipRow xi;
                    if (!(old & bit)) { xi.len = 0; xi.h = -1; xi.bits = 0; xi.pad = 0; P.xrow[i] = xi; }
                    else xi = P.xrow[i];
                    if (ue.len == 0) break;
                    int done = 0;
                    if (slip_abs(ue.len) <= 2 && slip_abs(xj.len) <= 2 && slip_abs(xi.len) <= 2) {
                        const int bt = (xi.bits > ue.bits + xj.bits ? xi.bits : ue.bits + xj.bits) + 1;
                        if (bt <= 126) {
                            const slip_u128 p2 = (slip_u128) slip_limb0((const dig_t *)(P.Ulimbs + ue.off)) * xjv;
                            const int s2 = slip_sgn(ue.len) * slip_sgn(xj.len), sx = slip_sgn(xi.len);
                            slip_u128 mag; int sT;
                            if (xi.len == 0) { mag = p2; sT = -s2; }
                            else {
                                const slip_u128 xv = (slip_u128) slip_limb0(P.xd + (int64_t) i * P.xcap);
                                if (sx == s2) { if (xv >= p2) { mag = xv - p2; sT = sx; } else { mag = p2 - xv; sT = -sx; } }
                                else { mag = xv + p2; sT = sx; }
                            }
                            slip_store_small(P, i, mag, sT, xi.h);
                            done = 1;
                        }
                    }
                    if (!done) { queue = 1; qi = i; }
                    } while (0);
                    const uint64_t qm = slip_ballot(queue);          /* queue slots per wave */
                    int qbase = 0;
                    if (lane == 0 && qm) qbase = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(qm));
                    qbase = (int) slip_shfl_u32((uint32_t) qbase, 0);
                    if (queue) {
                        const int at = qbase + slip_popc64(qm & ((1ull << lane) - 1ull));
                        work[2 * at] = (uint32_t)(m - m0); work[2 * at + 1] = (uint32_t) qi;
                    }
                }
                slip_block_sync();
                slip_drain(P, lds, 5, j, jp, n, m0, *wcnt, work, b0, b1, b2);
                if (tid == 0) *wcnt = 0;
                slip_block_sync();
                if (sv[SV_ERR]) break;
            }
            if (sv[SV_ERR]) break;
        }
    }
    slip_block_sync();
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;

    /* output: numerators in pivot-position order (the order SLIP_LU_solve returns before SLIP_permute_x) */
    {
        const int64_t obase = st->out_used;
        uint64_t run = 0;
        for (int p0 = 0; p0 < n; p0 += T) {
            const int pos = p0 + tid;
            int32_t xl = 0; int r = -1;
            if (pos < n && ((bm[pos >> 5] >> (pos & 31)) & 1u)) { r = P.row_perm[pos]; xl = P.xrow[r].len; }
            uint64_t e0, e1, t0_, t1_;
            slip_block_scan2((uint64_t) slip_limbs(xl), 0, scan_tmp, &e0, &e1, &t0_, &t1_);
            if (pos < n) {
                const int64_t off = obase + (int64_t)(run + e0);
                A.olen[(int64_t) c * n + pos] = xl;
                A.ooff[(int64_t) c * n + pos] = off;
                if (xl != 0 && off + slip_limbs(xl) <= A.ocap) {
                    const dig_t *src = P.xd + (int64_t) r * P.xcap;
                    dig_t *dst = (dig_t *)(A.olimbs + off);
                    const int lw = (slip_abs(xl) + 1) & ~1;
                    for (int d = 0; d < lw; d++) dst[d] = src[d];
                }
            }
            run += t0_;
        }
        if (obase + (int64_t) run > A.ocap) return SLIPDEV_GROW_U;       /* output slab too small: the host grows it */
        slip_block_sync();
        if (tid == 0) st->out_used = obase + (int64_t) run;
    }
    slip_block_sync();
    return SLIPDEV_OK;
}

/* kernel body of the solves: block 0 walks the right-hand sides, the others are helpers */
template <bool FAST>
SLIP_DEV void slip_solve_all(const SlipParams &P, SlipState *st, const SlipSolveArgs &A, uint32_t *lds)
{
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    if (slip_block() != 0) {
        const int wcap = P.wcap, wave = slip_wave();
        dig_t *hb0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                             : P.gscratch + ((int64_t) slip_block() * slip_nwaves() + wave) * 3 * wcap;
        slip_helper_loop(P, st, lds, hb0, hb0 + wcap, hb0 + 2 * wcap);
        return;
    }
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    if (slip_tid() == 0) sv[SV_GEN] = P.seq0;
    slip_block_sync();
    int c = st->solve_next, status = SLIPDEV_OK;
    for (; c < A.nrhs; c++) {
        status = slip_solve_rhs<FAST>(P, st, A, c, lds);
        if (status != SLIPDEV_OK) break;
    }
    slip_block_sync();
    if (slip_tid() == 0) {
        if (P.nhelpers > 0) {
            P.batch->kind = 0; P.batch->stamp = sv[SV_GEN] + 1;
            slip_agent_release();
            sv[SV_GEN] = sv[SV_GEN] + 1;
            slip_agent_store_i32(&P.batch->seq, sv[SV_GEN]);
        }
        st->solve_next = c; st->status = status; st->status_k = c; st->seq = sv[SV_GEN];
    }
}

#endif /* SLIP_REF_LU_KERNEL_H */
