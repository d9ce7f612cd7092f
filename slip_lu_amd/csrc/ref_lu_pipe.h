/* ref_lu_pipe.h -- the left-looking REF sparse LU column loop as a PIPELINE of column workers (gfx950).
 *
 * Replaces, on the device, the reference's hot path
 *   SLIP_LU/Source/SLIP_LU_factorize.c:190-264      (column loop, L/U split)
 *   SLIP_LU/Source/slip_REF_triangular_solve.c:65-265 (reach + history/IPGE sweep)
 *   SLIP_LU/Source/slip_reach.c, slip_dfs.c, slip_sort_xi.c (pattern, order)
 *   SLIP_LU/Source/slip_get_pivot.c:30-183 and the smallest/largest/nonzero searches
 * with an MI355X-first formulation rather than a translation.
 *
 * The reference runs the columns one after the other.  Here every workgroup of the grid is a COLUMN
 * WORKER: it draws the next column k from a ticket counter, scatters A(:,q[k]) into its own private dense
 * vector x and runs the ascending sweep over the pivotal part of the pattern AHEAD of the commit frontier
 * F: a source at pivot position jn can be applied as soon as column jn is committed, because everything
 * below the frontier is final (pivot rows, pivots, L columns) and the sweep's order is ascending anyway
 * (slip_REF_triangular_solve.c:124-131).  Rows found non-pivotal are re-examined when the frontier moves
 * (the row that became pivotal at position c is row_perm[c]).  Only what really depends on the previous
 * column is serial: when F == k the worker applies source k-1 if it is in the pattern, brings the rows to
 * level k-1, chooses the pivot, publishes it (pinv/row_perm swap, rho_k, column pointers: "stage 1",
 * F = k+1) and then writes the bulk of L(:,k), U(:,k) ("stage 2", Lready[k]) while the next worker is
 * already committing.  Many columns are in flight on different CUs; the chain is one short hop per column.
 *
 * Inside a worker the formulation of round 1 is kept:
 *  - No DFS and no sort: the pattern is a bitmap over pivot positions (LDS) read in ascending order.
 *  - A source's L column is streamed coalesced, one entry per LANE; one-limb updates finish in the lane
 *    (128-bit arithmetic), the others are queued in LDS and done one per WAVEFRONT (wave_bigint_reg.h).
 *  - Exact divisions are 2-adic with cached Newton inverses of the pivots (shared, extended on demand;
 *    concurrent extensions write identical digits).
 *  - Pivot search, permutation swap and the L/U append stay on the device.
 *
 * Memory visibility between workers (cdna_hip_programming.md, Guideline 16): everything another worker
 * may read during the launch (pinv, row_perm, pivot records, L structure and limbs, the inverse cache,
 * the frontier words) is read ONLY with sc1 loads (slip_ld_*: they bypass the reading CU's L1), so no acquire
 * fence is needed on the reading side.  On the writing side the small stage-1 data (the pivot's digits and
 * record, the permutation swap, the column pointers) and the inverse cache use write-through (sc1) stores,
 * drained (vmcnt(0)) by every storing wave before ONE lane raises the flag; the bulk of a column (L entries
 * and limbs) is written with plain coalesced stores and published by ONE agent-scope release fence (L2
 * write-back) before Lready[k].  A worker's x vector is private (plain accesses).  Nothing depends on
 * dispatch order or residency: a workgroup that is not resident holds no ticket.
 *
 * Values are sign-magnitude: a signed digit count (32-bit digits) plus the magnitude; all stores are
 * padded to whole 64-bit limbs.
 */
#ifndef SLIP_REF_LU_PIPE_H
#define SLIP_REF_LU_PIPE_H

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
    SLIPDEV_INTERNAL = 6     /* a wait was not answered / inconsistent state         */
};

/* state of one row of a worker's private dense scatter vector x */
typedef struct { int32_t len, h, bits, tag; } SlipRow;            /* signed digits, history, bit length, ticket of the column it belongs to */
/* one stored entry of L or U: where its limbs are, how long, how many bits */
typedef struct { int64_t off; int32_t len, bits; } SlipEnt;       /* off in 64-bit limbs */
/* one pivot rho[k] (= the pivot entry of L(:,k)) */
typedef struct {
    int64_t off; int32_t len, bits;          /* limbs in the L slab */
    uint64_t lo; int32_t ctz, invlen;        /* low limb; trailing zero bits; cached inverse digits */
    uint64_t inv64, pad;                     /* inverse of the odd part modulo 2^64 (one-limb pivots); pad: how many divisions by this
                                              * pivot found its cached inverse too short (slip_div_piv_reg decides on it) */
} SlipPiv;

/* mutable across launches; the words other workers poll sit in 128-byte lines of their own */
#ifndef SLIP_FARM_HINTS
#define SLIP_FARM_HINTS 16           /* hint slots (a power of two, at most 32: one line); 8 / 16 / 32 measured: model6 516 / 501 / 494 ms, C4 3.10 / 3.05 / 3.03, rl5934 119.3 / 120.1 / 121.7 */
#endif
typedef struct SlipState {
    int32_t F, Fpiv; int32_t padF[30];              /* ONE aligned 64-bit word: the commit frontier (columns < F have published
                                                       their pivot, stage 1) and, next to it, the pivot row of column F-1 */
    int64_t stop; int64_t padS[15];                 /* (column << 8) | SLIPDEV_* of the first column that cannot commit; min wins */
    int32_t F2; int32_t padF2[31];                  /* ready frontier: columns < F2 have published their L entries (stage 2) */
    int32_t ticket; int32_t padT[31];               /* next column ticket (monotonic across launches)                    */
    int32_t exited, padE[31];                       /* workers that have left the launch (the last one writes the summary) */
    int32_t farm_hint[32];                          /* +-(worker + 1) of workers whose update queue is open to helpers (slot worker % SLIP_FARM_HINTS, last writer wins; hints; negative: the bulk of a committed column) */
    int32_t committer_up, committer_where, padC[30];                 /* the committer workgroup of this launch is running (workers export packages only after they have seen it) */
    int32_t dbg_who, dbg_k, dbg_a, dbg_b;           /* which wait ran into the spin limit (diagnostic) */
    int32_t k_next, status, status_k, solve_next;
    int64_t Lnz, Lnl, Unz, Unl;                     /* mirrors of Lp/Lo/Up/Uo at the frontier (written at kernel end)     */
    int64_t Lnl_exact, Unl_exact;                   /* limbs actually stored                                             */
    int64_t out_used;                               /* solve: limbs of the output slab in use                            */
    unsigned long long c_upd, c_read, c_write, c_src, c_streamed, c_maxdig, c_macs, c_short, c_farm;   /* c_farm: queues opened to helpers (low word), items helpers ran (high word); c_short: columns committed by the short chain (high word: by the committer) */
    unsigned long long c_eng, c_retract;            /* c_eng: columns committed by the committer's chain engine (low word), late sources it applied (high word); c_retract: packages retracted (low), exported again (high) */
    unsigned long long prof[24];                    /* -DSLIP_PROFILE_PHASES builds only */
} SlipState;

/* immutable during a launch: passed by value, copied to LDS, private fields set per worker there */
/* A pointer to data that OTHER workgroups read or write during a launch: it has no operator[] and no
 * operator*, so a plain dereference -- a load the compiler may keep in a register or serve from a stale line, a store that
 * never leaves the L1 -- does not compile.  at(i) is the address for the accessors of wave_shim.h (slip_ld_* / slip_st_*:
 * sc1, agent scope; the atomics); fixed(i) is a plain read for launches in which nobody writes the array (the solves).
 * The host side reaches the raw pointer as p_ (allocation, copies, kernel arguments). */
#if defined(SLIP_EMULATE)
#define SLIP_HD
#else
#define SLIP_HD __host__ __device__ __forceinline__
#endif
template <class T> struct slip_shared {
    T *p_;
    SLIP_HD T *at(int64_t i = 0) const { return p_ + i; }
    SLIP_HD T fixed(int64_t i) const { return p_[i]; }
};

typedef struct SlipParams {
    int32_t n, pivot_scheme, limb_cap, tol_mode;    /* tol_mode 0: tol <= 0          */
    uint64_t tol_m; int32_t tol_e, k_stop;          /* tol = tol_m * 2^tol_e         */
    const int64_t *Ap; const int32_t *Ai; const int32_t *Alen; const int64_t *Aoff;
    const uint64_t *Alimbs; const int32_t *q;
    slip_shared<int32_t> pinv, row_perm;            /* shared: swapped at stage 1 of every column */
    SlipRow *xrow; uint32_t *xd;                    /* PRIVATE per worker (the kernel offsets the bases): row i's digits at xd[i*xcap] */
    slip_shared<SlipPiv> piv; slip_shared<uint32_t> invd;   /* shared: pivot p's inverse at invd[p*invcap] */
    int32_t xcap, invcap, wcap, bm_words;
    int64_t *Lp, *Lo; int32_t *Li; SlipEnt *Le; uint64_t *Llimbs; int64_t Lcap_nz, Lcap_nl;   /* Lo: limb offset of a column's first entry */
    int64_t *Up, *Uo; int32_t *Ui; SlipEnt *Ue; uint64_t *Ulimbs; int64_t Ucap_nz, Ucap_nl;
    slip_shared<int32_t> Lready;                    /* shared: column c's L entries and limbs are published (stage 2)  */
    int32_t *pat;                                   /* PRIVATE: pattern of the column (positions, ascending) when it exceeds the LDS cap */
    int32_t *rlist;                                 /* PRIVATE: rows of the pattern in discovery order                  */
    int32_t *rpos;                                  /* PRIVATE: their positions at commit time (patterns beyond the LDS cap) */
    int32_t *srow;                                  /* PRIVATE: rows in pattern order (patterns beyond the LDS cap)       */
    uint32_t *gscratch, *gbitmap;                   /* PRIVATE: used when LDS does not hold them */
    int32_t k0, t0;                                 /* ticket t0 + d is column k0 + d in this launch                    */
    int32_t bitmap_in_lds, scratch_in_lds;          /* where the bitmap / wave scratch live (generic kernel)            */
    int32_t nworkers, worker;
    int32_t no_early;                               /* diagnostics: 1 = every column takes the complete path (no early commit) */
    int32_t committer;                              /* 1: block 0 of the launch is the committer (ref_lu_pipe_commit.h), the others are column workers */
    int32_t quiet_neighbours;                       /* workers on CUs within this distance of the committer's stand aside (they share its instruction cache) */
    int32_t engine;                                 /* 1: the committer keeps a mirror of pinv in LDS and runs the chain engine (full packages of short one-limb columns) */
    slip_shared<uint32_t> pkg;                      /* shared: one package slot per worker (SLIP_PKG_WORDS words each)      */
    slip_shared<uint32_t> jobs; int32_t farm, in_factor; SlipState *st;     /* in_factor: a factorisation launch (the stop word names columns) */            /* shared: one job slot per worker (SLIP_JOB_WORDS words): a long update queue other workers help with */
    slip_shared<int32_t> sw_row, sw_pos;            /* shared: the swap log -- column c's pivot changed places with row sw_row[c] (= row_perm[c] before), which moved to position sw_pos[c] (= the pivot row's position before) */
    int64_t priv_rows;                              /* rows per worker of the private arrays (= n)                      */
    int32_t *dbg;
} SlipParams;

/* arguments of the REF triangular solves (SLIP_LU_solve.c:41-86) on resident factors */
typedef struct SlipSolveArgs {
    int32_t nrhs, pad;
    const int32_t *blen; const int64_t *boff; const uint64_t *blimbs;   /* dense b, entry (c,i) at c*n+i: signed digits, limb offset */
    int32_t *olen; int64_t *ooff; uint64_t *olimbs; int64_t ocap;       /* numerators over det = rho[n-1], by pivot position        */
    int64_t ostride;                                                     /* limbs of output slab reserved per right-hand side        */
} SlipSolveArgs;

#if defined(SLIP_PROFILE_PHASES) && !defined(SLIP_EMULATE)
#define SLIP_STAMP(slot) do { if (tid == 0) { unsigned long long now_ = clock64(); prof_[slot] += now_ - t_prev_; t_prev_ = now_; if (hstamp_) hstamp_[slot] = (int32_t) slip_realtime(); } } while (0)
#define SLIP_PROFILING 1
#define SLIP_STAMP_INIT() unsigned long long t_prev_ = clock64(); unsigned long long prof_[24] = {0}; int32_t *hstamp_ = (int32_t *) 0
#define SLIP_STAMP_FLUSH(st) do { if (tid == 0) for (int s_ = 0; s_ < 24; s_++) if (prof_[s_]) slip_agent_add_u64(&(st)->prof[s_], prof_[s_]); } while (0)
#else
#define SLIP_STAMP(slot) do { } while (0)
#define SLIP_STAMP_INIT() do { } while (0)
#define SLIP_STAMP_FLUSH(st) do { } while (0)
#endif

/* LDS layout in 32-bit words */
#define SLIP_SCRATCH_WAVES 16       /* waves a worker's share of the global scratch is sized for */
#define SLIP_LDS_VARS      0        /* 64 words of workgroup-shared scalars          */
#define SLIP_LDS_SCAN      64       /* 128 words: per-wave partials of scans/reductions */
#define SLIP_LDS_WORK      192      /* work lists: 2 x SLIP_WORK_CAP (m, i) pairs, or 1 x rows + 1 x 5-word row records */
#define SLIP_WORK_CAP      512
#define SLIP_WORK_WORDS    (6 * SLIP_WORK_CAP)
#define SLIP_FAST_CAP       SLIP_TAB_CAP    /* patterns up to this many rows may commit their pivot early */
#define SLIP_CAND_CAP       256             /* ... if no more than this many rows are candidates */
#define SLIP_LDS_TAB       (SLIP_LDS_WORK + SLIP_WORK_WORDS)   /* column table: row, len, bits, slab offset per pattern entry */
#define SLIP_TAB_CAP       1024
#define SLIP_PAT_CAP       1024     /* a pattern of at most this many entries stays in LDS               */
#define SLIP_LDS_PAT       (SLIP_LDS_TAB + 4 * SLIP_TAB_CAP)
#define SLIP_LDS_ROWS      (SLIP_LDS_PAT + SLIP_PAT_CAP)     /* row id of every pattern entry (same cap)        */
#define SLIP_LDS_DIROFF    (SLIP_LDS_ROWS + SLIP_PAT_CAP)    /* slab offset of a row multiplied straight into L */
#define SLIP_LDS_KEYS      (SLIP_LDS_DIROFF + SLIP_PAT_CAP)  /* leading 64 bits of a row multiplied straight into L */
#define SLIP_LDS_BITMAP    (SLIP_LDS_KEYS + 2 * SLIP_PAT_CAP)

enum { SV_ERR = 0, SV_CNT0 = 1 /* 3 rotating work counters */, SV_MAXDIG = 4, SV_F = 5 /* frontier as this worker knows it */,
       SV_LISTN = 6, SV_TMP = 7,
       SV_LNZ = 8 /* int64 slots from here */, SV_LNL = 10, SV_UNZ = 12, SV_UNL = 14,
       SV_LALLOC = 16 /* limbs of the L slab handed out to this column's direct rows */, SV_LEXACT = 18,
       SV_NROWS = 24 /* rows discovered so far (length of rlist) */, SV_K = 25, SV_TAG = 26, SV_ABORT = 27,
       SV_TMP2 = 28, SV_F2 = 29 /* ready frontier as this worker knows it */, SV_TMP3 = 30, SV_ACNT = 31 /* class-A rows of the early commit (zeroed at column start) */,
       SV_EPR = 32, SV_EPP = 33, SV_EST = 34 /* early commit: pivot row, its position, status (written by wave 0) */,
       SV_PP = 36 /* SLIP_PP_WORDS words: what the pre-pass of the commit chain found (slip_prepass) */,
       SV_PPF = 50 /* pre-pass: non-pivotal rows of a column that can travel as a FULL package (every value one limb), or -1 */,
       SV_PKGK = 51 /* kind of the exported package: 0 candidates, 1 full */, SV_NOK1 = 52 /* 1: no (more) full packages for this column */,
       SV_CUP = 53 /* the committer has been seen running */, SV_PPFL = 55 /* the frontier (threshold) the last pre-pass ran at */, SV_NOENG = 56 /* the column at which this WORKER last saw a pivot of more than one limb (full packages are not tried for a while); survives the columns */, SV_K1STAMP = 54 /* a full package is only exported beyond this frontier (the stamp of one that was sent back) */,
       SV_PKGVER = 60 /* version of this worker's exported package (0: none yet) */, SV_PKGX = 61 /* 1: exported and still valid */,
       SV_PKGF = 62 /* the frontier the package's positions were read at */ };
#define SLIP_PP_WORDS  14
/* a column's package for the committer (ref_lu_pipe_commit.h), offsets in 32-bit words */
#define SLIP_PKG_CANDS   16       /* a package lists at most this many candidates ... */
#ifndef SLIP_PKG_NROWMAX
#define SLIP_PKG_NROWMAX 256      /* ... of a pattern of at most this many rows (512 measured slower: the committer reads every row) */
#endif
#define SLIP_PKG_HDR     0        /* 64-bit {k+1, version}: even = valid, odd = being written or retracted; a column may export again */
#define SLIP_PKG_STAMP   2
#define SLIP_PKG_NROWS   3
#define SLIP_PKG_SUMS    4        /* SLIP_PP_WORDS words */
#define SLIP_PKG_STAMP0  18
#define SLIP_PKG_VER     19       /* the version the sums belong to (every candidate record carries it too) */
#define SLIP_PKG_WORKER  20       /* the exporting worker: its verdict goes to ITS mailbox (the slot may be reused by column k + nworkers before the worker has read it) */
#define SLIP_PKG_KIND    21       /* 0: candidates + the rows of the pattern (a late source sends it back); 1: FULL -- every non-pivotal row with its value */
#define SLIP_PKG_NFULL   22       /* kind 1: rows carried */
#define SLIP_PKG_FULLMAX 128      /* a full package carries at most this many non-pivotal rows ... */
#define SLIP_ENG_ROWS    256      /* ... which late sources may fill up to this many in the committer's chain engine */
#define SLIP_MIRROR_MAX  16384    /* the chain engine keeps pinv in LDS (16-bit): matrices up to this dimension */
#define SLIP_MBOX_HDR    32       /* a worker's mailbox, behind the package slots: the outcome words, then (full packages) the rows handed back */
#define SLIP_MBOX_WORDS  (SLIP_MBOX_HDR + 4 * SLIP_ENG_ROWS)
#define SLIP_PKG_OUT     0        /* the outcome words, as offsets into the exporting worker's MAILBOX (P.pkg.at() + nworkers * SLIP_PKG_WORDS + worker * SLIP_MBOX_WORDS) */
#define SLIP_PKG_CAND    64       /* 6 words per candidate: table index, value (2), aux, position, version */
#define SLIP_PKG_ROWS    160      /* kind 0: the rows of the pattern; kind 1: four arrays of SLIP_PKG_FULLMAX words: row, value (2), sign | history */
#define SLIP_PKG_WORDS   704

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

/* workgroup minimum of a 64-bit key (all threads get it) */
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

/* workgroup sum of four 64-bit counters (thread 0's view is complete; others too) */
SLIP_DEV void slip_block_sum4(unsigned long long v[4], uint64_t *tmp)
{
    const int lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    for (int q = 0; q < 4; q++)
        for (int d = 32; d >= 1; d >>= 1) v[q] += slip_shfl_u64(v[q], lane ^ d);
    if (lane == 0) for (int q = 0; q < 4; q++) tmp[4 * wave + q] = v[q];
    slip_block_sync();
    for (int q = 0; q < 4; q++) { unsigned long long s = 0; for (int w = 0; w < nw; w++) s += tmp[4 * w + q]; v[q] = s; }
    slip_block_sync();
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

/* ------------------------------------------------------------------ */
/* shared records: every field through sc1 loads / stores              */
/* ------------------------------------------------------------------ */
SLIP_DEV SlipPiv slip_ld_piv(const SlipPiv *p)
{
    const uint64_t *w = (const uint64_t *) p;
    const uint64_t w0 = slip_ld_u64(w), w1 = slip_ld_u64(w + 1), w2 = slip_ld_u64(w + 2), w3 = slip_ld_u64(w + 3), w4 = slip_ld_u64(w + 4);
    SlipPiv r;
    r.off = (int64_t) w0; r.len = (int32_t)(uint32_t) w1; r.bits = (int32_t)(uint32_t)(w1 >> 32);
    r.lo = w2; r.ctz = (int32_t)(uint32_t) w3; r.invlen = (int32_t)(uint32_t)(w3 >> 32); r.inv64 = w4; r.pad = 0;
    return r;
}
SLIP_DEV void slip_st_piv(SlipPiv *p, const SlipPiv &v)
{
    uint64_t *w = (uint64_t *) p;
    slip_st_u64(w, (uint64_t) v.off);
    slip_st_u64(w + 1, (uint64_t)(uint32_t) v.len | ((uint64_t)(uint32_t) v.bits << 32));
    slip_st_u64(w + 2, v.lo);
    slip_st_u64(w + 3, (uint64_t)(uint32_t) v.ctz | ((uint64_t)(uint32_t) v.invlen << 32));
    slip_st_u64(w + 4, v.inv64);
    slip_st_u64(w + 5, 0ull);
}
SLIP_DEV SlipEnt slip_ld_ent(const SlipEnt *p)
{
    const uint64_t *w = (const uint64_t *) p;
    const uint64_t w0 = slip_ld_u64(w), w1 = slip_ld_u64(w + 1);
    SlipEnt r; r.off = (int64_t) w0; r.len = (int32_t)(uint32_t) w1; r.bits = (int32_t)(uint32_t)(w1 >> 32);
    return r;
}
SLIP_DEV void slip_st_ent(SlipEnt *p, const SlipEnt &v)
{
    uint64_t *w = (uint64_t *) p;
    slip_st_u64(w, (uint64_t) v.off);
    slip_st_u64(w + 1, (uint64_t)(uint32_t) v.len | ((uint64_t)(uint32_t) v.bits << 32));
}
SLIP_DEV int32_t slip_piv_invlen(const SlipPiv *p) { return slip_ld_i32(&p->invlen); }

/* the frontier word: F in the low half, row_perm[F-1] in the high half (one 8-byte granule, one store, one load) */
SLIP_DEV int slip_ld_frontier(const SlipState *st, int *pivrow)
{
    const uint64_t w = slip_ld_u64((const uint64_t *) &st->F);
    *pivrow = (int)(uint32_t)(w >> 32);
    return (int)(uint32_t) w;
}
SLIP_DEV void slip_st_frontier(SlipState *st, int F, int pivrow)
{
    slip_st_u64((uint64_t *) &st->F, (uint64_t)(uint32_t) F | ((uint64_t)(uint32_t) pivrow << 32));
}

SLIP_DEV const dig_t *slip_piv_digits(const SlipParams &P, const SlipPiv &pv) { return (const dig_t *)(P.Llimbs + pv.off); }

SLIP_DEV SlipPiv slip_piv_none(void)
{
    SlipPiv p; p.off = 0; p.len = 0; p.bits = 0; p.lo = 1; p.ctz = 0; p.invlen = 0; p.inv64 = 1; p.pad = 0;
    return p;
}

/* register loads / stores of big integers by kind of memory:
 *   _s: shared during the launch (L slab, inverse cache): sc1
 *   _g: this worker's own global data (its x rows): plain global */
template <int D> SLIP_DEV WR<D> wr_load_s(const dig_t *p, int len)
{
    const int lane = slip_lane();
    WR<D> x;
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; x.d[q] = c < len ? slip_ld_u32(p + c) : 0u; }
    return x;
}
template <int D> SLIP_DEV void wr_store_s(dig_t *p, const WR<D> &x, int count)
{
    const int lane = slip_lane();
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; if (c < count) slip_st_u32(p + c, x.d[q]); }
}
template <int D> SLIP_DEV WR<D> wr_load_g(const dig_t *p, int len)
{
    const int lane = slip_lane();
    WR<D> x;
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; x.d[q] = c < len ? slip_gld_u32(p + c) : 0u; }
    return x;
}
template <int D> SLIP_DEV void wr_store_g(dig_t *p, const WR<D> &x, int count)
{
    const int lane = slip_lane();
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; if (c < count) slip_gst_u32(p + c, x.d[q]); }
}

/* copy `count` digits of shared data into wave scratch (for the routines that index operands freely) */
SLIP_DEV void slip_stage_shared(dig_t *dst, const dig_t *src, int count)
{
    for (int c = slip_lane(); c < count; c += SLIP_WAVE) dst[c] = slip_ld_u32(src + c);
    slip_wave_sync();
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
    /* Newton on the low word first (32-bit multiplies are a quarter of the work of 64-bit ones): 3 -> 6 -> 12 -> 24 -> 48 bits */
    const uint32_t d32 = (uint32_t) d;
    uint32_t x = d32;                                /* d * d = 1 mod 8 */
    x *= 2u - d32 * x; x *= 2u - d32 * x; x *= 2u - d32 * x; x *= 2u - d32 * x;
    uint64_t X = x;
    X *= 2 - d * X;                                  /* 32 -> 64 bits */
    return X;
}
/* v / d for an exact division whose quotient is known to fit 64 bits (bits(v) - bits(d) + 1 <= 64): the low 64 bits of
 * (v >> ctz) times the 64-bit inverse of d's odd part ARE the quotient */
SLIP_DEV uint64_t slip_divexact_to64(slip_u128 v, int z, uint64_t inv64) { return (uint64_t)(v >> z) * inv64; }
/* v / d exactly, v < 2^128, d a one-limb divisor with ctz z and 64-bit inverse of its odd part */
SLIP_DEV slip_u128 slip_divexact128(slip_u128 v, uint64_t d, int z, uint64_t inv64)
{
    const uint64_t dodd = d >> z;
    slip_u128 inv = (slip_u128) inv64;
    inv = inv * ((slip_u128) 2 - (slip_u128) dodd * inv);          /* 128-bit inverse by one Newton step */
    return (v >> z) * inv;
}
/* the 64-bit magnitude of a value of at most 2 digits, read as one aligned limb (private x row) */
SLIP_DEV uint64_t slip_limb0(const dig_t *p) { return *(const uint64_t *) p; }
/* the same from shared data (L slab) */
SLIP_DEV uint64_t slip_limb0_s(const dig_t *p) { return slip_ld_u64((const uint64_t *) p); }

/* store a value of at most 4 digits (magnitude mag, sign sgn) as row i of the private vector */
SLIP_DEV void slip_store_small(const SlipParams &P, int i, slip_u128 mag, int sgn, int h, int tag)
{
    const int bits = slip_bits128(mag), len = (bits + 31) >> 5;
    uint64_t *X = (uint64_t *)(P.xd + (int64_t) i * P.xcap);
    X[0] = (uint64_t) mag;
    if (len > 2) X[1] = (uint64_t)(mag >> 64);
    SlipRow r; r.len = sgn < 0 ? -len : len; r.h = h; r.bits = bits; r.tag = tag;
    P.xrow[i] = r;
}

/* ------------------------------------------------------------------ */
/* wave-level pieces                                                    */
/* ------------------------------------------------------------------ */
/* Shared inverse cache: inv(rho_p >> ctz) modulo B^invlen at invd[p*invcap].  Any worker may extend it: the
 * digits of a 2-adic inverse are unique, so concurrent extensions store identical values; a writer drains
 * its sc1 stores before it raises invlen (atomic max), a reader loads invlen first. */

/* Operands wider than 256 digits go through the routines of wave_bigint.h, which index memory freely with
 * plain accesses.  For these (rare, long) items the visibility rules are met with fences instead of sc1
 * accesses: an agent-scope acquire (L1 invalidate) before shared data is read, an agent-scope release (L2
 * write-back) after the inverse cache has been extended. */
SLIP_DEV int slip_ensure_inv(const SlipParams &P, int p, int want, dig_t *b0, dig_t *b1, dig_t *b2)
{
    int have = (int) slip_bcast0_u32((uint32_t) slip_piv_invlen(P.piv.at(p)));
    slip_agent_acquire();                     /* digits below `have` (and, if enough, all we need) are readable now */
    if (have >= want) return 0;
    if (want > P.invcap || want > P.wcap) return 1;
    int target = 2 * have > want ? 2 * have : want;
    if (target > P.invcap) target = P.invcap;
    if (target > P.wcap) target = P.wcap;
    const SlipPiv pv = slip_ld_piv(P.piv.at(p));
    const int ld = slip_abs(pv.len), z = pv.ctz;
    int lodd = ld - (z >> 5);
    if (lodd > target) lodd = target;
    wb_copy_shr(b0, slip_piv_digits(P, pv), ld, z, lodd);
    dig_t *inv = P.invd.at() + (int64_t) p * P.invcap;
    wb_inv_extend(inv, have, target, b0, lodd, b1, b2);
    slip_vm_drain();
    slip_agent_release();
    if (slip_lane() == 0) slip_agent_max_i32(&P.piv.at(p)->invlen, target);
    slip_wave_sync();
    return 0;
}

/* store a W-digit result (normalising, zero padded to whole limbs) as row i; 1 if it does not fit */
SLIP_DEV int slip_store_x(const SlipParams &P, int i, const dig_t *q, int W, int sign, int h, int tag)
{
    const int lane = slip_lane();
    const int len = wb_len(q, W);
    if (len > P.xcap) return 1;
    dig_t *X = P.xd + (int64_t) i * P.xcap;
    const int lw = (len + 1) & ~1;
    for (int c = lane; c < lw; c += SLIP_WAVE) X[c] = c < len ? q[c] : 0u;
    if (lane == 0) {
        SlipRow r; r.len = sign < 0 ? -len : len; r.h = h; r.tag = tag;
        r.bits = len ? 32 * len - slip_clz32(q[len - 1]) : 0;
        P.xrow[i] = r;
    }
    slip_wave_sync();
    return 0;
}

/* ---- register-resident versions (operands of at most 64*D digits; wave_bigint_reg.h) ---- */

/* the odd part of pivot pv (its digits shifted down by its trailing zero bits), low 64*D digits; b0: scratch for wide shifts */
template <int D> SLIP_DEV WR<D> slip_piv_odd_reg(const SlipParams &P, const SlipPiv &pv, dig_t *b0)
{
    const int ld = slip_abs(pv.len);
    WR<D> dodd;
    if (ld > 64 * D) {               /* the shift must see the digits above the register window */
        const dig_t *src = slip_piv_digits(P, pv);
        const int sw = pv.ctz >> 5, sb = pv.ctz & 31, lane = slip_lane();
#pragma unroll
        for (int q = 0; q < D; q++) {
            const int idx = 64 * q + lane + sw;
            const uint32_t lo = idx < ld ? slip_ld_u32(src + idx) : 0u, hi = idx + 1 < ld ? slip_ld_u32(src + idx + 1) : 0u;
            dodd.d[q] = sb ? ((lo >> sb) | (hi << (32 - sb))) : lo;
        }
    } else dodd = wr_shr<D>(wr_load_s<D>(slip_piv_digits(P, pv), ld), pv.ctz, b0);
    return dodd;
}

/* cached inverse of pivot p's odd part to `want` digits, register Newton; b0: scratch for wide shifts */
template <int D> SLIP_DEV int slip_ensure_inv_reg(const SlipParams &P, int p, int want, dig_t *b0)
{
    int have = (int) slip_bcast0_u32((uint32_t) slip_piv_invlen(P.piv.at(p)));
    if (have >= want) return 0;
    if (want > P.invcap) return 1;
    int target = 2 * have > want ? 2 * have : want;
    if (target > P.invcap) target = P.invcap;
    if (target > 64 * D) target = 64 * D;
    const SlipPiv pv = slip_ld_piv(P.piv.at(p));
    const WR<D> dodd = slip_piv_odd_reg<D>(P, pv, b0);
    dig_t *inv = P.invd.at() + (int64_t) p * P.invcap;
    WR<D> V = wr_inv_extend<D>(wr_load_s<D>(inv, have), have, target, dodd);
    wr_store_s<D>(inv, V, target);
    slip_vm_drain();
    if (slip_lane() == 0) slip_agent_max_i32(&P.piv.at(p)->invlen, target);
    slip_wave_sync();
    return 0;
}

/* Y / rho[pd] exactly, Y already shifted down by the divisor's trailing zero bits and cut to the W digits of the quotient;
 * for divisors that belong to ONE row's history (the rho[h] of slip_REF_triangular_solve.c:147,226,255), not to a source
 * every row of the update divides by.  The cached 2-adic inverse is used when it is long enough.  Otherwise the first
 * SLIP_INV_DEMAND divisions by this pivot go digit by digit (wr_div_hensel: half the cost of extending the inverse for ONE
 * use), and a pivot that is asked for again gets its inverse extended.  The count lives in the pivot's record; a stale read
 * changes which way the same quotient is computed, nothing else.
 * Measured (MI355X, kernel ms, never / first two / always digit by digit): C4 window 3.83 / 3.72-3.84 / 3.93, NSR8K 507 /
 * 512 / 526, model6 628 / 627-636 / 644, d18512 550 / 550 / 550: the inverses ARE reused -- the rows of a heavy column share
 * few history levels -- so the Newton extension is paid once per (pivot, width) and a product against the cached inverse
 * (one wr_mul) beats W dependent steps every time after that.  One digit-by-digit division per pivot is the default. */
#ifndef SLIP_INV_DEMAND
#define SLIP_INV_DEMAND 1
#endif
template <int D> SLIP_DEV int slip_div_piv_reg(const SlipParams &P, WR<D> &Y, int W, int pd, const SlipPiv &d, dig_t *b0)
{
    const int have = (int) slip_bcast0_u32((uint32_t) slip_piv_invlen(P.piv.at(pd)));
    if (have < W) {
        const uint32_t asked = slip_bcast0_u32(slip_ld_u32((const uint32_t *) &P.piv.at(pd)->pad));
        if (asked < (uint32_t) SLIP_INV_DEMAND) {
            if (slip_lane() == 0) (void) slip_agent_add_i32((int32_t *) &P.piv.at(pd)->pad, 1)       /* (result unused: the no-return form) */;
            const WR<D> dodd = slip_piv_odd_reg<D>(P, d, b0);
            Y = wr_div_hensel<D>(Y, W, dodd);
            return 0;
        }
        const int e = slip_ensure_inv_reg<D>(P, pd, W, b0);
        if (e) return e;
    }
    const WR<D> I = wr_load_s<D>(P.invd.at() + (int64_t) pd * P.invcap, W);
    Y = wr_mask<D>(wr_mul<D>(I, W, Y), W);
    return 0;
}

/* store the low digits of q (normalising, padded to whole limbs) as row i */
template <int D> SLIP_DEV int slip_store_x_reg(const SlipParams &P, int i, const WR<D> &q, int sign, int h, int tag)
{
    const int len = wr_len<D>(q);
    if (len > P.xcap) return 1;
    dig_t *X = P.xd + (int64_t) i * P.xcap;
    wr_store_g<D>(X, q, (len + 1) & ~1);                 /* digits above len are zero in q */
    const uint32_t top = len ? wr_digit<D>(q, len - 1) : 0u;
    if (slip_lane() == 0) {
        SlipRow r; r.len = sign < 0 ? -len : len; r.h = h; r.tag = tag;
        r.bits = len ? 32 * len - slip_clz32(top) : 0;
        P.xrow[i] = r;
    }
    slip_wave_sync();
    return 0;
}

/* make the cached inverse of pivot p cover `want` digits, with whichever arithmetic fits the width */
SLIP_DEV int slip_ensure_inv_any(const SlipParams &P, int p, int want, dig_t *b0, dig_t *b1, dig_t *b2)
{
    if (want <= 64)  return slip_ensure_inv_reg<1>(P, p, want, b0);
    if (want <= 128) return slip_ensure_inv_reg<2>(P, p, want, b0);
    if (want <= 192) return slip_ensure_inv_reg<3>(P, p, want, b0);
    if (want <= 256) return slip_ensure_inv_reg<4>(P, p, want, b0);
    return slip_ensure_inv(P, p, want, b0, b1, b2);
}

/* History update of row r (slip_REF_triangular_solve.c:139-149, 248-257), one wavefront:
 *     x[r] <- x[r] * rho[pm] / rho[pd]      (pd < 0: no division); the row's history tag becomes newh (SLIP_KEEP_H: stays) */
#define SLIP_KEEP_H (-1000000000)
template <int D> SLIP_DEV int slip_history_wave_reg(const SlipParams &P, int r, int pm, int pd, dig_t *b0, int newh)
{
    SlipRow xr = P.xrow[r];
    if (newh != SLIP_KEEP_H) xr.h = newh;
    const int lx = slip_abs(xr.len);
    const SlipPiv m = slip_ld_piv(P.piv.at(pm));
    const int lm = slip_abs(m.len);
    int sign = slip_sgn(xr.len) * slip_sgn(m.len);
    if (pd >= 0) {
        const SlipPiv d = slip_ld_piv(P.piv.at(pd));
        const int W = (xr.bits + m.bits - d.bits + 1 + 31) >> 5;
        if (W > P.invcap) return 1;
        WR<D> X = wr_load_g<D>(P.xd + (int64_t) r * P.xcap, lx), M = wr_load_s<D>(slip_piv_digits(P, m), lm);
        WR<D> Y = lx <= lm ? wr_mul<D>(X, lx < 64 * D ? lx : 64 * D, M) : wr_mul<D>(M, lm < 64 * D ? lm : 64 * D, X);
        Y = wr_mask<D>(wr_shr<D>(Y, d.ctz, b0), W);
        { const int e = slip_div_piv_reg<D>(P, Y, W, pd, d, b0); if (e) return e; }
        return slip_store_x_reg<D>(P, r, Y, sign * slip_sgn(d.len), xr.h, xr.tag);
    }
    WR<D> X = wr_load_g<D>(P.xd + (int64_t) r * P.xcap, lx), M = wr_load_s<D>(slip_piv_digits(P, m), lm);
    WR<D> Y = lx <= lm ? wr_mul<D>(X, lx, M) : wr_mul<D>(M, lm, X);
    return slip_store_x_reg<D>(P, r, Y, sign, xr.h, xr.tag);
}

/* compare the magnitudes of two register numbers: -1, 0, +1 (all lanes call) */
template <int D> SLIP_DEV int wr_cmp(const WR<D> &A, const WR<D> &B)
{
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {
        const uint64_t df = slip_ballot(A.d[q] != B.d[q]);
        if (df) {
            const int t = 63 - slip_clz64(df);
            return slip_readlane(A.d[q], t) > slip_readlane(B.d[q], t) ? 1 : -1;
        }
    }
    return 0;
}

/* Is |x[r] * rho[pm] / rho[pd]| (pd >= 0: the row was last updated at column pd) on the far side of |a * rho[pm]|?  Nothing is
 * stored: the row keeps its state.  Used by the pre-pass to rule class-B candidates out BEFORE the column's turn: every
 * candidate's final value is its level-pm value times the same rho[k-1] / rho[pm], so the order at level pm is the final
 * order.  Returns +1 / 0 / -1 for |B value| > / = / < |a rho[pm]|, or 2 when the widths do not fit the register path.
 * (The inverse of rho[pd] this leaves in the shared cache is the one the row's final history update needs.) */
template <int D> SLIP_DEV int slip_cand_compare_reg(const SlipParams &P, int r, int pm, int pd, uint32_t a0, uint32_t a1, int nd, dig_t *b0)
{
    const SlipRow xr = P.xrow[r];
    const int lx = slip_abs(xr.len);
    const SlipPiv m = slip_ld_piv(P.piv.at(pm));
    const int lm = slip_abs(m.len);
    const SlipPiv d = slip_ld_piv(P.piv.at(pd));
    const int W = (xr.bits + m.bits - d.bits + 1 + 31) >> 5;
    if (W > P.invcap) return 2;
    const WR<D> M = wr_load_s<D>(slip_piv_digits(P, m), lm);
    WR<D> X = wr_load_g<D>(P.xd + (int64_t) r * P.xcap, lx);
    WR<D> Y = lx <= lm ? wr_mul<D>(X, lx < 64 * D ? lx : 64 * D, M) : wr_mul<D>(M, lm < 64 * D ? lm : 64 * D, X);
    Y = wr_mask<D>(wr_shr<D>(Y, d.ctz, b0), W);
    if (slip_div_piv_reg<D>(P, Y, W, pd, d, b0)) return 2;
    WR<D> A;
    if (nd == 1) A = wr_mul_digit<D>(a0, M);
    else {
        WR<D> S = wr_zero<D>();
        if (slip_lane() == 0) S.d[0] = a0;
        if (slip_lane() == 1) S.d[0] = a1;
        A = wr_mul<D>(S, nd, M);
    }
    return wr_cmp<D>(Y, A);
}
SLIP_DEVN int slip_cand_compare_out(const SlipParams *Pg, int r, int pm, int pd, uint32_t a0, uint32_t a1, int nd, dig_t *b0)
{
    const SlipParams &P = *Pg;
    const SlipRow xr = P.xrow[r];
    const SlipPiv m = slip_ld_piv(P.piv.at(pm)), d = slip_ld_piv(P.piv.at(pd));
    /* widths: the shifted product x * rho[pm] and a * rho[pm] must fit the registers */
    const int Wn = ((xr.bits + m.bits + 31) >> 5) + 1, Wa = slip_abs(m.len) + 2;
    const int Wmax = Wn > Wa ? Wn : Wa;
    if (Wmax > 256 || Wn > P.invcap || (xr.bits + m.bits - d.bits + 1 + 31) / 32 > P.invcap) return 2;
    if (Wmax <= 64)  return slip_cand_compare_reg<1>(P, r, pm, pd, a0, a1, nd, b0);
    if (Wmax <= 128) return slip_cand_compare_reg<2>(P, r, pm, pd, a0, a1, nd, b0);
    if (Wmax <= 192) return slip_cand_compare_reg<3>(P, r, pm, pd, a0, a1, nd, b0);
    return slip_cand_compare_reg<4>(P, r, pm, pd, a0, a1, nd, b0);
}

SLIP_DEV int slip_history_wave(const SlipParams &P, int r, int pm, int pd, dig_t *b0, dig_t *b1, dig_t *b2, int newh)
{
    SlipRow xr = P.xrow[r];
    if (newh != SLIP_KEEP_H) xr.h = newh;             /* the history tag the updated row carries */
    const int lx = slip_abs(xr.len);
    const dig_t *X = P.xd + (int64_t) r * P.xcap;
    const SlipPiv m = slip_ld_piv(P.piv.at(pm));
    const int lm = slip_abs(m.len);
    int sign = slip_sgn(xr.len) * slip_sgn(m.len);
    int bq = xr.bits + m.bits;
    {
        /* widths: W digits of result; the shifted product needs W + ceil(ctz/32) */
        int Wn = (bq + 31) >> 5;
        if (pd >= 0) { const SlipPiv d0 = slip_ld_piv(P.piv.at(pd)); Wn = ((bq - d0.bits + 1 + 31) >> 5) + ((d0.ctz + 31) >> 5); }
        if (Wn > P.wcap) return 1;
        if (Wn <= 64)  return slip_history_wave_reg<1>(P, r, pm, pd, b0, newh);
        if (Wn <= 128) return slip_history_wave_reg<2>(P, r, pm, pd, b0, newh);
        if (Wn <= 192) return slip_history_wave_reg<3>(P, r, pm, pd, b0, newh);
        if (Wn <= 256) return slip_history_wave_reg<4>(P, r, pm, pd, b0, newh);
    }
    /* wide operands (see slip_ensure_inv) */
    slip_agent_acquire();
    if (pd < 0) {
        const int W = (bq + 31) >> 5;
        if (W > P.wcap) return 1;
        wb_mul_lo(b0, X, lx, slip_piv_digits(P, m), lm, W);
        return slip_store_x(P, r, b0, W, sign, xr.h, xr.tag);
    }
    const SlipPiv d = slip_ld_piv(P.piv.at(pd));
    bq -= d.bits - 1;
    const int W = (bq + 31) >> 5, zh = d.ctz, W2 = W + ((zh + 31) >> 5);
    if (W2 > P.wcap) return 1;
    { const int e = slip_ensure_inv(P, pd, W, b0, b1, b2); if (e) return e; }
    wb_mul_lo(b0, X, lx, slip_piv_digits(P, m), lm, W2);
    wb_copy_shr(b1, b0, W2, zh, W);
    wb_mul_lo(b2, b1, W, P.invd.at() + (int64_t) pd * P.invcap, W, W);
    return slip_store_x(P, r, b2, W, sign * slip_sgn(d.len), xr.h, xr.tag);
}

/* One IPGE update (slip_REF_triangular_solve.c:156-241) of target row i by source row j
 * (pivot position jn) through the L entry m, one wavefront, everything modulo B^W:
 *     x[i] <- ( hist(x[i]) * rho[jn] - L_m * x[j] ) / rho[jn-1]                          */
struct SlipIpgePlan { int W, W1, W2, hist, hdiv, has_d; };

/* bit bounds -> working widths of one IPGE update (scalar code) */
SLIP_DEV SlipIpgePlan slip_ipge_plan(const SlipParams &P, const SlipRow &xi, const SlipRow &xj, const SlipEnt &le,
                                     const SlipPiv &R, int jn)
{
    SlipIpgePlan pl;
    const int lx = slip_abs(xi.len), hi = xi.h, br = R.bits;
    pl.has_d = jn >= 1;
    int bd = 0, zd = 0;
    if (pl.has_d) { const SlipPiv Dd = slip_ld_piv(P.piv.at(jn - 1)); bd = Dd.bits; zd = Dd.ctz; }
    pl.hist = lx != 0 && pl.has_d && hi < jn - 1;
    pl.hdiv = pl.hist && hi > -1;
    int bh = 0, zh = 0;
    if (pl.hdiv) { const SlipPiv H = slip_ld_piv(P.piv.at(hi)); bh = H.bits; zh = H.ctz; }
    const int bxp = !lx ? 0 : (!pl.hist ? xi.bits : (pl.hdiv ? xi.bits + bd - bh + 1 : xi.bits + bd));
    const int b1b = lx ? bxp + br : 0, b2b = le.bits + xj.bits;
    const int bnum = (b1b > b2b ? b1b : b2b) + 1;
    const int bq = pl.has_d ? bnum - bd + 1 : bnum;
    pl.W = (bq + 1 + 31) >> 5;                       /* + sign bit */
    pl.W1 = pl.W + (pl.has_d ? ((zd + 31) >> 5) : 0);
    pl.W2 = pl.W1 + (pl.hdiv ? ((zh + 31) >> 5) : 0);
    return pl;
}

template <int D> SLIP_DEV int slip_ipge_wave_reg(const SlipParams &P, int i, int j, int jn, int64_t m, dig_t *b0,
                                                 const SlipRow &xi, const SlipRow &xj, const SlipEnt &le, const SlipPiv &R,
                                                 int W, int W1, int hist, int hdiv)
{
    const int has_d = jn >= 1;
    const int lx = slip_abs(xi.len);
    SlipPiv Dv = slip_piv_none();
    if (has_d) Dv = slip_ld_piv(P.piv.at(jn - 1));
    if (hdiv && W1 > P.invcap) return 1;
    if (has_d) { const int e = slip_ensure_inv_reg<D>(P, jn - 1, W, b0); if (e) return e; }
    const int CAP = 64 * D;
    const int lr = slip_abs(R.len) < W1 ? slip_abs(R.len) : W1;
    WR<D> Rr = wr_load_s<D>(slip_piv_digits(P, R), lr);
    /* P1 = hist(x_i) * rho_jn  (mod B^W1), sign s1 */
    int s1 = slip_sgn(xi.len) * slip_sgn(R.len);
    WR<D> P1 = wr_zero<D>();
    if (lx) {
        const int lxe = lx < CAP ? lx : CAP;
        WR<D> Y = wr_load_g<D>(P.xd + (int64_t) i * P.xcap, lxe);
        int ly = lxe;
        if (hist) {
            const int ld = slip_abs(Dv.len) < CAP ? slip_abs(Dv.len) : CAP;
            WR<D> Dd = wr_load_s<D>(slip_piv_digits(P, Dv), ld);
            Y = ly <= ld ? wr_mul<D>(Y, ly, Dd) : wr_mul<D>(Dd, ld, Y);
            s1 *= slip_sgn(Dv.len);
            ly = W1;
            if (hdiv) {
                const SlipPiv H = slip_ld_piv(P.piv.at(xi.h));
                Y = wr_mask<D>(wr_shr<D>(Y, H.ctz, b0), W1);
                { const int e = slip_div_piv_reg<D>(P, Y, W1, xi.h, H, b0); if (e) return e; }
                s1 *= slip_sgn(H.len);
            }
            Y = wr_mask<D>(Y, W1);
        }
        P1 = ly <= lr ? wr_mul<D>(Y, ly, Rr) : wr_mul<D>(Rr, lr, Y);
    }
    /* P2 = L_m * x_j, sign s2 */
    const int ll = slip_abs(le.len) < W1 ? slip_abs(le.len) : W1, lj = slip_abs(xj.len) < W1 ? slip_abs(xj.len) : W1;
    WR<D> Lm = wr_load_s<D>((const dig_t *)(P.Llimbs + le.off), ll), Xj = wr_load_g<D>(P.xd + (int64_t) j * P.xcap, lj);
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
        WR<D> ID = wr_load_s<D>(P.invd.at() + (int64_t)(jn - 1) * P.invcap, W);
        T = wr_mask<D>(wr_mul<D>(ID, W, T), W);
        sT *= slip_sgn(Dv.len);
    }
    if (wr_digit<D>(T, W - 1) >> 31) { T = wr_mask<D>(wr_addsub<D>(wr_zero<D>(), T, 1), W); sT = -sT; }
    (void) m;
    return slip_store_x_reg<D>(P, i, T, sT, jn, xi.tag);
}

SLIP_DEV int slip_ipge_wave(const SlipParams &P, int i, int j, int jn, int64_t m, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const SlipRow xi = P.xrow[i], xj = P.xrow[j];
    const SlipEnt le = slip_ld_ent(&P.Le[m]);
    const SlipPiv R = slip_ld_piv(P.piv.at(jn));
    const int lx = slip_abs(xi.len), sx = slip_sgn(xi.len);
    const dig_t *X = P.xd + (int64_t) i * P.xcap;
    const int lr = slip_abs(R.len), sr = slip_sgn(R.len);
    const int ll = slip_abs(le.len), sl = slip_sgn(le.len);
    const dig_t *Lm = (const dig_t *)(P.Llimbs + le.off);
    const int lj = slip_abs(xj.len), sj = slip_sgn(xj.len);
    const dig_t *Xj = P.xd + (int64_t) j * P.xcap;
    const int hi = xi.h;
    const SlipIpgePlan pl = slip_ipge_plan(P, xi, xj, le, R, jn);
    const int has_d = pl.has_d, hist = pl.hist, hdiv = pl.hdiv, W = pl.W, W1 = pl.W1, W2 = pl.W2;
    if (W2 > P.wcap) return 1;
    /* operands that fit 256 digits stay in registers */
    if (W2 <= 64)  return slip_ipge_wave_reg<1>(P, i, j, jn, m, b0, xi, xj, le, R, W, W1, hist, hdiv);
    if (W2 <= 128) return slip_ipge_wave_reg<2>(P, i, j, jn, m, b0, xi, xj, le, R, W, W1, hist, hdiv);
    if (W2 <= 192) return slip_ipge_wave_reg<3>(P, i, j, jn, m, b0, xi, xj, le, R, W, W1, hist, hdiv);
    if (W2 <= 256) return slip_ipge_wave_reg<4>(P, i, j, jn, m, b0, xi, xj, le, R, W, W1, hist, hdiv);
    SlipPiv D = slip_piv_none();
    if (has_d) D = slip_ld_piv(P.piv.at(jn - 1));
    const int ld = slip_abs(D.len), sd = has_d ? slip_sgn(D.len) : 1, zd = D.ctz;
    int zh = 0, sh = 1;
    if (hdiv) { const SlipPiv H = slip_ld_piv(P.piv.at(hi)); zh = H.ctz; sh = slip_sgn(H.len); }
    if (hdiv) { const int e = slip_ensure_inv(P, hi, W1, b0, b1, b2); if (e) return e; }
    if (has_d) { const int e = slip_ensure_inv(P, jn - 1, W, b0, b1, b2); if (e) return e; }
    slip_agent_acquire();                      /* wide operands (see slip_ensure_inv): plain loads from here on */
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
        wb_mul_lo(b0, b1, W1, P.invd.at() + (int64_t) hi * P.invcap, W1, W1);
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
        wb_mul_lo(b2, b0, W, P.invd.at() + (int64_t)(jn - 1) * P.invcap, W, W);
        Q = b2; sT *= sd;
    }
    /* two's complement -> sign-magnitude */
    if (Q[W - 1] >> 31) {
        wb_addsub(Q, (const dig_t *) 0, 0, Q, W, W, 1, 0u);
        sT = -sT;
    }
    return slip_store_x(P, i, Q, W, sT, jn, xi.tag);
}

/* is |a| * 2^sa >= |b| * 2^sb ?  a, b normalised and in wave-addressable memory (scratch / private);
 * scratch b0, b1 of wcap digits */
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
SLIP_DEV int slip_history_small(const SlipParams &P, const SlipRow &xr, uint64_t xv, const SlipPiv &m, int pd,
                                slip_u128 *out, int *osgn)
{
    if (slip_abs(xr.len) > 2 || slip_abs(m.len) > 2) return 0;
    slip_u128 y = (slip_u128) xv * m.lo;
    int s = slip_sgn(xr.len) * slip_sgn(m.len);
    if (pd >= 0) {
        const SlipPiv d = slip_ld_piv(P.piv.at(pd));
        if (slip_abs(d.len) > 2) return 0;
        y = slip_divexact128(y, d.lo, d.ctz, d.inv64);
        s *= slip_sgn(d.len);
    }
    *out = y; *osgn = s;
    return 1;
}

/* left-aligned leading 64 bits of a normalised l-digit magnitude; shared != 0: the digits live in the L slab */
SLIP_DEV uint64_t slip_top64(const dig_t *X, int l, int shared)
{
    const uint32_t x1 = shared ? slip_ld_u32(X + l - 1) : X[l - 1];
    const uint32_t x2 = l >= 2 ? (shared ? slip_ld_u32(X + l - 2) : X[l - 2]) : 0u;
    uint64_t top = ((uint64_t) x1 << 32) | x2;
    const int sh = slip_clz32(x1);
    if (sh) {
        const uint32_t x3 = l >= 3 ? (shared ? slip_ld_u32(X + l - 3) : X[l - 3]) : 0u;
        top = (top << sh) | (uint64_t)(x3 >> (32 - sh));
    }
    return top;
}

/* compare the magnitudes of two normalised numbers of equal length l that may live in shared memory: -1, 0, +1
 * (wave-cooperative; all lanes call) */
SLIP_DEV int slip_cmp_mag(const dig_t *a, int sa, const dig_t *b, int sb, int l)
{
    const int lane = slip_lane();
    for (int base = ((l - 1) >> 6) << 6; base >= 0 && l > 0; base -= SLIP_WAVE) {
        const int c = base + lane;
        const uint32_t av = c < l ? (sa ? slip_ld_u32(a + c) : a[c]) : 0u, bv = c < l ? (sb ? slip_ld_u32(b + c) : b[c]) : 0u;
        const uint64_t df = slip_ballot(av != bv);
        if (df) {
            const int t = 63 - slip_clz64(df);
            const uint32_t at = slip_shfl_u32(av, t), bt = slip_shfl_u32(bv, t);
            return at > bt ? 1 : -1;
        }
    }
    return 0;
}

/* rows[t] (one-limb values, never updated: h < 0) times the long pivot M: the pivot's digits stay in
 * registers, every wave takes rows in turn (slip_REF_triangular_solve.c:248-257 for untouched rows).
 * rec3: word 3 of the row's record = (pattern index << 3) | (negative << 2) | digits of the one-limb value;
 * ctab != null (this workgroup's LDS): also enter the row into the column table and its leading 64 bits into the key
 * list, so that the pivot search does not read back through memory what this CU has just produced */
/* what the early commit wants to know about a candidate row it has just multiplied (LDS arrays indexed by the row's
 * place in the worker's row table): the search key, the low limb and the trailing zeros for the pivot record */
struct SlipCandOut { uint32_t *k0, *k1; uint32_t *inf; dig_t *stage; int slotw, nstage, kind; };

template <int D> SLIP_DEV void slip_mul_row_finish(const SlipParams &P, const SlipPiv &M, const WR<D> &Y, int r, uint32_t rec3, int64_t off,
                                                   uint32_t slot_off, uint32_t *ctab, uint32_t *ckeys, int tag, const SlipCandOut *co = (const SlipCandOut *) 0,
                                                   int li = 0)
{
    const int len = wr_len<D>(Y);
    /* bulk L data: plain (coalesced) stores; the worker's release fence before Lready[k] publishes them (a 4-byte
     * write-through store is one fabric write per lane: 6x the time of these rows).  A candidate of the early commit may
     * become the pivot: its digits are also left in an LDS slot, from where the publishing wave writes them through. */
#ifdef SLIP_BULK_SC1
    wr_store_s<D>((dig_t *)(P.Llimbs + off), Y, (len + 1) & ~1);
#else
    wr_store_g<D>((dig_t *)(P.Llimbs + off), Y, (len + 1) & ~1);
#endif
    const uint32_t d1 = len ? wr_digit<D>(Y, len - 1) : 0u;
    const int neg = (int)((rec3 >> 2) & 1u) ^ (M.len < 0);
    const int32_t slen = neg ? -len : len;
    const int bits = len ? 32 * len - slip_clz32(d1) : 0;
    if (co) {
        const uint32_t d2 = len >= 2 ? wr_digit<D>(Y, len - 2) : 0u, d3 = len >= 3 ? wr_digit<D>(Y, len - 3) : 0u;
        const int staged = li < co->nstage;
        if (staged) {
            dig_t *sl = co->stage + li * co->slotw;
            const int lane = slip_lane();
#pragma unroll
            for (int q = 0; q < D; q++) { const int c = 64 * q + lane; if (c < ((len + 1) & ~1)) sl[c] = Y.d[q]; }
        }
        if (slip_lane() == 0) {
            uint64_t top = ((uint64_t) d1 << 32) | d2;
            const int sh = len ? slip_clz32(d1) : 0;
            if (sh) top = (top << sh) | (uint64_t)(d3 >> (32 - sh));
            const int ti = (int)(rec3 >> 3);
            uint64_t key = ((uint64_t) bits << 40) | (top >> 24);
            if (co->kind == 1) key = ~key;
            co->k0[ti] = (uint32_t) key; co->k1[ti] = (uint32_t)(key >> 32);
            if (staged) co->inf[ti] |= ((uint32_t)(li + 1) << 26) | (neg ? 0x80000000u : 0u);   /* where the publishing wave finds the digits; the sign */
        }
    }
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
        SlipRow nr; nr.len = slen; nr.h = -2;                /* h == -2: never updated (h = -1) AND the value lives in the L slab ... */
        nr.tag = tag;
        nr.bits = bits;
        P.xrow[r] = nr;
        *(int64_t *)(P.xd + (int64_t) r * P.xcap) = off;     /* ... at this limb offset */
    }
}

template <int D> SLIP_DEV int slip_mul_rows_reg(const SlipParams &P, const SlipPiv &M, const dig_t *Md, int md_shared, const uint32_t *recs,
                                                int first, int stride, int nrows, int64_t slab_base, uint32_t *ctab, uint32_t *ckeys, int tag,
                                                const SlipCandOut *co)
{
    const int lane = slip_lane();
    const WR<D> Mr = md_shared ? wr_load_s<D>(Md, slip_abs(M.len)) : wr_load<D>(Md, slip_abs(M.len));
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
        slip_mul_row_finish<D>(P, M, Y0, (int) recs[5 * t], w0, slab_base + (int64_t) recs[5 * t + 4], recs[5 * t + 4], ctab, ckeys, tag, co, t);
        slip_mul_row_finish<D>(P, M, Y1, (int) recs[5 * u], w1, slab_base + (int64_t) recs[5 * u + 4], recs[5 * u + 4], ctab, ckeys, tag, co, u);
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
        slip_mul_row_finish<D>(P, M, Y, (int) recs[5 * t], w, slab_base + (int64_t) recs[5 * t + 4], recs[5 * t + 4], ctab, ckeys, tag, co, t);
    }
    return 0;
}

/* rows [first, first+stride, ...) of a column's one-limb-times-pivot list (5-word records), pivot M = rho[k-1];
 * Md: staged copy of the pivot's digits in LDS (md_shared 0), or the L slab itself (md_shared 1) */
SLIP_DEV int slip_mul_rows_any(const SlipParams &P, const SlipPiv &M, const dig_t *Md, int md_shared, const uint32_t *recs, int first, int stride, int nrows,
                               int64_t slab_base, uint32_t *ctab, uint32_t *ckeys, int tag, const SlipCandOut *co = (const SlipCandOut *) 0)
{
    const int Dm = (slip_abs(M.len) + 2 + 63) >> 6;
    if (Dm <= 1) return slip_mul_rows_reg<1>(P, M, Md, md_shared, recs, first, stride, nrows, slab_base, ctab, ckeys, tag, co);
    if (Dm == 2) return slip_mul_rows_reg<2>(P, M, Md, md_shared, recs, first, stride, nrows, slab_base, ctab, ckeys, tag, co);
    if (Dm == 3) return slip_mul_rows_reg<3>(P, M, Md, md_shared, recs, first, stride, nrows, slab_base, ctab, ckeys, tag, co);
    return slip_mul_rows_reg<4>(P, M, Md, md_shared, recs, first, stride, nrows, slab_base, ctab, ckeys, tag, co);
}

/* ---- REF triangular solves (SLIP_LU_solve.c:41-86): the two extra wave-level operations ---- */

/* x[r] <- x[r] / rho[p], exact (slip_back_sub.c:43: divide by the diagonal of U = the pivot) */
SLIP_DEV int slip_divexact_wave(const SlipParams &P, int r, int p, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const SlipRow xr = P.xrow[r];
    const int lx = slip_abs(xr.len);
    const SlipPiv d = slip_ld_piv(P.piv.at(p));
    const int bq = xr.bits - d.bits + 1;
    const int W = bq > 0 ? (bq + 31) >> 5 : 1, zh = d.ctz, W2 = W + ((zh + 31) >> 5);
    if (W2 > P.wcap) return 1;
    { const int e = slip_ensure_inv_any(P, p, W, b0, b1, b2); if (e) return e; }
    slip_agent_acquire();
    wb_copy_shr(b1, P.xd + (int64_t) r * P.xcap, lx, zh, W);
    wb_mul_lo(b2, b1, W, P.invd.at() + (int64_t) p * P.invcap, W, W);
    return slip_store_x(P, r, b2, W, slip_sgn(xr.len) * slip_sgn(d.len), xr.h, xr.tag);
}

/* x[i] <- x[i] - U_m * x[j]   (slip_back_sub.c:44-50); the factors are not written during a solve: plain loads */
SLIP_DEV int slip_submul_wave(const SlipParams &P, int i, int j, int64_t m, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const SlipRow xi = P.xrow[i], xj = P.xrow[j];
    const SlipEnt ue = P.Ue[m];
    const int lx = slip_abs(xi.len), sx = slip_sgn(xi.len);
    const int bt = (xi.bits > ue.bits + xj.bits ? xi.bits : ue.bits + xj.bits) + 1;
    const int W = (bt + 1 + 31) >> 5;                       /* + sign bit */
    if (W > P.wcap) return 1;
    wb_mul_lo(b2, (const dig_t *)(P.Ulimbs + ue.off), slip_abs(ue.len), P.xd + (int64_t) j * P.xcap, slip_abs(xj.len), W);
    const int s2 = slip_sgn(ue.len) * slip_sgn(xj.len);
    if (lx == 0) return slip_store_x(P, i, b2, W, -s2, xi.h, xi.tag);
    int sT = sx;
    wb_addsub(b1, P.xd + (int64_t) i * P.xcap, lx, b2, W, W, sx == s2);       /* |x| -/+ |U x_j| */
    if (sx == s2 && (b1[W - 1] >> 31)) { wb_addsub(b1, (const dig_t *) 0, 0, b1, W, W, 1, 0u); sT = -sT; }
    (void) b0;
    return slip_store_x(P, i, b1, W, sT, xi.h, xi.tag);
}

/* kind 1: IPGE updates of source (j, jn), items = (m - m0, i) pairs; kind 2: history rows of column k (division by rho[h]);
 * kind 4: x * rho[k-1]; kind 5: back-substitution updates (m - m0, i) of source j */
SLIP_DEV int slip_run_item(const SlipParams &P, int kind, int j, int jn, int k, int64_t m0, uint32_t ia, uint32_t ib,
                           dig_t *b0, dig_t *b1, dig_t *b2)
{
    if (kind == 1) return slip_ipge_wave(P, (int) ib, j, jn, m0 + (int64_t) ia, b0, b1, b2);
    if (kind == 5) return slip_submul_wave(P, (int) ib, j, m0 + (int64_t) ia, b0, b1, b2);
    const int r = (int) ia;
    if (kind == 4) return slip_history_wave(P, r, k - 1, -1, b0, b1, b2, SLIP_KEEP_H);      /* x * rho[k-1] */
    { const int h = P.xrow[r].h; return slip_history_wave(P, r, k - 1, h, b0, b1, b2, k - 1); }   /* now at level k-1 */
}

/* ---- out-of-line entry points (one copy each; the parameters are the workgroup's LDS copy; items travel by value:
 * (m - m0, i) for kinds 1 and 5, (row, -) otherwise) ---- */
SLIP_DEVN int slip_run_item_out(const SlipParams *Pg, int kind, int j, int jn, int k, int64_t m0, uint32_t ia, uint32_t ib,
                                dig_t *b0, dig_t *b1, dig_t *b2)
{
    return slip_run_item(*Pg, kind, j, jn, k, m0, ia, ib, b0, b1, b2);
}
SLIP_DEVN int slip_history_wave_out(const SlipParams *Pg, int r, int pm, int pd, dig_t *b0, dig_t *b1, dig_t *b2)
{
    return slip_history_wave(*Pg, r, pm, pd, b0, b1, b2, SLIP_KEEP_H);
}
SLIP_DEVN int slip_divexact_out(const SlipParams *Pg, int r, int p, dig_t *b0, dig_t *b1, dig_t *b2)
{
    return slip_divexact_wave(*Pg, r, p, b0, b1, b2);
}
/* the exact tolerance test of the diagonal preference: |num| * 2^(-te) >= tol_m * |den|  (slip_get_pivot.c:89-118);
 * num / den in wave-addressable memory */
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

/* ------------------------------------------------------------------ */
/* Helping with a long update queue.  The updates of one source on the rows of a column are independent wave items; when
 * the source arrives late (the row became pivotal a few columns before k) they are the critical path of the whole
 * factorisation while most workers wait for their turn.  The owner publishes such a queue as a JOB in its slot of P.jobs;
 * workers that are waiting take items from it with an atomic counter and run them on the OWNER's private x rows.
 * Slot (words): 0 gate = open bit + 256 * helpers inside (atomics only); 2 kind, 3 j, 4 jn, 5 k, 6-7 m0, 8 items, 11 error;
 * 16 next item; 32.. the items.  Visibility: the owner writes its dirty lines back (agent release) before it opens the
 * gate; a helper invalidates (agent acquire) when it enters and writes back before it leaves; the owner closes the gate,
 * waits for the helpers to leave, writes back and invalidates.  Rows share cache lines: the L2s write back the bytes
 * they own. */
/* Spins are bounded by ITERATION counts (each iteration sleeps): a wait that is never answered ends the launch
 * with SLIPDEV_INTERNAL instead of hanging the device. */
#define SLIP_SPIN_LIMIT 40000000ull
#define SLIP_JOB_WORDS      (32 + 2 * SLIP_WORK_CAP)
#ifndef SLIP_FARM_MIN_ITEMS
#define SLIP_FARM_MIN_ITEMS 16          /* ... and shorter queues neither */
#endif
#ifndef SLIP_POLL_NEAR
#define SLIP_POLL_NEAR 16               /* a worker this close to its turn polls the frontier at the short interval */
#endif
#ifndef SLIP_POLL_MAXREPS
#define SLIP_POLL_MAXREPS 32            /* further away: min(distance, this) long sleeps between two polls */
#endif
#ifndef SLIP_FARM_URGENT_DIST
#define SLIP_FARM_URGENT_DIST 48        /* a waiting worker this close to its own turn only helps with queues the frontier waits for */
#endif
#ifndef SLIP_FARM_FEW_LIMBS
#define SLIP_FARM_FEW_LIMBS 96          /* ... unless every item is at least this long (then two are enough) */
#endif
#ifndef SLIP_FARM_KIND2
#define SLIP_FARM_KIND2     1
#endif
#ifndef SLIP_FARM_KIND2_COST
#define SLIP_FARM_KIND2_COST 16384      /* items * limbs^2: the division queues of committed columns (their readers wait for stage 2);
                                         * measured on the C4 window: 262144 -> 3.63 ms, 65536 -> 3.36, 16384 -> 3.31, 4096 -> 3.38, 1024 -> 3.35 */
#endif
#ifndef SLIP_FARM_NEAR_DIV
#define SLIP_FARM_NEAR_DIV  1125        /* ... and only when the column's turn comes before the worker alone would be done */
#endif
#ifndef SLIP_FARM_MAX_HELPERS
#define SLIP_FARM_MAX_HELPERS 24         /* every helper costs its XCD an L2 invalidate and a write-back */
#endif
#ifndef SLIP_FARM_REMOTE_MAX
#define SLIP_FARM_REMOTE_MAX 4           /* helpers from other dies join only while fewer than this many are inside; the owner's die up to MAX_HELPERS.
                                         * C4 window, (remote, max): (12, 12) 3.40 ms, (0, 12) 3.40, (4, 12) 3.15, (4, 24) 3.05, (2, 24) 3.10, (6, 24) 3.14,
                                         * (4, 32) 3.07, (8, 32) 3.10; (0, x) costs model6 10 %: a few remote helpers are better than none */
#endif
#ifndef SLIP_FARM_MIN_COST
#define SLIP_FARM_MIN_COST  8192        /* items * limbs(rho)^2 below which a queue is not worth publishing */
#endif

/* take items until none is left (all waves of the calling workgroup); items: the owner's list in LDS, or null = the job's copy */
SLIP_DEV int slip_farm_items(const SlipParams &P, uint32_t *jb, int kind, int j, int jn, int k, int64_t m0, int nq, const uint32_t *wl,
                             dig_t *b0, dig_t *b1, dig_t *b2)
{
    int err = 0, cnt = 0;
    for (;;) {
        int t = 0;
        if (slip_lane() == 0) t = slip_agent_add_i32((int32_t *)(jb + 16), 1);
        t = (int) slip_bcast0_u32((uint32_t) t);
        if (t >= nq) break;
        uint32_t it0, it1;                       /* kinds 1 and 5: (entry, row) pairs; kind 2: rows */
        if (kind == 1 || kind == 5) {
            if (wl) { it0 = wl[2 * t]; it1 = wl[2 * t + 1]; }
            else { it0 = slip_ld_u32(jb + 32 + 2 * t); it1 = slip_ld_u32(jb + 32 + 2 * t + 1); }
        } else { it0 = wl ? wl[t] : slip_ld_u32(jb + 32 + t); it1 = 0u; }
        const int e = slip_run_item_out(&P, kind, j, jn, k, m0, it0, it1, b0, b1, b2);
        if (e) err = e;
        cnt++;
    }
    if (!wl && cnt && slip_lane() == 0) slip_agent_add_u64(&P.st->c_farm, (unsigned long long) cnt << 32);
    return err;
}

/* thread 0 of a waiting worker: is there a job to help with?  returns slot + 1, or 0 */
/* urgent_only: the caller's own turn is near -- it only helps with queues the frontier waits for (kind 1: the hint is
 * positive), not with the bulk of committed columns (kind 2: negative) */
/* The hint words are ONE line: they are read in one round of loads (one after the other, each with its own wait, they were
 * most of a waiting worker's poll iteration: 8-16 dependent L2 round trips, 11 us from a commit to the waiting worker seeing
 * it).  hv[]: the line as 64-bit words (slip_farm_hints_load); the slots are walked from the worker's own on, so that the
 * helpers of several open queues spread. */
struct SlipHints { uint64_t w[SLIP_FARM_HINTS / 2]; };
SLIP_DEV SlipHints slip_farm_hints_load(const SlipState *st)
{
    SlipHints H;
#pragma unroll
    for (int q = 0; q < SLIP_FARM_HINTS / 2; q++) H.w[q] = slip_ld_u64((const uint64_t *) &st->farm_hint[2 * q]);
    return H;
}
SLIP_DEV int slip_farm_peek_loaded(const SlipParams &P, const SlipHints &H, int urgent_only)
{
    if (!P.farm) return 0;
    uint64_t any = 0;
#pragma unroll
    for (int q = 0; q < SLIP_FARM_HINTS / 2; q++) any |= H.w[q];
    if (!any) return 0;                                   /* (the usual case: nothing is open) */
    const int start = P.worker & (SLIP_FARM_HINTS - 1);
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
#pragma unroll
        for (int q = 0; q < SLIP_FARM_HINTS; q++) {
            if ((pass == 0) != (q >= start)) continue;
            int h = (int)(int32_t)(uint32_t)(H.w[q >> 1] >> (32 * (q & 1)));
            if (h < 0) { if (urgent_only) continue; h = -h; }
            if (h <= 0 || h - 1 == P.worker || h > P.nworkers) continue;
            const uint32_t *jb_ = P.jobs.at() + (int64_t)(h - 1) * SLIP_JOB_WORDS;
            const uint32_t gate = slip_ld_u32(jb_);
            if (!(gate & 1u)) continue;
            /* the items work on the OWNER's private rows, which live in the owner's XCD's L2: a helper on another die fetches and
             * writes them across the fabric.  Helpers of the owner's die come first; others only while few are inside. */
            if (SLIP_FARM_REMOTE_MAX < SLIP_FARM_MAX_HELPERS && slip_ld_u32(jb_ + 9) != slip_xcc_id() && (int)(gate >> 8) >= SLIP_FARM_REMOTE_MAX) continue;
            return h;
        }
    }
    return 0;
}
SLIP_DEV int slip_farm_peek(const SlipParams &P, SlipState *st, int urgent_only = 0)
{
    if (!P.farm) return 0;
    const SlipHints H = slip_farm_hints_load(st);
    return slip_farm_peek_loaded(P, H, urgent_only);
}

/* all threads of a waiting worker: help with the job in `slot` */
SLIP_DEV void slip_farm_help(const SlipParams &P, SlipState *st, uint32_t *lds, int slot, dig_t *b0, dig_t *b1, dig_t *b2)
{
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint32_t *jb = P.jobs.at() + (int64_t) slot * SLIP_JOB_WORDS;
    (void) st;
    slip_block_sync();
    if (slip_tid() == 0) {
        const int old = slip_agent_add_i32((int32_t *) jb, 256);
        const int in = (old & 1) && (old >> 8) < SLIP_FARM_MAX_HELPERS;
        sv[SV_TMP2] = in;
        if (!in) slip_agent_add_i32((int32_t *) jb, -256);
    }
    slip_block_sync();
    if (!sv[SV_TMP2]) return;
    if (slip_wave() == 0) slip_agent_acquire();
    slip_block_sync();
    const int kind = (int) slip_ld_u32(jb + 2), j = (int) slip_ld_u32(jb + 3), jn = (int) slip_ld_u32(jb + 4), k = (int) slip_ld_u32(jb + 5);
    const int64_t m0 = (int64_t) slip_ld_u64((const uint64_t *)(jb + 6));
    const int nq = (int) slip_ld_u32(jb + 8);
    /* the items work on the owner's private rows: this worker's parameter block points there for the duration */
    SlipParams &Pm = const_cast<SlipParams &>(P);
    SlipRow *my_xrow = P.xrow; uint32_t *my_xd = P.xd;
    slip_block_sync();
    Pm.xrow = my_xrow + (int64_t)(slot - P.worker) * P.priv_rows;
    Pm.xd = my_xd + (int64_t)(slot - P.worker) * P.priv_rows * P.xcap;
    slip_block_sync();
    const int e = slip_farm_items(P, jb, kind, j, jn, k, m0, nq, (const uint32_t *) 0, b0, b1, b2);
    if (e && slip_lane() == 0) slip_st_u32(jb + 11, (uint32_t) e);
    slip_vm_drain();
    slip_block_sync();
    Pm.xrow = my_xrow; Pm.xd = my_xd;
    if (slip_wave() == 0) slip_agent_release();
    slip_block_sync();
    if (slip_tid() == 0) slip_agent_add_i32((int32_t *) jb, -256);
    slip_block_sync();
}

/* drain a queue of wave-level items with this workgroup's waves; errors land in sv[SV_ERR].
 * Called by all threads after a workgroup barrier; returns after a workgroup barrier with every item done. */
SLIP_DEV void slip_drain(const SlipParams &P, uint32_t *lds, int kind, int j, int jn, int k, int64_t m0, int nq,
                         const uint32_t *wl, dig_t *b0, dig_t *b1, dig_t *b2)
{
    const int lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    /* (only a column whose turn is near: further away the worker has the time, and every helper costs its XCD an L2 write-back
     * and invalidate) */
    /* (a handful of items is worth opening too when each of them is huge: model6's columns have fewer than 16 rows of 200-364
     * limbs, 100+ us per item -- 635 -> 530 ms; the stride tells whether such values exist at all, before the pivot is looked at) */
    if (P.farm && (kind == 1 || kind == 5 || (kind == 2 && SLIP_FARM_KIND2)) && !sv[SV_ERR] &&
        (nq >= SLIP_FARM_MIN_ITEMS || (nq >= 2 && P.xcap >= 4 * SLIP_FARM_FEW_LIMBS))) {
        /* kind 2: the rows of a committed column that still need their division (its readers wait for its stage 2).  The
         * protocol carries them and a 700-row column of the C4 window then takes 0.35 ms instead of 1.5 -- but the window as a
         * whole got slower (median 6.37 against 6.06 ms over 30 runs: every helper costs its XCD an L2 invalidate), so: off */
        /* (kind 5, back substitution: the multiplier is x[j]; the solves have no frontier, every long queue is opened) */
        const int lr = kind == 5 ? slip_limbs(P.xrow[j].len) : slip_limbs(slip_ld_piv(P.piv.at(kind == 1 ? jn : k - 1)).len);
        /* the queue alone takes about nq * 8 lr^2 / waves cycles; the frontier moves a column every few microseconds */
        const int64_t cost = (int64_t) nq * lr * lr;
        if ((nq >= SLIP_FARM_MIN_ITEMS || lr >= SLIP_FARM_FEW_LIMBS) && cost >= (kind == 2 ? (int64_t) SLIP_FARM_KIND2_COST : (int64_t) SLIP_FARM_MIN_COST) && (kind == 2 || kind == 5 || !P.in_factor || SLIP_FARM_NEAR_DIV == 0 || (int64_t)(sv[SV_K] - sv[SV_F]) <= cost / ((int64_t) nw * (SLIP_FARM_NEAR_DIV ? SLIP_FARM_NEAR_DIV : 1)) + 2)) {
            /* a long queue of long updates: open it to the workers that are waiting */
            const int tid = slip_tid(), T = slip_nthreads();
            uint32_t *jb = P.jobs.at() + (int64_t) P.worker * SLIP_JOB_WORDS;
            for (int c = tid; c < ((kind == 1 || kind == 5) ? 2 * nq : nq); c += T) slip_st_u32(jb + 32 + c, wl[c]);
            if (tid == 0) {
                slip_st_u32(jb + 2, (uint32_t) kind); slip_st_u32(jb + 3, (uint32_t) j); slip_st_u32(jb + 4, (uint32_t) jn); slip_st_u32(jb + 5, (uint32_t) k);
                slip_st_u64((uint64_t *)(jb + 6), (uint64_t) m0); slip_st_u32(jb + 8, (uint32_t) nq); slip_st_u32(jb + 9, slip_xcc_id()); slip_st_u32(jb + 11, 0u); slip_st_u32(jb + 16, 0u);
            }
            slip_vm_drain();
            slip_block_sync();
            if (tid == 0) { slip_agent_release(); slip_agent_add_i32((int32_t *) jb, 1); slip_st_i32(&P.st->farm_hint[P.worker & (SLIP_FARM_HINTS - 1)], kind != 2 ? P.worker + 1 : -(P.worker + 1)); slip_agent_add_u64(&P.st->c_farm, 1ull); }
            const int e = slip_farm_items(P, jb, kind, j, jn, k, m0, nq, wl, b0, b1, b2);
            if (e && lane == 0) sv[SV_ERR] = e;
            slip_vm_drain();
            slip_block_sync();
            if (tid == 0) {
                slip_agent_add_i32((int32_t *) jb, -1);                          /* closed: nobody new gets in */
                { const int hh = slip_ld_i32(&P.st->farm_hint[P.worker & (SLIP_FARM_HINTS - 1)]); if (hh == P.worker + 1 || hh == -(P.worker + 1)) slip_st_i32(&P.st->farm_hint[P.worker & (SLIP_FARM_HINTS - 1)], 0); }
                unsigned long long spins = 0;
                while ((slip_agent_add_i32((int32_t *) jb, 0) >> 8) != 0) { slip_sleep_short(); if (++spins > SLIP_SPIN_LIMIT) { sv[SV_ERR] = SLIPDEV_INTERNAL; P.st->dbg_who = 5; P.st->dbg_k = sv[SV_K]; P.st->dbg_a = slip_agent_add_i32((int32_t *) jb, 0); break; } }
                const int he = (int) slip_ld_u32(jb + 11);
                if (he) sv[SV_ERR] = he;
                slip_agent_release(); slip_agent_acquire();                      /* the helpers' results, next to what this CU wrote */
            }
            slip_block_sync();
            return;
        }
    }
    if (!sv[SV_ERR])
        for (int t = wave; t < nq; t += nw) {
            /* a column beyond the one that stopped the factorisation can never commit: its long queues are not worth finishing
             * (the launch ends when the last workgroup leaves).  One wave looks at the stop word between its items. */
            if (P.in_factor && kind == 1) {
                if (wave == nw - 1 && lane == 0 && (slip_ld_i64(&P.st->stop) >> 8) < (int64_t) sv[SV_K]) sv[SV_ABORT] = 1;
                if ((int) slip_bcast0_u32((uint32_t) sv[SV_ABORT])) break;      /* one lane's view for the whole wave */
            }
            const int e = slip_run_item_out(&P, kind, j, jn, k, m0, (kind == 1 || kind == 5) ? wl[2 * t] : wl[t], (kind == 1 || kind == 5) ? wl[2 * t + 1] : 0u, b0, b1, b2);
            if (e && lane == 0) sv[SV_ERR] = e;
        }
    slip_block_sync();
    if (P.in_factor && kind == 1 && sv[SV_ABORT] && !sv[SV_ERR]) { slip_block_sync(); if (slip_tid() == 0) sv[SV_ERR] = 100; slip_block_sync(); }    /* = SLIPDEV_ABORTED: the column is given up */
}

#include "ref_lu_pipe_cols.h"
#include "ref_lu_pipe_commit.h"

#endif /* SLIP_REF_LU_PIPE_H */
