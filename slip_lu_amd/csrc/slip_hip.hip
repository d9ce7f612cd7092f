#include <vector>
/* slip_hip.hip -- libslip_hip.so: kernels' entry points and the host side of the
 * C ABI declared in include/slip_hip.h.
 *
 * Host responsibilities (the part of SLIP_LU/Source/SLIP_LU_factorize.c:58-211
 * that is not arithmetic): validate, upload A and q once, size the workers'
 * private scatter vectors and the shared L / U slabs in HBM, launch the grid of
 * column workers (ref_lu_pipe.h), and grow a buffer and relaunch from the first
 * uncommitted column when the kernel asks (the reference doubles L/U at
 * :200-211 and lets GMP realloc x).
 *
 * Built by hipcc for gfx950.  With -DSLIP_EMULATE (tests/emu, g++) the same
 * source runs the kernel lane-by-lane on the CPU for unit tests; that build is
 * never the product.
 */
#include "ref_lu_pipe.h"
#include "slip_matgen.h"
#include "../../include/slip_hip.h"

#ifdef SLIP_EMULATE
#include "hip_rt_emu.h"
#endif

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define SLIP_LDS_MAX_WORDS 40000          /* of the 40960 words (160 KiB) per CU */
#define SLIP_BITMAP_LDS_MAX_WORDS 16384   /* n <= 524288 keeps the pattern bitmap in LDS */

/* ------------------------------------------------------------------ */
/* kernels                                                             */
/* ------------------------------------------------------------------ */
/* register-resident primitives (wave_bigint_reg.h) for the unit tests: op 10 product, 11 add, 12 sub,
 * 13 inverse of odd a, 14 a >> lb (lb = shift in bits), 15 a / b for odd b (Hensel); all modulo B^W, W <= 256 */
template <int D> SLIP_DEV void slip_reg_op_test_d(int op, const uint32_t *A, int la, const uint32_t *B, int lb, int W,
                                                   uint32_t *O, uint32_t *scratch)
{
    WR<D> a = wr_load<D>(A, la < W ? la : W), r;
    if (op == 14) r = wr_shr<D>(a, lb, scratch);
    else if (op == 13) r = wr_inv_extend<D>(wr_zero<D>(), 0, W, a);
    else {
        WR<D> b = wr_load<D>(B, lb < W ? lb : W);
        const int ea = la < W ? la : W, eb = lb < W ? lb : W;
        if (op == 10) r = ea <= eb ? wr_mul<D>(a, ea, b) : wr_mul<D>(b, eb, a);
        else if (op == 15) r = wr_div_hensel<D>(a, W, b);
        else r = wr_addsub<D>(a, b, op == 12);
    }
    wr_store<D>(O, r, W);
}
SLIP_DEV void slip_reg_op_test(int op, const uint32_t *A, int la, const uint32_t *B, int lb, int W, uint32_t *O, uint32_t *scratch)
{
    if (op >= 20) {                                   /* raw cross-lane primitives: W must be 64 */
        const int lane = slip_lane();
        const uint32_t v = A[lane];
        uint32_t r = 0;
        if (op == 24) r = slip_wave_sum_u32(v);
        else if (op == 25) r = slip_wave_max_u32(v);
        else if (op == 26) r = slip_wave_min_u32(v);
        else if (op == 27) r = slip_bcast0_u32(v);
        else if (op == 20) r = slip_dpp_shr1(v, 0xAAAAu);
        else if (op == 21) r = slip_dpp_shl1(v, 0xBBBBu);
        else if (op == 22) r = slip_readlane(v, lb);
        else if (op == 23) r = slip_shfl_up_u32(v, 1);
        O[lane] = r;
        return;
    }
    const int D = (W + 63) >> 6;
    if (D <= 1) slip_reg_op_test_d<1>(op, A, la, B, lb, W, O, scratch);
    else if (D == 2) slip_reg_op_test_d<2>(op, A, la, B, lb, W, O, scratch);
    else if (D == 3) slip_reg_op_test_d<3>(op, A, la, B, lb, W, O, scratch);
    else slip_reg_op_test_d<4>(op, A, la, B, lb, W, O, scratch);
}

/* a worker's view of the parameters: the private arrays start at this workgroup's share */
SLIP_DEV void slip_worker_params(SlipParams *Pw, const SlipParams &P, int worker)
{
    *Pw = P;
    Pw->worker = worker;
    Pw->xrow = P.xrow + (int64_t) worker * P.priv_rows;
    Pw->xd = P.xd + (int64_t) worker * P.priv_rows * P.xcap;
    Pw->pat = P.pat + (int64_t) worker * P.priv_rows;
    Pw->rlist = P.rlist + (int64_t) worker * P.priv_rows;
    Pw->rpos = P.rpos + (int64_t) worker * P.priv_rows;
    Pw->srow = P.srow + (int64_t) worker * P.priv_rows;
    Pw->gbitmap = P.gbitmap + (int64_t) worker * (P.bitmap_in_lds ? 0 : P.bm_words + 64);
    Pw->gscratch = P.gscratch + (int64_t) worker * (P.scratch_in_lds ? 0 : (int64_t) SLIP_SCRATCH_WAVES * 3 * P.wcap);
}

/* the last worker to leave records where the factorisation stands (the host reads the state once) */
SLIP_DEV void slip_worker_exit(const SlipParams &P, SlipState *st)
{
    slip_block_sync();
    if (slip_tid() == 0) {
        if (slip_agent_add_i32(&st->exited, 1) == P.nworkers - 1) {
            int pr_; const int F = slip_ld_frontier(st, &pr_);
            st->k_next = F;
            st->Lnz = slip_ld_i64(&P.Lp[F]); st->Lnl = slip_ld_i64(&P.Lo[F]);
            st->Unz = slip_ld_i64(&P.Up[F]); st->Unl = slip_ld_i64(&P.Uo[F]);
        }
    }
}

/* Subtree farm: rescale a block's factors by the pivots the OTHER blocks produced meanwhile.
 * One wavefront per stored entry: out = entry * scale[s], s = the entry's column (L) or the pivot position of its
 * row (U).  Entries are written into new slabs at offsets the host laid out from the operands' lengths. */
struct SlipRescaleArgs {
    const SlipEnt *ent; const uint64_t *limbs; const int32_t *idx; int64_t nz;     /* source entries; idx: row ids */
    const int64_t *colp; int32_t ncols;                                            /* L: column pointers (scale = column); U: null */
    const int32_t *pinv;                                                           /* U: scale = pinv[row]                          */
    const int32_t *slen; const int64_t *soff; const uint64_t *slimbs;              /* scales: signed digit counts, limb offsets      */
    SlipEnt *oent; uint64_t *olimbs;                                               /* destination (oent[e].off set by the host)     */
    int64_t *pividx; const int32_t *row_perm;                                      /* L: which entry of column k is the pivot         */
};

template <int D> SLIP_DEV void slip_rescale_entry_reg(const dig_t *a, int la, const dig_t *b, int lb, dig_t *out, int W, int *len_out, uint32_t *top_out)
{
    WR<D> A = wr_load<D>(a, la), B = wr_load<D>(b, lb);
    WR<D> Y = la <= lb ? wr_mul<D>(A, la, B) : wr_mul<D>(B, lb, A);
    const int len = wr_len<D>(Y);
    wr_store<D>(out, Y, (len + 1) & ~1);
    (void) W;
    *len_out = len; *top_out = len ? wr_digit<D>(Y, len - 1) : 0u;
}

#ifndef SLIP_EMULATE
__global__ void __launch_bounds__(256)
slip_rescale_kernel(SlipRescaleArgs A)
#else
static void slip_rescale_body(SlipRescaleArgs A)
#endif
{
    const int lane = slip_lane();
    const int64_t wave0 = (int64_t) slip_block() * slip_nwaves() + slip_wave(), nwaves = (int64_t) slip_nblocks() * slip_nwaves();
    for (int64_t e = wave0; e < A.nz; e += nwaves) {
        const SlipEnt en = A.ent[e];
        const int row = A.idx[e];
        int sidx;
        if (A.colp) {                       /* column of entry e: binary search over the column pointers */
            int lo = 0, hi = A.ncols - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (A.colp[mid] <= e) lo = mid; else hi = mid - 1; }
            sidx = lo;
        } else sidx = A.pinv[row];
        const int32_t sl = A.slen[sidx];
        const dig_t *a = (const dig_t *)(A.limbs + en.off), *b = (const dig_t *)(A.slimbs + A.soff[sidx]);
        const int la = slip_abs(en.len), lb = slip_abs(sl);
        SlipEnt o = A.oent[e];
        dig_t *out = (dig_t *)(A.olimbs + o.off);
        int len = 0; uint32_t top = 0;
        if (la == 0) { len = 0; }
        else {
            const int W = la + lb;
            if (W <= 64) slip_rescale_entry_reg<1>(a, la, b, lb, out, W, &len, &top);
            else if (W <= 128) slip_rescale_entry_reg<2>(a, la, b, lb, out, W, &len, &top);
            else if (W <= 192) slip_rescale_entry_reg<3>(a, la, b, lb, out, W, &len, &top);
            else if (W <= 256) slip_rescale_entry_reg<4>(a, la, b, lb, out, W, &len, &top);
            else {
                const int Wp = (W + 1) & ~1;
                wb_mul_lo(out, a, la, b, lb, Wp);
                len = wb_len(out, Wp);
                top = len ? out[len - 1] : 0u;
            }
        }
        if (lane == 0) {
            o.len = (slip_sgn(en.len) * slip_sgn(sl)) < 0 ? -len : len;
            o.bits = len ? 32 * len - slip_clz32(top) : 0;
            A.oent[e] = o;
            if (A.colp && A.row_perm[sidx] == row) A.pividx[sidx] = e;
        }
    }
}

#ifndef SLIP_EMULATE
#define SLIP_MAX_WAVES 8                   /* at most 512 threads per worker: 256 VGPRs per lane, two waves per SIMD */
template <bool FAST>
__global__ void __launch_bounds__(64 * SLIP_MAX_WAVES)
slip_factor_kernel(SlipParams P, SlipState *st)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t slip_lds[];
    /* the parameters live in LDS for the whole launch (held in SGPRs across the column loop they made the
     * compiler spill scalars into VGPR lanes all over the lane-level phases); the private arrays are this worker's */
    __shared__ SlipParams sP;
    if (threadIdx.x == 0) slip_worker_params(&sP, P, (int) blockIdx.x);
    __syncthreads();
    /* where this workgroup runs: HW_ID (cu_id bits 11:8, sh_id 12, se_id 15:13) and the XCC id.  The instruction cache is
     * shared by neighbouring CUs: a column worker next to the committer evicts the committer's short loop all the time
     * (the worker's code is far larger than the cache), so the workers on the committer's neighbours stand aside
     * (P.quiet_neighbours; placement is read, never assumed: a worker that does not see the committer just works) */
    const uint32_t hwid = (uint32_t) __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = (uint32_t) __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
    const uint32_t where = ((hwid >> 8) & 0xFFu) | (xcc << 8);          /* cu | sh | se | xcc */
    if (threadIdx.x == 0 && blockIdx.x < 2048) P.dbg[24 * (int64_t) P.n + 2048 + blockIdx.x] = (int32_t)(where | 0x10000u);
    if (sP.committer && blockIdx.x == 0) {
        if (threadIdx.x == 0) slip_st_i32(&st->committer_where, (int32_t)(where | 0x10000u));
        slip_committer<FAST>(sP, st, slip_lds);
    } else {
        int stand_aside = 0;
        if (sP.committer && sP.quiet_neighbours) {
            __shared__ int aside_;
            if (threadIdx.x == 0) {
                int cw = 0;
                for (int spin = 0; spin < 4000 && !(cw = slip_ld_i32(&st->committer_where)); spin++) slip_sleep();
                aside_ = 0;
                if (cw) {
                    const uint32_t c = (uint32_t) cw;
                    const int same_array = ((c >> 4) & 0xFFu) == ((where >> 4) & 0xFFu);       /* sh, se, xcc */
                    const int dcu = (int)(c & 0xFu) - (int)(where & 0xFu);
                    aside_ = same_array && dcu >= -sP.quiet_neighbours && dcu <= sP.quiet_neighbours;
                }
            }
            __syncthreads();
            stand_aside = aside_;
        }
        if (!stand_aside) slip_factor_worker<FAST>(sP, st, slip_lds);
    }
    slip_worker_exit(sP, st);
}

template <bool FAST>
__global__ void __launch_bounds__(64 * SLIP_MAX_WAVES)
slip_solve_kernel(SlipParams P, SlipState *st, SlipSolveArgs A, int32_t *rhs_done)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t slip_lds[];
    __shared__ SlipParams sP;
    if (threadIdx.x == 0) slip_worker_params(&sP, P, (int) blockIdx.x);
    __syncthreads();
    slip_solve_worker<FAST>(sP, st, A, rhs_done, slip_lds);
}

/* unit-test kernel: block b performs operation b with one wavefront */
extern "C" __global__ void __launch_bounds__(64)
slip_wave_op_kernel(int op, int la, int lb, int W, const uint32_t *a, const uint32_t *b,
                    uint32_t *out, uint32_t *scratch)
{
    const int blk = slip_block();
    const uint32_t *A = a + (int64_t) blk * la, *B = b + (int64_t) blk * (lb > 0 ? lb : 1);
    uint32_t *O = out + (int64_t) blk * W, *s0 = scratch + (int64_t) blk * 2 * (W + 1), *s1 = s0 + W + 1;
    if (op == 0) wb_mul_lo(O, A, la, B, lb, W);
    else if (op == 1) wb_addsub(O, A, la, B, lb, W, 0);
    else if (op == 2) wb_addsub(O, A, la, B, lb, W, 1);
    else if (op == 3) wb_inv_extend(O, 0, W, A, la, s0, s1);
    else slip_reg_op_test(op, A, la, B, lb, W, O, s0);
}

/* micro-benchmark kernel (development aid): cycles per wave-level primitive, nwaves waves busy */
extern "C" __global__ void __launch_bounds__(512)
slip_wave_bench_kernel(int op, int la, int lb, int W, int iters, int out_in_lds,
                       const uint32_t *a, const uint32_t *b, uint32_t *gout, unsigned long long *cycles)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t slip_lds[];
    const int wave = slip_wave();
    uint32_t *O = out_in_lds ? slip_lds + wave * 2 * (W + 2) : gout + (int64_t) wave * 2 * (W + 2);
    uint32_t *O2 = O + W + 2;
    const uint32_t *A = a + (int64_t) wave * la, *B = b + (int64_t) wave * lb;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        if (op == 0) wb_mul_lo(O, A, la, B, lb, W);
        else if (op == 1) wb_addsub(O, A, la, B, lb, W, 1);
        else if (op == 2) { int l = wb_len(B, lb); if (l == 12345) O[0] = 1; }
        else if (op == 3) wb_copy_shr(O, B, lb, 3, W);
        else if (op == 4) { wb_mul_lo(O, A, la, B, lb, W); wb_mul_lo(O2, O, W, A, la, W); }
        else if (op == 5) slip_wave_sync();
        else if (op == 6) {                    /* register product, operands already loaded */
            const int ea = la < W ? la : W;
            if (W <= 64) { WR<1> x = wr_load<1>(A, ea), y = wr_load<1>(B, lb < W ? lb : W); wr_store<1>(O, wr_mul<1>(x, ea, y), W); }
            else if (W <= 128) { WR<2> x = wr_load<2>(A, ea), y = wr_load<2>(B, lb < W ? lb : W); wr_store<2>(O, wr_mul<2>(x, ea, y), W); }
            else if (W <= 192) { WR<3> x = wr_load<3>(A, ea), y = wr_load<3>(B, lb < W ? lb : W); wr_store<3>(O, wr_mul<3>(x, ea, y), W); }
            else { WR<4> x = wr_load<4>(A, ea), y = wr_load<4>(B, lb < W ? lb : W); wr_store<4>(O, wr_mul<4>(x, ea, y), W); }
        }
        else if (op == 7 || op == 8) {         /* exact division of a W-digit number: 7 Hensel, 8 Newton inverse to W digits + product */
            const int ea = la < W ? la : W, eb = lb < W ? lb : W;
#define SLIP_BENCH_DIV(DD) do { WR<DD> x = wr_load<DD>(A, ea), y = wr_load<DD>(B, eb); if (slip_lane() == 0) y.d[0] |= 1u; \
                if (op == 7) wr_store<DD>(O, wr_div_hensel<DD>(x, W, y), W); \
                else { WR<DD> v = wr_inv_extend<DD>(wr_zero<DD>(), 0, W, y); wr_store<DD>(O, wr_mask<DD>(wr_mul<DD>(v, W, x), W), W); } } while (0)
            if (W <= 64) SLIP_BENCH_DIV(1); else if (W <= 128) SLIP_BENCH_DIV(2); else if (W <= 192) SLIP_BENCH_DIV(3); else SLIP_BENCH_DIV(4);
#undef SLIP_BENCH_DIV
        }
    }
    unsigned long long t1 = clock64();
    if (slip_lane() == 0) cycles[wave] = (t1 - t0) / (unsigned long long) iters;
}
#else
#define SLIP_MAX_WAVES 16                  /* emulation build */
#endif

/* ------------------------------------------------------------------ */
/* host state                                                          */
/* ------------------------------------------------------------------ */
struct slip_hip_factor {
    SlipParams P;         /* kernel arguments (device pointers inside; private arrays: base of worker 0) */
    SlipState hs;         /* host mirror of the mutable device state     */
    SlipState *ds;        /* device copy the kernel works on             */
    int32_t n; int64_t annz, alimbs;
    int32_t waves, lds_words, bitmap_in_lds, scratch_in_lds;
    int32_t nworkers;      /* column workers = workgroups of a launch; private arrays are sized for this many */
    int32_t workers_asked; /* slip_hip_options.workers (0: as many as can be resident) */
    int32_t no_committer;  /* diagnostics: every column is committed by its own worker */
    int32_t no_farm;       /* diagnostics: long update queues are not opened to other workers */
    int32_t no_engine;     /* diagnostics: no full packages (the committer's chain engine stays off) */
    int32_t engine_ok;     /* the committer's LDS layout with the pinv mirror fits */
    int32_t last_status, window_end, launches;
    int32_t factors_only;  /* built from given factors (slip_hip_factor_from_factors): solve only, no A */
    double kernel_ms, solve_ms;
    hipEvent_t ev0, ev1;
    int32_t *ident;        /* device: 0..n-1 (reset copies it into pinv / row_perm) */
    /* subtree farm: the rescaled copy of the committed factors (slip_hip_factor_rescale); download serves it while it exists */
    SlipEnt *rsLe, *rsUe; uint64_t *rsLl, *rsUl; int64_t rsLnl, rsUnl, rsLexact, rsUexact; int64_t *rspiv; int32_t rescaled;
    /* owned device arrays that are only reachable through const pointers in P */
    int64_t *dAp; int32_t *dAi, *dAlen; int64_t *dAoff; uint64_t *dAlimbs; int32_t *dq;
};

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        fprintf(stderr, "slip_hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        return SLIP_HIP_DEVICE_ERROR; } } while (0)

static void rescale_drop(slip_hip_factor *f);

/* Device buffers go through a process-level pool: the drop-in SLIP_LU_factorize creates and destroys a
 * handle per call, and a handle's private row arrays are gigabytes -- hipMalloc / hipFree of those cost milliseconds each.
 * A freed buffer of at least 1 MiB is kept (up to SLIP_HIP_POOL_MB megabytes in all, default 32 768; 0 switches the pool off)
 * and handed to the next request it fits within a quarter of its size.  Nothing relies on fresh memory being zero: every
 * array that must start cleared is cleared where it is allocated or at reset.  Handles are destroyed after their streams
 * have been synchronised (run / solve return synchronised), so a pooled buffer is never still in use. */
#include <mutex>
#include <unordered_map>
struct SlipPoolEnt { void *p; size_t bytes; };
static std::mutex slip_pool_mu;
static std::vector<SlipPoolEnt> slip_pool;
static std::unordered_map<void *, size_t> slip_pool_live;
static size_t slip_pool_bytes = 0;
static size_t slip_pool_limit(void)
{
    static long long mb = -1;
    if (mb < 0) { const char *e = getenv("SLIP_HIP_POOL_MB"); mb = e ? atoll(e) : 32768; if (mb < 0) mb = 0; }
    return (size_t) mb << 20;
}
static int dev_malloc_bytes(void **q, size_t bytes)
{
    {
        std::lock_guard<std::mutex> g(slip_pool_mu);
        size_t best = (size_t) -1, at = 0;
        for (size_t t = 0; t < slip_pool.size(); t++)
            if (slip_pool[t].bytes >= bytes && slip_pool[t].bytes - bytes <= bytes / 4 && slip_pool[t].bytes < best) { best = slip_pool[t].bytes; at = t; }
        if (best != (size_t) -1) {
            *q = slip_pool[at].p; slip_pool_live[*q] = best; slip_pool_bytes -= best;
            slip_pool[at] = slip_pool.back(); slip_pool.pop_back();
            return 0;
        }
    }
    if (hipMalloc(q, bytes) != hipSuccess) {
        /* the pool may be what is in the way: give it back and try once more */
        std::vector<SlipPoolEnt> drop;
        { std::lock_guard<std::mutex> g(slip_pool_mu); drop.swap(slip_pool); slip_pool_bytes = 0; }
        for (const SlipPoolEnt &e : drop) hipFree(e.p);
        if (drop.empty() || hipMalloc(q, bytes) != hipSuccess) return SLIP_HIP_OUT_OF_MEMORY;
    }
    std::lock_guard<std::mutex> g(slip_pool_mu);
    slip_pool_live[*q] = bytes;
    return 0;
}
static void dev_free(void *p)
{
    if (!p) return;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> g(slip_pool_mu);
        auto it = slip_pool_live.find(p);
        if (it != slip_pool_live.end()) { bytes = it->second; slip_pool_live.erase(it); }
        if (bytes >= ((size_t) 1 << 20) && slip_pool_bytes + bytes <= slip_pool_limit()) {
            slip_pool.push_back(SlipPoolEnt{p, bytes}); slip_pool_bytes += bytes;
            return;
        }
    }
    hipFree(p);
}
/* give every pooled buffer back to the runtime (tests; a host that wants the memory) */
extern "C" void slip_hip_pool_release(void)
{
    std::vector<SlipPoolEnt> drop;
    { std::lock_guard<std::mutex> g(slip_pool_mu); drop.swap(slip_pool); slip_pool_bytes = 0; }
    for (const SlipPoolEnt &e : drop) hipFree(e.p);
}

template <class T> static int dev_alloc(T **p, int64_t count)
{
    void *q = NULL;
    if (dev_malloc_bytes(&q, (size_t)(count > 0 ? count : 1) * sizeof(T))) return SLIP_HIP_OUT_OF_MEMORY;
    *p = (T *) q;
    return 0;
}
template <class T> static int dev_grow(T **p, int64_t old_count, int64_t new_count)
{
    T *q = NULL;
    if (dev_alloc(&q, new_count)) return SLIP_HIP_OUT_OF_MEMORY;
    if (old_count > 0 && hipMemcpy(q, *p, (size_t) old_count * sizeof(T), hipMemcpyDeviceToDevice) != hipSuccess) { dev_free(q); return SLIP_HIP_DEVICE_ERROR; }
    dev_free(*p);
    *p = q;
    return 0;
}

extern "C" const char *slip_hip_version(void) { return "slip_hip 0.2 (gfx950, column-worker pipeline)"; }

extern "C" void slip_hip_default_options(slip_hip_options *o)
{
    /* SLIP_LU_internal.h:136-149: pivot = SLIP_TOL_SMALLEST, tol = 1 */
    o->pivot = 3; o->tol = 1.0; o->limb_cap = 0; o->waves = 0; o->lnz_hint = 0; o->unz_hint = 0;
    o->workers = 0; o->reserved = 0;
}

extern "C" int slip_hip_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

extern "C" void slip_hip_free(void *p) { free(p); }

/* ---- triplet files into limb slabs and back, host only, no GMP ----
 * The reference reads "m n nz" and nz lines "i j value" with gmp_fscanf %Zd (SLIP_LU/Demo/demos.c:245-331: indices are
 * 1-based unless the FIRST entry holds a 0) and builds the CSC by a counting sort on the columns that keeps the file
 * order inside a column and keeps duplicates (SLIP_LU/Source/slip_trip_to_mat.c:23-69).  Here the decimal strings go
 * straight into 64-bit limbs (base-10^18 chunks multiplied in), one pass over the file, one allocation per array. */
static int trip_parse_value(const char *s, const char *e, std::vector<uint64_t> &mag, int *neg)
{
    *neg = 0; mag.clear();
    if (s < e && (*s == '-' || *s == '+')) { *neg = *s == '-'; s++; }
    if (s >= e) return -1;
    for (const char *p = s; p < e; p++) if (*p < '0' || *p > '9') return -1;
    while (s < e) {
        const int take = (int)((e - s) % 18 ? (e - s) % 18 : 18);
        uint64_t chunk = 0, scale = 1;
        for (int q = 0; q < take; q++) { chunk = chunk * 10u + (uint64_t)(s[q] - '0'); scale *= 10u; }
        s += take;
        /* mag = mag * scale + chunk */
        unsigned __int128 carry = chunk;
        for (size_t l = 0; l < mag.size(); l++) { carry += (unsigned __int128) mag[l] * scale; mag[l] = (uint64_t) carry; carry >>= 64; }
        if (carry) mag.push_back((uint64_t) carry);
    }
    while (!mag.empty() && mag.back() == 0) mag.pop_back();
    if (mag.empty()) *neg = 0;
    return 0;
}

extern "C" int slip_hip_read_triplet(const char *path, int32_t *n_out, int64_t **Ap_out, int32_t **Ai_out, int32_t **Alen_out,
                                     uint64_t **Alimbs_out, int64_t *nlimbs_out)
{
    if (!path || !n_out || !Ap_out || !Ai_out || !Alen_out || !Alimbs_out) return SLIP_HIP_INCORRECT_INPUT;
    FILE *fp = fopen(path, "rb");
    if (!fp) return SLIP_HIP_INCORRECT_INPUT;
    std::vector<char> buf;
    { char tmp[1 << 16]; size_t r; while ((r = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + r); }
    fclose(fp);
    buf.push_back('\0');
    const char *p = buf.data(), *end = buf.data() + buf.size() - 1;
    auto token = [&](const char **ts, const char **te) -> int {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
        if (p >= end) return -1;
        *ts = p;
        while (p < end && !(*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
        *te = p;
        return 0;
    };
    auto int_token = [&](long long *v) -> int {
        const char *ts, *te;
        if (token(&ts, &te)) return -1;
        char *ep = NULL; const long long x = strtoll(ts, &ep, 10);
        if (ep != te) return -1;
        *v = x; return 0;
    };
    long long m, n, nz;
    if (int_token(&m) || int_token(&n) || int_token(&nz)) return SLIP_HIP_INCORRECT_INPUT;
    if (n <= 0 || m != n || nz <= 0 || n > INT32_MAX || nz > INT32_MAX) return SLIP_HIP_INCORRECT_INPUT;
    std::vector<int32_t> I((size_t) nz), J((size_t) nz), L((size_t) nz);
    std::vector<int64_t> off((size_t) nz + 1);
    std::vector<uint64_t> limbs, mag;
    limbs.reserve((size_t) nz);
    long long dec = 0;
    for (long long t = 0; t < nz; t++) {
        long long i, j; const char *ts, *te; int neg;
        if (int_token(&i) || int_token(&j) || token(&ts, &te) || trip_parse_value(ts, te, mag, &neg)) return SLIP_HIP_INCORRECT_INPUT;
        if (t == 0) dec = (i < j ? i : j) == 0 ? 0 : 1;           /* demos.c:290-299 */
        i -= dec; j -= dec;
        if (i < 0 || j < 0 || i >= n || j >= n) return SLIP_HIP_INCORRECT_INPUT;
        I[(size_t) t] = (int32_t) i; J[(size_t) t] = (int32_t) j;
        off[(size_t) t] = (int64_t) limbs.size();
        L[(size_t) t] = neg ? -(int32_t) mag.size() : (int32_t) mag.size();
        limbs.insert(limbs.end(), mag.begin(), mag.end());
    }
    off[(size_t) nz] = (int64_t) limbs.size();
    /* counting sort on the columns, file order kept inside a column (slip_trip_to_mat.c:47-64) */
    int64_t *Ap = (int64_t *) calloc((size_t) n + 1, sizeof(int64_t));
    int32_t *Ai = (int32_t *) malloc((size_t) nz * sizeof(int32_t)), *Alen = (int32_t *) malloc((size_t) nz * sizeof(int32_t));
    uint64_t *Al = (uint64_t *) malloc((limbs.size() ? limbs.size() : 1) * sizeof(uint64_t));
    if (!Ap || !Ai || !Alen || !Al) { free(Ap); free(Ai); free(Alen); free(Al); return SLIP_HIP_OUT_OF_MEMORY; }
    for (long long t = 0; t < nz; t++) Ap[J[(size_t) t] + 1]++;
    for (long long c = 0; c < n; c++) Ap[c + 1] += Ap[c];
    std::vector<int64_t> w(Ap, Ap + n), src((size_t) nz);
    for (long long t = 0; t < nz; t++) { const int64_t q = w[(size_t) J[(size_t) t]]++; Ai[q] = I[(size_t) t]; Alen[q] = L[(size_t) t]; src[(size_t) q] = t; }
    int64_t o = 0;
    for (long long q = 0; q < nz; q++) {
        const int64_t t = src[(size_t) q], l = off[(size_t) t + 1] - off[(size_t) t];
        if (l) memcpy(Al + o, limbs.data() + off[(size_t) t], (size_t) l * 8);
        o += l;
    }
    *n_out = (int32_t) n; *Ap_out = Ap; *Ai_out = Ai; *Alen_out = Alen; *Alimbs_out = Al;
    if (nlimbs_out) *nlimbs_out = o;
    return SLIP_HIP_OK;
}

/* the inverse: CSC limb slabs to a triplet file the reference's SLIP_tripread accepts (1-based, decimal) */
extern "C" int slip_hip_write_triplet(const char *path, int32_t n, const int64_t *Ap, const int32_t *Ai, const int32_t *Alen, const uint64_t *Alimbs)
{
    if (!path || n <= 0 || !Ap || !Ai || !Alen || !Alimbs) return SLIP_HIP_INCORRECT_INPUT;
    FILE *fp = fopen(path, "wb");
    if (!fp) return SLIP_HIP_INCORRECT_INPUT;
    fprintf(fp, "%d %d %lld\n", n, n, (long long) Ap[n]);
    int64_t o = 0;
    std::vector<uint64_t> mag; std::vector<uint64_t> chunks;
    for (int32_t c = 0; c < n; c++)
        for (int64_t q = Ap[c]; q < Ap[c + 1]; q++) {
            const int l = Alen[q] < 0 ? -Alen[q] : Alen[q];
            mag.assign(Alimbs + o, Alimbs + o + l); o += l;
            chunks.clear();
            while (!mag.empty()) {                                   /* divide by 10^18, collect the remainders */
                unsigned __int128 rem = 0;
                for (size_t k = mag.size(); k-- > 0;) { rem = (rem << 64) | mag[k]; mag[k] = (uint64_t)(rem / 1000000000000000000ull); rem %= 1000000000000000000ull; }
                chunks.push_back((uint64_t) rem);
                while (!mag.empty() && mag.back() == 0) mag.pop_back();
            }
            fprintf(fp, "%d %d %s", Ai[q] + 1, c + 1, Alen[q] < 0 ? "-" : "");
            if (chunks.empty()) fprintf(fp, "0");
            else {
                fprintf(fp, "%llu", (unsigned long long) chunks.back());
                for (size_t k = chunks.size() - 1; k-- > 0;) fprintf(fp, "%018llu", (unsigned long long) chunks[k]);
            }
            fprintf(fp, "\n");
        }
    return fclose(fp) == 0 ? SLIP_HIP_OK : SLIP_HIP_DEVICE_ERROR;
}

extern "C" int slip_hip_matgen(int32_t n, double density, int32_t bits, uint64_t seed,
                               int64_t **Ap, int32_t **Ai, int64_t **Ax)
{
    return slip_matgen_csc(n, density, bits, seed, Ap, Ai, Ax) ? SLIP_HIP_OUT_OF_MEMORY : SLIP_HIP_OK;
}

/* choose waves / LDS split for the current xcap */
static void plan_launch(slip_hip_factor *f)
{
    SlipParams *P = &f->P;
    P->wcap = P->xcap + 8;
    P->bm_words = (P->n + 31) / 32;
    f->bitmap_in_lds = P->bm_words <= SLIP_BITMAP_LDS_MAX_WORDS;
    int fixed = SLIP_LDS_BITMAP + (f->bitmap_in_lds ? P->bm_words : 0);
    int nw = f->waves;
    /* per wave 3 scratch buffers of wcap digits, plus one workgroup-shared buffer */
    while (nw > 4 && fixed + (int64_t)(nw * 3 + 1) * P->wcap > SLIP_LDS_MAX_WORDS) nw /= 2;
    f->scratch_in_lds = fixed + (int64_t)(nw * 3 + 1) * P->wcap <= SLIP_LDS_MAX_WORDS;
    f->waves = nw;
    f->lds_words = fixed + (f->scratch_in_lds ? (nw * 3 + 1) * P->wcap : 0);
    P->bitmap_in_lds = f->bitmap_in_lds; P->scratch_in_lds = f->scratch_in_lds;
    /* block 0 of a factorisation launch may be the committer (ref_lu_pipe_commit.h): its rings, package copies and -- for
     * matrices up to SLIP_MIRROR_MAX rows -- the chain engine's mirror of pinv live in the same dynamic LDS */
    f->engine_ok = P->n <= SLIP_MIRROR_MAX && !f->no_engine;
    { const int cw = slip_commit_lds_words(P->n, f->engine_ok); if (!f->factors_only && cw <= SLIP_LDS_MAX_WORDS && cw > f->lds_words) f->lds_words = cw; }
}

#ifdef SLIP_EMULATE
/* test hooks of the CPU emulation build (never in the product library) */
static int64_t slip_emu_budget = 96ll << 30;
extern "C" void slip_emu_set_budget(long long bytes) { slip_emu_budget = bytes > 0 ? bytes : (96ll << 30); }
#endif
/* how many column workers: as many as can be resident (LDS-limited workgroups per CU times the CUs), capped by the
 * columns there are and by what their private vectors may take of the HBM */
static int32_t default_workers(const slip_hip_factor *f, int32_t xcap, int32_t asked)
{
    int cus = 256;
#ifndef SLIP_EMULATE
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
#else
    cus = 3;
#endif
    const int64_t lds_bytes = (int64_t) f->lds_words * 4;
    int per_cu = (int)((160 * 1024) / (lds_bytes > 0 ? lds_bytes : 1));
    const int by_waves = 8 / (f->waves > 0 ? f->waves : 1);           /* 232 VGPRs: two waves per SIMD, eight per CU */
    if (per_cu > by_waves) per_cu = by_waves;
#ifndef SLIP_EMULATE
    /* ... and never more than the runtime says can be resident with this launch shape (the default grid is what fits the
     * chip at once; larger grids asked for explicitly still work -- a workgroup that is not resident holds no ticket) */
    {
        int api = 0;
        const int fast = f->bitmap_in_lds && f->scratch_in_lds;
        const void *fn = fast ? (const void *) slip_factor_kernel<true> : (const void *) slip_factor_kernel<false>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes) == hipSuccess &&
            (fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, slip_factor_kernel<true>, 64 * f->waves, (size_t) lds_bytes)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, slip_factor_kernel<false>, 64 * f->waves, (size_t) lds_bytes)) == hipSuccess &&
            api >= 1 && api < per_cu) per_cu = api;
    }
#endif
    if (per_cu < 1) per_cu = 1;
    int64_t w = asked > 0 ? asked : (int64_t) cus * per_cu;
    const int64_t per_worker = (int64_t) f->n * (16 + 4 * (int64_t) xcap + 16) + 4096;
#ifdef SLIP_EMULATE
    const int64_t budget = slip_emu_budget;                             /* tests shrink it to force the re-application at a wider stride */
#else
    const int64_t budget = 96ll << 30;                                  /* of the 288 GB */
#endif
    if (w * per_worker > budget) w = budget / per_worker;
    if (w > f->n) w = f->n;
    if (w < 1) w = 1;
    return (int32_t) w;
}

/* the workers' private arrays (x rows and their digits, pattern overflow, row lists, HBM bitmap / scratch) and the
 * shared inverse cache for stride xcap */
static int alloc_x(slip_hip_factor *f, int32_t xcap, int keep_rows)
{
    SlipParams *P = &f->P;
    xcap = (xcap + 3) & ~3;
    /* plan for the new stride on a copy: the handle keeps its working buffers until every new one exists */
    const int32_t old_xcap = P->xcap, old_invcap = P->invcap, old_waves = f->waves;
    P->xcap = xcap; P->invcap = xcap + 8;
    plan_launch(f);
    /* how many workers: what was asked for (or what can be resident), never more than the HBM budget allows at THIS stride,
     * and never more than the private row arrays were sized for */
    int32_t W32 = default_workers(f, xcap, f->workers_asked);
    if (keep_rows && f->nworkers > 0 && W32 > f->nworkers) W32 = f->nworkers;
    const int64_t W = W32, n = P->n;
    uint32_t *nxd = NULL, *ninvd = NULL, *ngs = NULL, *ngb = NULL;
    int rc = 0;
    if (dev_alloc(&nxd, W * n * xcap) || dev_alloc(&ninvd, n * (int64_t) P->invcap) ||
        dev_alloc(&ngs, f->scratch_in_lds ? 1 : W * SLIP_SCRATCH_WAVES * 3 * (int64_t) P->wcap) ||
        dev_alloc(&ngb, f->bitmap_in_lds ? 1 : W * ((int64_t) P->bm_words + 64))) rc = SLIP_HIP_OUT_OF_MEMORY;
    SlipRow *nxrow = NULL; int32_t *npat = NULL, *nrlist = NULL, *nrpos = NULL, *nsrow = NULL; uint32_t *npkg = NULL, *njobs = NULL;
    if (!rc && !keep_rows) {
        if (dev_alloc(&nxrow, W * n) || dev_alloc(&npat, W * n) || dev_alloc(&nrlist, W * n) ||
            dev_alloc(&nrpos, W * n) || dev_alloc(&nsrow, W * n)) rc = SLIP_HIP_OUT_OF_MEMORY;
        /* tags start at 0 = "belongs to no column"; tickets count from 1 */
        else if (hipMemset(nxrow, 0, (size_t)(W * n) * sizeof(SlipRow)) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    }
    if (!rc && (!keep_rows || !P->pkg.p_)) {
        if (dev_alloc(&npkg, W * (int64_t)(SLIP_PKG_WORDS + SLIP_MBOX_WORDS)) || dev_alloc(&njobs, W * (int64_t) SLIP_JOB_WORDS)) rc = SLIP_HIP_OUT_OF_MEMORY;
        else if (hipMemset(njobs, 0, (size_t) W * SLIP_JOB_WORDS * 4) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    }
    if (rc) {
        /* nothing of the handle has been touched: it stays usable at the old stride */
        dev_free(nxd); dev_free(ninvd); dev_free(ngs); dev_free(ngb); dev_free(nxrow); dev_free(npat); dev_free(nrlist); dev_free(nrpos); dev_free(nsrow); dev_free(npkg); dev_free(njobs);
        if (old_xcap > 0) { P->xcap = old_xcap; P->invcap = old_invcap; f->waves = old_waves; plan_launch(f); }
        return rc;
    }
    dev_free(P->xd); dev_free(P->invd.p_); dev_free(P->gscratch); dev_free(P->gbitmap);
    P->xd = nxd; P->invd.p_ = ninvd; P->gscratch = ngs; P->gbitmap = ngb;
    if (!keep_rows) {
        dev_free(P->xrow); dev_free(P->pat); dev_free(P->rlist); dev_free(P->rpos); dev_free(P->srow);
        P->xrow = nxrow; P->pat = npat; P->rlist = nrlist; P->rpos = nrpos; P->srow = nsrow;
    }
    if (npkg) { dev_free(P->pkg.p_); dev_free(P->jobs.p_); P->pkg.p_ = npkg; P->jobs.p_ = njobs; }
    f->nworkers = W32; P->nworkers = W32; P->priv_rows = n;
    return 0;
}

static int upload_state(slip_hip_factor *f, hipStream_t stream)
{
    CK(hipMemcpyAsync(f->ds, &f->hs, sizeof(SlipState), hipMemcpyHostToDevice, stream));
    return 0;
}

extern "C" int slip_hip_factor_reset(slip_hip_factor *f)
{
    if (!f || f->factors_only) return SLIP_HIP_INCORRECT_INPUT;
    SlipParams *P = &f->P;
    const int32_t n = f->n;
    if (!P->xd || !P->xrow) return SLIP_HIP_OUT_OF_MEMORY;
    if (f->rescaled) rescale_drop(f);
    /* tickets double as row tags and are never reused -- until they would wrap: then every private row is untagged and
     * the count starts over (a launch draws at most one ticket per column plus one per worker) */
    if (f->hs.ticket > (1 << 30)) {
        CK(hipMemsetAsync(P->xrow, 0, (size_t)((int64_t) f->nworkers * n) * sizeof(SlipRow), 0));
        f->hs.ticket = 0;
    }
    CK(hipMemcpyAsync(P->pinv.p_, f->ident, (size_t) n * 4, hipMemcpyDeviceToDevice, 0));
    CK(hipMemcpyAsync(P->row_perm.p_, f->ident, (size_t) n * 4, hipMemcpyDeviceToDevice, 0));
    CK(hipMemsetAsync(P->Lready.p_, 0, (size_t) n * 4, 0));
    CK(hipMemsetAsync(P->Lp, 0, 8, 0)); CK(hipMemsetAsync(P->Up, 0, 8, 0));
    CK(hipMemsetAsync(P->Lo, 0, 8, 0)); CK(hipMemsetAsync(P->Uo, 0, 8, 0));
    {
        const int32_t ticket = f->hs.ticket;           /* the tickets (= row tags) never go back */
        memset(&f->hs, 0, sizeof f->hs);
        f->hs.ticket = ticket;
        f->hs.stop = INT64_MAX;
    }
    { const int e = upload_state(f, 0); if (e) return e; }
    CK(hipStreamSynchronize(0));
    f->last_status = 0; f->window_end = 0; f->kernel_ms = 0; f->launches = 0;
    return SLIP_HIP_OK;
}

extern "C" void slip_hip_factor_destroy(slip_hip_factor *f)
{
    if (!f) return;
    SlipParams *P = &f->P;
    dev_free(f->dAp); dev_free(f->dAi); dev_free(f->dAlen); dev_free(f->dAoff); dev_free(f->dAlimbs); dev_free(f->dq);
    dev_free(P->pinv.p_); dev_free(P->row_perm.p_); dev_free(P->xrow); dev_free(P->xd);
    dev_free(P->piv.p_); dev_free(P->invd.p_);
    dev_free(P->Lp); dev_free(P->Lo); dev_free(P->Li); dev_free(P->Le); dev_free(P->Llimbs);
    dev_free(P->Up); dev_free(P->Uo); dev_free(P->Ui); dev_free(P->Ue); dev_free(P->Ulimbs);
    dev_free(P->Lready.p_); dev_free(P->pat); dev_free(P->rlist); dev_free(P->rpos); dev_free(P->srow); dev_free(P->gscratch); dev_free(P->gbitmap); dev_free(P->dbg); dev_free(P->pkg.p_); dev_free(P->jobs.p_); dev_free(P->sw_row.p_); dev_free(P->sw_pos.p_);
    dev_free(f->ds); dev_free(f->ident);
    rescale_drop(f);
    if (f->ev0) hipEventDestroy(f->ev0);
    if (f->ev1) hipEventDestroy(f->ev1);
    free(f);
}

/* options -> the handle's launch shape */
static void apply_options(slip_hip_factor *f, const slip_hip_options &opt)
{
    f->waves = opt.waves > 0 ? opt.waves : 8;
    if (f->waves > SLIP_MAX_WAVES) f->waves = SLIP_MAX_WAVES;
    f->workers_asked = opt.workers > 0 ? (opt.workers > 4096 ? 4096 : opt.workers) : 0;
    f->nworkers = 0;       /* chosen in alloc_x once the LDS need is known */
    f->P.no_early = opt.reserved & 1;
    f->no_committer = (opt.reserved >> 1) & 1;
    f->no_farm = (opt.reserved >> 2) & 1;
    f->no_engine = (opt.reserved >> 3) & 1;
    f->P.quiet_neighbours = (opt.reserved >> 4) & 3;      /* experiment: workers within this many CUs of the committer stand aside */
}

static int make_ident(slip_hip_factor *f)
{
    const int32_t n = f->n;
    int32_t *id = (int32_t *) malloc((size_t) n * 4);
    if (!id) return SLIP_HIP_OUT_OF_MEMORY;
    for (int32_t i = 0; i < n; i++) id[i] = i;
    int rc = dev_alloc(&f->ident, n);
    if (!rc && hipMemcpy(f->ident, id, (size_t) n * 4, hipMemcpyHostToDevice) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    free(id);
    return rc;
}

extern "C" int slip_hip_factor_create(slip_hip_factor **out, int32_t n,
                                      const int64_t *Ap, const int32_t *Ai,
                                      const int32_t *Alen, const uint64_t *Alimbs,
                                      const int32_t *q, const slip_hip_options *opt_in)
{
    /* SLIP_LU_factorize.c:48-52: any missing argument is SLIP_INCORRECT_INPUT */
    if (!out || n <= 0 || !Ap || !Ai || !Alen || !Alimbs || !q) return SLIP_HIP_INCORRECT_INPUT;
    *out = NULL;
    if (slip_hip_device_count() <= 0) {
        fprintf(stderr, "slip_hip: no HIP device available -- this library has no CPU fallback\n");
        return SLIP_HIP_DEVICE_ERROR;
    }
    slip_hip_options opt;
    if (opt_in) opt = *opt_in; else slip_hip_default_options(&opt);
    if (opt.pivot < 0 || opt.pivot > 5) return SLIP_HIP_INCORRECT_INPUT;
    /* column and row numbers travel in 24-bit fields of the commit protocol (verdict words, package records) */
    if (n >= (1 << 24) - 1) return SLIP_HIP_INCORRECT_INPUT;
    const int64_t annz = Ap[n];
    if (Ap[0] != 0 || annz < 1) return SLIP_HIP_INCORRECT_INPUT;

    /* ---- host-side preparation of A: bounds check, de-duplicate, digit counts ---- */
    int64_t *hAp = (int64_t *) malloc(((size_t) n + 1) * 8);
    int32_t *hAi = (int32_t *) malloc((size_t) annz * 4), *hAlen = (int32_t *) malloc((size_t) annz * 4);
    int64_t *hAoff = (int64_t *) malloc((size_t) annz * 8), *inoff = (int64_t *) malloc(((size_t) annz + 1) * 8);
    int32_t *last = (int32_t *) malloc((size_t) n * 4);
    char *seen = (char *) calloc((size_t) n, 1);
    if (!hAp || !hAi || !hAlen || !hAoff || !inoff || !last || !seen) {
        free(hAp); free(hAi); free(hAlen); free(hAoff); free(inoff); free(last); free(seen);
        return SLIP_HIP_OUT_OF_MEMORY;
    }
    int bad = 0;
    inoff[0] = 0;
    for (int64_t p = 0; p < annz; p++) inoff[p + 1] = inoff[p] + (Alen[p] < 0 ? -(int64_t) Alen[p] : Alen[p]);
    for (int32_t j = 0; j < n && !bad; j++) { if (q[j] < 0 || q[j] >= n || seen[q[j]]) bad = 1; else seen[q[j]] = 1; }
    for (int32_t i = 0; i < n; i++) last[i] = -1;
    uint64_t *hAlimbs = (uint64_t *) malloc((size_t)(inoff[annz] > 0 ? inoff[annz] : 1) * 8);
    if (!hAlimbs) bad = 2;
    int64_t onz = 0, ol = 0;
    int32_t maxdig = 1;
    for (int32_t j = 0; j < n && !bad; j++) {
        if (Ap[j + 1] < Ap[j]) { bad = 1; break; }
        hAp[j] = onz;
        /* a repeated row keeps its LAST value (slip_get_column.c:22 overwrites) */
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int32_t r = Ai[p];
            if (r < 0 || r >= n) { bad = 1; break; }
            last[r] = (int32_t)(p - Ap[j]);
        }
        if (bad) break;
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) {
            int32_t r = Ai[p];
            if (last[r] != (int32_t)(p - Ap[j])) continue;
            int64_t l = inoff[p + 1] - inoff[p];
            const uint64_t *src = Alimbs + inoff[p];
            while (l > 0 && src[l - 1] == 0) l--;
            int32_t dig = (int32_t)(2 * l);
            if (l > 0 && (src[l - 1] >> 32) == 0) dig--;
            hAi[onz] = r; hAlen[onz] = Alen[p] < 0 ? -dig : dig; hAoff[onz] = ol;
            memcpy(hAlimbs + ol, src, (size_t) l * 8);
            if (dig > maxdig) maxdig = dig;
            ol += l; onz++;
        }
        for (int64_t p = Ap[j]; p < Ap[j + 1]; p++) last[Ai[p]] = -1;
    }
    hAp[n] = onz;
    free(inoff); free(last); free(seen);
    if (bad) { free(hAp); free(hAi); free(hAlen); free(hAoff); free(hAlimbs); return bad == 2 ? SLIP_HIP_OUT_OF_MEMORY : SLIP_HIP_INCORRECT_INPUT; }

    slip_hip_factor *f = (slip_hip_factor *) calloc(1, sizeof(slip_hip_factor));
    if (!f) { free(hAp); free(hAi); free(hAlen); free(hAoff); free(hAlimbs); return SLIP_HIP_OUT_OF_MEMORY; }
    SlipParams *P = &f->P;
    f->n = n; f->annz = onz; f->alimbs = ol;
    apply_options(f, opt);
    P->n = n; P->pivot_scheme = opt.pivot; P->limb_cap = opt.limb_cap;
    if (!(opt.tol > 0)) { P->tol_mode = 0; P->tol_m = 0; P->tol_e = 0; }
    else {
        int e; double fr = frexp(opt.tol, &e);            /* mpq_set_d takes the double exactly */
        P->tol_mode = 1; P->tol_m = (uint64_t) ldexp(fr, 53); P->tol_e = e - 53;
    }
    int rc = 0;
#define A_(call) do { if (!rc) rc = (call); } while (0)
    A_(dev_alloc(&f->dAp, (int64_t) n + 1)); A_(dev_alloc(&f->dAi, onz)); A_(dev_alloc(&f->dAlen, onz));
    A_(dev_alloc(&f->dAoff, onz)); A_(dev_alloc(&f->dAlimbs, ol)); A_(dev_alloc(&f->dq, n));
    A_(dev_alloc(&P->pinv.p_, n)); A_(dev_alloc(&P->row_perm.p_, n));
    A_(dev_alloc(&P->piv.p_, n)); A_(dev_alloc(&P->Lready.p_, n)); A_(dev_alloc(&P->dbg, 24 * (int64_t) n + 4096)); A_(dev_alloc(&P->sw_row.p_, n)); A_(dev_alloc(&P->sw_pos.p_, n));
    A_(make_ident(f));
    if (!rc && hipMemset(P->dbg, 0, ((size_t) n * 24 + 4096) * 4) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc && hipMemset(P->piv.p_, 0, (size_t) n * sizeof(SlipPiv)) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    /* initial sizes: S->lnz/unz only size the first allocation in the reference too */
    P->Lcap_nz = opt.lnz_hint > 0 ? opt.lnz_hint : 4 * onz + n;
    P->Ucap_nz = opt.unz_hint > 0 ? opt.unz_hint : 4 * onz + n;
    if (P->Lcap_nz < n) P->Lcap_nz += n;
    if (P->Ucap_nz < n) P->Ucap_nz += n;
    /* window mode: the cap, plus room for the working width of a division item on a value at the cap (result + shifted-out zeros of rho[h] + 1) */
    const int32_t cap_digits = opt.limb_cap > 0 ? 2 * opt.limb_cap + 40 : 0;
    int32_t xcap0 = cap_digits > 0 ? cap_digits : (2 * maxdig + 8 > 16 ? 2 * maxdig + 8 : 16);
    P->Lcap_nl = P->Lcap_nz * 2;
    P->Ucap_nl = P->Ucap_nz * 2;
    A_(dev_alloc(&P->Lp, (int64_t) n + 1)); A_(dev_alloc(&P->Lo, (int64_t) n + 1)); A_(dev_alloc(&P->Li, P->Lcap_nz)); A_(dev_alloc(&P->Le, P->Lcap_nz));
    A_(dev_alloc(&P->Llimbs, P->Lcap_nl));
    A_(dev_alloc(&P->Up, (int64_t) n + 1)); A_(dev_alloc(&P->Uo, (int64_t) n + 1)); A_(dev_alloc(&P->Ui, P->Ucap_nz)); A_(dev_alloc(&P->Ue, P->Ucap_nz));
    A_(dev_alloc(&P->Ulimbs, P->Ucap_nl));
    A_(dev_alloc(&f->ds, 1));
    if (!rc) rc = alloc_x(f, xcap0, 0);
#undef A_
    if (!rc) {
        if (hipMemcpy(f->dAp, hAp, ((size_t) n + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAi, hAi, (size_t) onz * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAlen, hAlen, (size_t) onz * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAoff, hAoff, (size_t) onz * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAlimbs, hAlimbs, (size_t) ol * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dq, q, (size_t) n * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = SLIP_HIP_DEVICE_ERROR;
    }
    free(hAp); free(hAi); free(hAlen); free(hAoff); free(hAlimbs);
    P->Ap = f->dAp; P->Ai = f->dAi; P->Alen = f->dAlen; P->Aoff = f->dAoff; P->Alimbs = f->dAlimbs; P->q = f->dq;
    if (!rc && (hipEventCreate(&f->ev0) != hipSuccess || hipEventCreate(&f->ev1) != hipSuccess)) rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc) { f->hs.ticket = 0; rc = slip_hip_factor_reset(f); }
    if (rc) { slip_hip_factor_destroy(f); return rc; }
    *out = f;
    return SLIP_HIP_OK;
}

/* new stride of the dense vectors; x and the inverse cache are scratch, the pivot records of the
 * K committed columns survive (their cached inverses are recomputed on demand) */
static int grow_x_keep(slip_hip_factor *f, int64_t xcap, int32_t K)
{
    SlipParams *P = &f->P;
    if (xcap > (1 << 28)) return SLIP_HIP_OUT_OF_MEMORY;
    SlipPiv *keep = NULL;
    if (K > 0) {
        keep = (SlipPiv *) malloc((size_t) K * sizeof(SlipPiv));
        if (!keep) return SLIP_HIP_OUT_OF_MEMORY;
        if (hipMemcpy(keep, P->piv.p_, (size_t) K * sizeof(SlipPiv), hipMemcpyDeviceToHost) != hipSuccess) { free(keep); return SLIP_HIP_DEVICE_ERROR; }
    }
    /* a wider stride may change how many workers fit: alloc_x re-applies the HBM budget at the new stride and never goes
     * beyond the count the private row arrays were sized for; on failure the handle keeps its old buffers */
    int e = alloc_x(f, (int32_t) xcap, 1);
    if (!e && K > 0) {
        for (int32_t k = 0; k < K; k++) keep[k].invlen = 0;
        if (hipMemcpy(P->piv.p_, keep, (size_t) K * sizeof(SlipPiv), hipMemcpyHostToDevice) != hipSuccess) e = SLIP_HIP_DEVICE_ERROR;
    }
    free(keep);
    return e;
}

#ifdef SLIP_EMULATE
static unsigned long long slip_emu_seed = 1;
extern "C" void slip_emu_set_seed(unsigned long long s) { slip_emu_seed = s; }
/* weak-store mode of the emulator (tests/emu/fiber_emu.h): sc1 stores land late and out of order */
extern "C" void slip_emu_set_weak(int on) { emu::set_weak(on); }
/* overwrite the length of entry t of L (isU 0) or U (isU 1): what a protocol error on the device would leave behind */
extern "C" int slip_emu_corrupt_entry(slip_hip_factor *f, int isU, long long t, int newlen)
{
    if (!f || t < 0 || t >= (isU ? f->hs.Unz : f->hs.Lnz)) return -1;
    SlipEnt *e = (isU ? f->P.Ue : f->P.Le) + t;
    e->len = newlen;
    return 0;
}
#endif

static int launch_columns(slip_hip_factor *f, hipStream_t stream)
{
    f->P.k0 = f->hs.F; f->P.t0 = f->hs.ticket;
    f->hs.stop = INT64_MAX; f->hs.exited = 0; for (int q_ = 0; q_ < 32; q_++) f->hs.farm_hint[q_] = 0; f->hs.dbg_who = 0; f->hs.committer_up = 0; f->hs.committer_where = 0;
    f->P.st = f->ds; f->P.in_factor = 1;
    { const int e = upload_state(f, stream); if (e) return e; }
    /* no more workers than columns left */
    int32_t W = f->nworkers;
    /* block 0 of a launch with at least two workgroups of at least two waves is the committer (ref_lu_pipe_commit.h) */
    f->P.committer = W >= 2 && f->waves >= 2 && f->bitmap_in_lds && f->scratch_in_lds && !f->P.no_early && !f->no_committer && f->P.pivot_scheme != 2 && f->P.pkg.p_ != NULL;
    if (W > f->P.k_stop - f->hs.F + f->P.committer) W = f->P.k_stop - f->hs.F + f->P.committer;
    if (W < 1) W = 1;
    if (W < 2) f->P.committer = 0;
    if (f->P.committer && slip_commit_lds_words(f->P.n, 0) > f->lds_words) f->P.committer = 0;      /* (cannot happen: plan_launch made room) */
    f->P.engine = f->P.committer && f->engine_ok && slip_commit_lds_words(f->P.n, 1) <= f->lds_words;
    f->P.nworkers = W;
    if (f->P.committer) CK(hipMemsetAsync(f->P.pkg.p_, 0, (size_t) W * (SLIP_PKG_WORDS + SLIP_MBOX_WORDS) * 4, stream));
    f->P.farm = W >= 2 && !f->no_farm && f->P.jobs.p_ != NULL;
    if (f->P.farm) CK(hipMemsetAsync(f->P.jobs.p_, 0, (size_t) W * SLIP_JOB_WORDS * 4, stream));
    CK(hipEventRecord(f->ev0, stream));
#ifndef SLIP_EMULATE
    const size_t lds_bytes = (size_t) f->lds_words * 4;
    const dim3 grid(W), block(64 * f->waves);
#define SLIP_LAUNCH(FAST) do { \
        CK(hipFuncSetAttribute((const void *) slip_factor_kernel<FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes)); \
        hipLaunchKernelGGL((slip_factor_kernel<FAST>), grid, block, lds_bytes, stream, f->P, f->ds); } while (0)
    if (f->bitmap_in_lds && f->scratch_in_lds) SLIP_LAUNCH(true);
    else SLIP_LAUNCH(false);
#undef SLIP_LAUNCH
    CK(hipGetLastError());
#else
    {
        const SlipParams P = f->P; SlipState *ds = f->ds;
        const int fast = f->bitmap_in_lds && f->scratch_in_lds;
        const size_t words = (((size_t) f->lds_words + 64) + 3) & ~(size_t) 3;      /* every emulated workgroup's LDS 16-byte aligned, as on the device */
        uint32_t *lds_all = (uint32_t *) calloc((size_t) W * words, 4);
        if (!lds_all) return SLIP_HIP_OUT_OF_MEMORY;
        emu::set_seed(slip_emu_seed);
        emu::launch(W, 64 * f->waves, [P, ds, fast, lds_all, words]() {
            SlipParams Pw;
            slip_worker_params(&Pw, P, slip_block());
            uint32_t *lds = lds_all + (size_t) slip_block() * words;
            if (Pw.committer && slip_block() == 0) { if (fast) slip_committer<true>(Pw, ds, lds); else slip_committer<false>(Pw, ds, lds); }
            else if (fast) slip_factor_worker<true>(Pw, ds, lds);
            else slip_factor_worker<false>(Pw, ds, lds);
            slip_worker_exit(Pw, ds);
        });
        free(lds_all);
    }
#endif
    CK(hipEventRecord(f->ev1, stream));
    CK(hipMemcpyAsync(&f->hs, f->ds, sizeof(SlipState), hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, f->ev0, f->ev1));
    f->kernel_ms += ms;
    f->launches++;
    /* where the launch ended: the frontier; why: the stop word, if it names the frontier column */
    SlipState *h = &f->hs;
    h->k_next = h->F;
    h->status = SLIPDEV_OK; h->status_k = h->F;
    if ((int)(h->stop & 0xFF) == SLIPDEV_INTERNAL) h->status = SLIPDEV_INTERNAL;
    else if (h->stop != INT64_MAX && (h->stop >> 8) == (int64_t) h->F) h->status = (int)(h->stop & 0xFF);
    else if (h->F < f->P.k_stop) h->status = SLIPDEV_INTERNAL;       /* the workers left without a reason */
    return 0;
}

extern "C" int slip_hip_factor_run(slip_hip_factor *f, int32_t kmax, void *stream_v)
{
    if (!f || f->factors_only) return SLIP_HIP_INCORRECT_INPUT;
    hipStream_t stream = (hipStream_t) stream_v;
    SlipParams *P = &f->P;
    SlipState *h = &f->hs;
    if (!P->xd || !P->xrow) return SLIP_HIP_OUT_OF_MEMORY;
    if (kmax <= 0 || kmax > f->n) kmax = f->n;
    f->kernel_ms = 0; f->launches = 0; f->window_end = 0;
    P->k_stop = kmax;
    int rc = SLIP_HIP_OK;
    while (h->F < kmax) {
        int e = launch_columns(f, stream);
        if (e) { rc = e; break; }
        if (h->status == SLIPDEV_OK) continue;
        if (h->status == SLIPDEV_SINGULAR) { rc = SLIP_HIP_SINGULAR; break; }
        if (h->status == SLIPDEV_WINDOW_END) { f->window_end = 1; break; }
        if (h->status == SLIPDEV_GROW_L) {
            /* the slab that ran out is the one to double (cf. slip_sparse_realloc.c) */
            int64_t nz = P->Lcap_nz * 2, nl = P->Lcap_nl * 2;
            if ((e = dev_grow(&P->Li, h->Lnz, nz)) || (e = dev_grow(&P->Le, h->Lnz, nz)) ||
                (e = dev_grow(&P->Llimbs, h->Lnl, nl))) { rc = e; break; }
            P->Lcap_nz = nz; P->Lcap_nl = nl;
        } else if (h->status == SLIPDEV_GROW_U) {
            int64_t nz = P->Ucap_nz * 2, nl = P->Ucap_nl * 2;
            if ((e = dev_grow(&P->Ui, h->Unz, nz)) || (e = dev_grow(&P->Ue, h->Unz, nz)) ||
                (e = dev_grow(&P->Ulimbs, h->Unl, nl))) { rc = e; break; }
            P->Ucap_nz = nz; P->Ucap_nl = nl;
        } else if (h->status == SLIPDEV_GROW_X) {
            e = grow_x_keep(f, (int64_t) P->xcap * 2, h->F);
            if (e) { rc = e; break; }
        } else {
            fprintf(stderr, "slip_hip: kernel stopped with internal status %d at column %d (frontier %d, ready %d, stop %lld; wait %d of column %d: %d %d)\n",
                    h->status, h->status_k, h->F, h->F2, (long long) h->stop, h->dbg_who, h->dbg_k, h->dbg_a, h->dbg_b);
            rc = SLIP_HIP_DEVICE_ERROR; break;
        }
    }
    f->last_status = rc;
    return rc;
}

/* ---- a handle around GIVEN factors (the caller's L, U, pinv): what SLIP_LU_solve receives ---- */
static int slab_to_entries(int64_t nz, const int32_t *len, const uint64_t *limbs, SlipEnt *ent, int32_t *maxdig, int64_t *nl_out)
{
    int64_t o = 0;
    for (int64_t t = 0; t < nz; t++) {
        int64_t l = len[t] < 0 ? -(int64_t) len[t] : len[t];
        const uint64_t *src = limbs + o;
        int64_t le = l;
        while (le > 0 && src[le - 1] == 0) le--;
        int32_t dig = (int32_t)(2 * le);
        if (le > 0 && (src[le - 1] >> 32) == 0) dig--;
        ent[t].off = o; ent[t].len = len[t] < 0 ? -dig : dig;
        ent[t].bits = dig ? 32 * dig - __builtin_clz(dig & 1 ? (uint32_t) src[le - 1] : (uint32_t)(src[le - 1] >> 32)) : 0;
        if (dig > *maxdig) *maxdig = dig;
        o += l;
    }
    *nl_out = o;
    return 0;
}

extern "C" int slip_hip_factor_from_factors(slip_hip_factor **out, int32_t n,
                                            const int64_t *Lp, const int32_t *Li, const int32_t *Llen, const uint64_t *Llimbs,
                                            const int64_t *Up, const int32_t *Ui, const int32_t *Ulen, const uint64_t *Ulimbs,
                                            const int32_t *pinv, const slip_hip_options *opt_in)
{
    if (!out || n <= 0 || !Lp || !Li || !Llen || !Llimbs || !Up || !Ui || !Ulen || !Ulimbs || !pinv) return SLIP_HIP_INCORRECT_INPUT;
    *out = NULL;
    if (slip_hip_device_count() <= 0) {
        fprintf(stderr, "slip_hip: no HIP device available -- this library has no CPU fallback\n");
        return SLIP_HIP_DEVICE_ERROR;
    }
    slip_hip_options opt;
    if (opt_in) opt = *opt_in; else slip_hip_default_options(&opt);
    const int64_t lnz = Lp[n], unz = Up[n];
    if (Lp[0] != 0 || Up[0] != 0 || lnz < n || unz < n) return SLIP_HIP_INCORRECT_INPUT;
    int32_t *rowperm = (int32_t *) malloc((size_t) n * 4);
    SlipEnt *Le = (SlipEnt *) malloc((size_t) lnz * sizeof(SlipEnt)), *Ue = (SlipEnt *) malloc((size_t) unz * sizeof(SlipEnt));
    SlipPiv *piv = (SlipPiv *) calloc((size_t) n, sizeof(SlipPiv));
    int64_t *Lo = (int64_t *) malloc(((size_t) n + 1) * 8), *Uo = (int64_t *) malloc(((size_t) n + 1) * 8);
    if (!rowperm || !Le || !Ue || !piv || !Lo || !Uo) { free(rowperm); free(Le); free(Ue); free(piv); free(Lo); free(Uo); return SLIP_HIP_OUT_OF_MEMORY; }
    int bad = 0;
    for (int32_t i = 0; i < n; i++) rowperm[i] = -1;
    for (int32_t i = 0; i < n && !bad; i++) { if (pinv[i] < 0 || pinv[i] >= n || rowperm[pinv[i]] >= 0) bad = 1; else rowperm[pinv[i]] = i; }
    for (int64_t t = 0; t < lnz && !bad; t++) if (Li[t] < 0 || Li[t] >= n) bad = 1;
    for (int64_t t = 0; t < unz && !bad; t++) if (Ui[t] < 0 || Ui[t] >= n) bad = 1;
    int32_t maxdig = 1; int64_t lnl = 0, unl = 0;
    if (!bad) { slab_to_entries(lnz, Llen, Llimbs, Le, &maxdig, &lnl); slab_to_entries(unz, Ulen, Ulimbs, Ue, &maxdig, &unl); }
    /* pivot records: rho_k is the entry of L(:,k) in the pivot row (slip_get_pivot.c:178-182) */
    for (int32_t k = 0; k < n && !bad; k++) {
        if (Lp[k + 1] < Lp[k] || Up[k + 1] <= Up[k]) { bad = 1; break; }
        int64_t at = -1;
        for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) if (Li[t] == rowperm[k]) at = t;
        if (at < 0 || Le[at].len == 0 || Ui[Up[k + 1] - 1] != rowperm[k]) { bad = 1; break; }
        const uint64_t *pv = Llimbs + Le[at].off;
        const int32_t dig = Le[at].len < 0 ? -Le[at].len : Le[at].len;
        int z = 0; { int64_t w = 0; while (pv[w] == 0) { w++; z += 64; } z += __builtin_ctzll(pv[w]); }
        SlipPiv pr; memset(&pr, 0, sizeof pr);
        pr.off = Le[at].off; pr.len = Le[at].len; pr.bits = Le[at].bits; pr.ctz = z; pr.invlen = 0; pr.lo = pv[0];
        if (dig <= 2) { uint64_t d = pr.lo >> z, x = d; for (int r = 0; r < 5; r++) x *= 2 - d * x; pr.inv64 = x; }
        piv[k] = pr;
    }
    if (!bad) for (int32_t k = 0; k <= n; k++) { Lo[k] = k < n ? Le[Lp[k]].off : lnl; Uo[k] = k < n ? Ue[Up[k]].off : unl; }
    if (bad) { free(rowperm); free(Le); free(Ue); free(piv); free(Lo); free(Uo); return SLIP_HIP_INCORRECT_INPUT; }

    slip_hip_factor *f = (slip_hip_factor *) calloc(1, sizeof(slip_hip_factor));
    if (!f) { free(rowperm); free(Le); free(Ue); free(piv); free(Lo); free(Uo); return SLIP_HIP_OUT_OF_MEMORY; }
    SlipParams *P = &f->P;
    f->n = n; f->factors_only = 1;
    apply_options(f, opt);
    if (f->workers_asked <= 0) f->workers_asked = 64;  /* right-hand sides in flight; more are taken in turn */
    P->n = n; P->pivot_scheme = opt.pivot; P->limb_cap = 0; P->k_stop = n;
    P->Lcap_nz = lnz; P->Ucap_nz = unz; P->Lcap_nl = lnl > 0 ? lnl : 1; P->Ucap_nl = unl > 0 ? unl : 1;
    int rc = 0;
#define A_(call) do { if (!rc) rc = (call); } while (0)
    A_(dev_alloc(&P->pinv.p_, n)); A_(dev_alloc(&P->row_perm.p_, n));
    A_(dev_alloc(&P->piv.p_, n)); A_(dev_alloc(&P->Lready.p_, n)); A_(dev_alloc(&P->dbg, 24 * (int64_t) n + 4096)); A_(dev_alloc(&P->sw_row.p_, n)); A_(dev_alloc(&P->sw_pos.p_, n));
    A_(dev_alloc(&P->Lp, (int64_t) n + 1)); A_(dev_alloc(&P->Lo, (int64_t) n + 1)); A_(dev_alloc(&P->Li, lnz)); A_(dev_alloc(&P->Le, lnz)); A_(dev_alloc(&P->Llimbs, P->Lcap_nl));
    A_(dev_alloc(&P->Up, (int64_t) n + 1)); A_(dev_alloc(&P->Uo, (int64_t) n + 1)); A_(dev_alloc(&P->Ui, unz)); A_(dev_alloc(&P->Ue, unz)); A_(dev_alloc(&P->Ulimbs, P->Ucap_nl));
    A_(dev_alloc(&f->ds, 1));
    if (!rc && hipMemset(P->dbg, 0, ((size_t) n * 24 + 4096) * 4) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc) rc = alloc_x(f, 2 * maxdig + 8, 0);
#undef A_
#define UP_(dst, src, bytes) do { if (!rc && (bytes) > 0 && hipMemcpy((void *)(dst), (src), (size_t)(bytes), hipMemcpyHostToDevice) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR; } while (0)
    UP_(P->pinv.p_, pinv, (size_t) n * 4); UP_(P->row_perm.p_, rowperm, (size_t) n * 4); UP_(P->piv.p_, piv, (size_t) n * sizeof(SlipPiv));
    UP_(P->Lp, Lp, ((size_t) n + 1) * 8); UP_(P->Lo, Lo, ((size_t) n + 1) * 8); UP_(P->Li, Li, (size_t) lnz * 4); UP_(P->Le, Le, (size_t) lnz * sizeof(SlipEnt)); UP_(P->Llimbs, Llimbs, (size_t) lnl * 8);
    UP_(P->Up, Up, ((size_t) n + 1) * 8); UP_(P->Uo, Uo, ((size_t) n + 1) * 8); UP_(P->Ui, Ui, (size_t) unz * 4); UP_(P->Ue, Ue, (size_t) unz * sizeof(SlipEnt)); UP_(P->Ulimbs, Ulimbs, (size_t) unl * 8);
    if (!rc && (hipEventCreate(&f->ev0) != hipSuccess || hipEventCreate(&f->ev1) != hipSuccess)) rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc) {
        memset(&f->hs, 0, sizeof f->hs);
        f->hs.F = n; f->hs.F2 = n; f->hs.stop = INT64_MAX;
        f->hs.k_next = n; f->hs.Lnz = lnz; f->hs.Unz = unz; f->hs.Lnl = lnl; f->hs.Unl = unl; f->hs.Lnl_exact = lnl; f->hs.Unl_exact = unl;
        f->hs.c_maxdig = (unsigned long long) maxdig;
        UP_(f->ds, &f->hs, sizeof(SlipState));
    }
#undef UP_
    free(rowperm); free(Le); free(Ue); free(piv); free(Lo); free(Uo);
    if (rc) { slip_hip_factor_destroy(f); return rc; }
    *out = f;
    return SLIP_HIP_OK;
}

/* ---- continue from a prefix (the subtree farm's last step) ----
 * The first K columns of the factorisation are GIVEN (the blocks' columns, factorised elsewhere and rescaled to the global
 * pivot chain: slip_lu_amd/parallel.py): L(:,k), U(:,k) in the reference's CSC with ORIGINAL row ids and the pivot row of
 * every column.  The handle is put into the state a launch that stopped at column K leaves -- the permutation replayed from
 * the pivot rows (slip_get_pivot.c:164-176), the pivots' records, the column pointers, every given column ready -- and
 * slip_hip_factor_run goes on with column K (the separator columns of the farm), SLIP_LU_factorize.c:190-264 from k = K. */
extern "C" int slip_hip_factor_set_prefix(slip_hip_factor *f, int32_t K,
                                          const int64_t *Lp, const int32_t *Li, const int32_t *Llen, const uint64_t *Llimbs,
                                          const int64_t *Up, const int32_t *Ui, const int32_t *Ulen, const uint64_t *Ulimbs,
                                          const int32_t *piv_row)
{
    if (!f || f->factors_only || K < 0 || K > f->n) return SLIP_HIP_INCORRECT_INPUT;
    if (K > 0 && (!Lp || !Li || !Llen || !Llimbs || !Up || !Ui || !Ulen || !Ulimbs || !piv_row)) return SLIP_HIP_INCORRECT_INPUT;
    { const int e = slip_hip_factor_reset(f); if (e) return e; }
    if (K == 0) return SLIP_HIP_OK;
    SlipParams *P = &f->P;
    const int32_t n = f->n;
    const int64_t lnz = Lp[K], unz = Up[K];
    if (Lp[0] != 0 || Up[0] != 0 || lnz < K || unz < K) return SLIP_HIP_INCORRECT_INPUT;
    int32_t *pinv = (int32_t *) malloc((size_t) n * 4), *rowperm = (int32_t *) malloc((size_t) n * 4);
    int32_t *swr = (int32_t *) malloc((size_t) K * 4), *swp = (int32_t *) malloc((size_t) K * 4), *ready = (int32_t *) malloc((size_t) K * 4);
    SlipEnt *Le = (SlipEnt *) malloc((size_t) lnz * sizeof(SlipEnt)), *Ue = (SlipEnt *) malloc((size_t) unz * sizeof(SlipEnt));
    SlipPiv *piv = (SlipPiv *) calloc((size_t) K, sizeof(SlipPiv));
    int64_t *Lo = (int64_t *) malloc(((size_t) K + 1) * 8), *Uo = (int64_t *) malloc(((size_t) K + 1) * 8);
    int rc = SLIP_HIP_OK;
#define FREE_ALL_() do { free(pinv); free(rowperm); free(swr); free(swp); free(ready); free(Le); free(Ue); free(piv); free(Lo); free(Uo); } while (0)
    if (!pinv || !rowperm || !swr || !swp || !ready || !Le || !Ue || !piv || !Lo || !Uo) { FREE_ALL_(); return SLIP_HIP_OUT_OF_MEMORY; }
    int bad = 0;
    for (int64_t t = 0; t < lnz && !bad; t++) if (Li[t] < 0 || Li[t] >= n) bad = 1;
    for (int64_t t = 0; t < unz && !bad; t++) if (Ui[t] < 0 || Ui[t] >= n) bad = 1;
    int32_t maxdig = 1; int64_t lnl = 0, unl = 0;
    if (!bad) { slab_to_entries(lnz, Llen, Llimbs, Le, &maxdig, &lnl); slab_to_entries(unz, Ulen, Ulimbs, Ue, &maxdig, &unl); }
    /* the permutation: column k's pivot row changes places with the row at position k (slip_get_pivot.c:164-176) */
    for (int32_t i = 0; i < n; i++) { pinv[i] = i; rowperm[i] = i; }
    for (int32_t k = 0; k < K && !bad; k++) {
        const int32_t r = piv_row[k];
        if (r < 0 || r >= n || pinv[r] < k || Lp[k + 1] < Lp[k] || Up[k + 1] <= Up[k]) { bad = 1; break; }
        const int32_t p = pinv[r], d = rowperm[k];
        swr[k] = d; swp[k] = p; ready[k] = 1;
        rowperm[k] = r; rowperm[p] = d; pinv[r] = k; pinv[d] = p;
        if (p == k) { pinv[r] = k; rowperm[k] = r; }
        /* rho_k is the entry of L(:,k) in the pivot row (slip_get_pivot.c:178-182), and the last entry of U(:,k) */
        int64_t at = -1;
        for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) if (Li[t] == r) at = t;
        if (at < 0 || Le[at].len == 0 || Ui[Up[k + 1] - 1] != r) { bad = 1; break; }
        const uint64_t *pv = Llimbs + Le[at].off;
        const int32_t dig = Le[at].len < 0 ? -Le[at].len : Le[at].len;
        int z = 0; { int64_t w = 0; while (pv[w] == 0) { w++; z += 64; } z += __builtin_ctzll(pv[w]); }
        SlipPiv pr; memset(&pr, 0, sizeof pr);
        pr.off = Le[at].off; pr.len = Le[at].len; pr.bits = Le[at].bits; pr.ctz = z; pr.invlen = 0; pr.lo = pv[0];
        if (dig <= 2) { uint64_t dd = pr.lo >> z, x = dd; for (int q = 0; q < 5; q++) x *= 2 - dd * x; pr.inv64 = x; }
        piv[k] = pr;
    }
    if (!bad) for (int32_t k = 0; k <= K; k++) { Lo[k] = k < K ? Le[Lp[k]].off : lnl; Uo[k] = k < K ? Ue[Up[k]].off : unl; }
    if (bad) { FREE_ALL_(); return SLIP_HIP_INCORRECT_INPUT; }
    /* room for the prefix and as much again (the run doubles a slab that fills up) */
    if (lnz + n > P->Lcap_nz || lnl + n > P->Lcap_nl) {
        const int64_t nz = 2 * (lnz + n), nl = 2 * (lnl + n);
        if ((rc = dev_grow(&P->Li, 0, nz)) || (rc = dev_grow(&P->Le, 0, nz)) || (rc = dev_grow(&P->Llimbs, 0, nl))) { FREE_ALL_(); return rc; }
        P->Lcap_nz = nz; P->Lcap_nl = nl;
    }
    if (unz + n > P->Ucap_nz || unl + n > P->Ucap_nl) {
        const int64_t nz = 2 * (unz + n), nl = 2 * (unl + n);
        if ((rc = dev_grow(&P->Ui, 0, nz)) || (rc = dev_grow(&P->Ue, 0, nz)) || (rc = dev_grow(&P->Ulimbs, 0, nl))) { FREE_ALL_(); return rc; }
        P->Ucap_nz = nz; P->Ucap_nl = nl;
    }
#define UP_(dst, src, bytes) do { if (!rc && (bytes) > 0 && hipMemcpy((void *)(dst), (src), (size_t)(bytes), hipMemcpyHostToDevice) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR; } while (0)
    UP_(P->pinv.p_, pinv, (size_t) n * 4); UP_(P->row_perm.p_, rowperm, (size_t) n * 4); UP_(P->piv.p_, piv, (size_t) K * sizeof(SlipPiv));
    UP_(P->sw_row.p_, swr, (size_t) K * 4); UP_(P->sw_pos.p_, swp, (size_t) K * 4); UP_(P->Lready.p_, ready, (size_t) K * 4);
    UP_(P->Lp, Lp, ((size_t) K + 1) * 8); UP_(P->Lo, Lo, ((size_t) K + 1) * 8); UP_(P->Li, Li, (size_t) lnz * 4); UP_(P->Le, Le, (size_t) lnz * sizeof(SlipEnt)); UP_(P->Llimbs, Llimbs, (size_t) lnl * 8);
    UP_(P->Up, Up, ((size_t) K + 1) * 8); UP_(P->Uo, Uo, ((size_t) K + 1) * 8); UP_(P->Ui, Ui, (size_t) unz * 4); UP_(P->Ue, Ue, (size_t) unz * sizeof(SlipEnt)); UP_(P->Ulimbs, Ulimbs, (size_t) unl * 8);
    if (!rc) {
        SlipState *h = &f->hs;
        h->F = K; h->Fpiv = piv_row[K - 1]; h->F2 = K; h->k_next = K; h->status_k = K;
        h->Lnz = lnz; h->Unz = unz; h->Lnl = lnl; h->Unl = unl; h->Lnl_exact = lnl; h->Unl_exact = unl;
        h->c_maxdig = (unsigned long long) maxdig;
        rc = upload_state(f, 0);
        if (!rc && hipStreamSynchronize(0) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    }
#undef UP_
    FREE_ALL_();
#undef FREE_ALL_
    /* values wider than the private x rows were sized for: the stride grows now rather than in the first launch */
    if (!rc && 2 * (int64_t) maxdig + 8 > P->xcap) rc = grow_x_keep(f, 2 * (int64_t) maxdig + 8, K);
    f->last_status = rc;
    return rc;
}

/* ---- REF triangular solves on the resident factors (SLIP_LU_solve.c:41-86) ---- */
#ifndef SLIP_SOLVE_HELPERS
#define SLIP_SOLVE_HELPERS 16
#endif
static int launch_solve(slip_hip_factor *f, const SlipSolveArgs &A, int32_t *rhs_done, hipStream_t stream)
{
    f->P.t0 = f->hs.ticket;
    f->hs.stop = INT64_MAX; f->hs.exited = 0;
    f->P.farm = 0; f->P.committer = 0; f->P.st = f->ds; f->P.in_factor = 0;
    for (int q_ = 0; q_ < 32; q_++) f->hs.farm_hint[q_] = 0;
    { const int e = upload_state(f, stream); if (e) return e; }
    /* one workgroup per right-hand side in flight; with fewer right-hand sides than workgroups, up to SLIP_SOLVE_HELPERS more
     * that only help with the long update queues (ref_lu_pipe_cols.h: slip_solve_worker) */
    int32_t W = f->nworkers;
    if (W > A.nrhs) {
        int32_t H = f->no_farm || !f->P.jobs.p_ ? 0 : SLIP_SOLVE_HELPERS;
        if (A.nrhs + H > W) H = W - A.nrhs;
        W = A.nrhs + H;
        f->P.farm = H > 0;
    }
    if (W < 1) W = 1;
    f->P.nworkers = W;
    if (f->P.farm) CK(hipMemsetAsync(f->P.jobs.p_, 0, (size_t) W * SLIP_JOB_WORDS * 4, stream));
    CK(hipEventRecord(f->ev0, stream));
#ifndef SLIP_EMULATE
    const size_t lds_bytes = (size_t) f->lds_words * 4;
    const dim3 grid(W), block(64 * f->waves);
#define SLIP_LAUNCH(FAST) do { \
        CK(hipFuncSetAttribute((const void *) slip_solve_kernel<FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes)); \
        hipLaunchKernelGGL((slip_solve_kernel<FAST>), grid, block, lds_bytes, stream, f->P, f->ds, A, rhs_done); } while (0)
    if (f->bitmap_in_lds && f->scratch_in_lds) SLIP_LAUNCH(true);
    else SLIP_LAUNCH(false);
#undef SLIP_LAUNCH
    CK(hipGetLastError());
#else
    {
        const SlipParams P = f->P; SlipState *ds = f->ds;
        const int fast = f->bitmap_in_lds && f->scratch_in_lds;
        const size_t words = (((size_t) f->lds_words + 64) + 3) & ~(size_t) 3;      /* every emulated workgroup's LDS 16-byte aligned, as on the device */
        uint32_t *lds_all = (uint32_t *) calloc((size_t) W * words, 4);
        if (!lds_all) return SLIP_HIP_OUT_OF_MEMORY;
        emu::set_seed(slip_emu_seed);
        emu::launch(W, 64 * f->waves, [P, ds, fast, A, rhs_done, lds_all, words]() {
            SlipParams Pw;
            slip_worker_params(&Pw, P, slip_block());
            uint32_t *lds = lds_all + (size_t) slip_block() * words;
            if (fast) slip_solve_worker<true>(Pw, ds, A, rhs_done, lds);
            else slip_solve_worker<false>(Pw, ds, A, rhs_done, lds);
        });
        free(lds_all);
    }
#endif
    CK(hipEventRecord(f->ev1, stream));
    CK(hipMemcpyAsync(&f->hs, f->ds, sizeof(SlipState), hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, f->ev0, f->ev1));
    f->solve_ms += ms;
    return 0;
}

extern "C" int slip_hip_factor_solve(slip_hip_factor *f, int32_t nrhs, const int32_t *blen, const uint64_t *blimbs,
                                     int32_t **xlen_out, uint64_t **xlimbs_out, int64_t *xnl_out, void *stream_v)
{
    if (!f || nrhs <= 0 || !blen || !blimbs || !xlen_out || !xlimbs_out || !xnl_out) return SLIP_HIP_INCORRECT_INPUT;
    *xlen_out = NULL; *xlimbs_out = NULL; *xnl_out = 0;
    const int32_t n = f->n;
    if (f->hs.F != n) return SLIP_HIP_INCORRECT_INPUT;          /* needs the complete factorisation */
    if (!f->P.xd || !f->P.xrow) return SLIP_HIP_OUT_OF_MEMORY;
    hipStream_t stream = (hipStream_t) stream_v;
    SlipParams *P = &f->P;
    const int64_t ne = (int64_t) n * nrhs;
    /* b: signed limb counts -> signed digit counts + offsets (zero high limbs trimmed) */
    int32_t *hlen = (int32_t *) malloc((size_t) ne * 4);
    int64_t *hoff = (int64_t *) malloc((size_t) ne * 8);
    if (!hlen || !hoff) { free(hlen); free(hoff); return SLIP_HIP_OUT_OF_MEMORY; }
    int64_t o = 0; int32_t maxdig = 1;
    for (int64_t t = 0; t < ne; t++) {
        int64_t l = blen[t] < 0 ? -(int64_t) blen[t] : blen[t];
        hoff[t] = o;
        const uint64_t *src = blimbs + o;
        o += l;
        while (l > 0 && src[l - 1] == 0) l--;
        int32_t dig = (int32_t)(2 * l);
        if (l > 0 && (src[l - 1] >> 32) == 0) dig--;
        hlen[t] = blen[t] < 0 ? -dig : dig;
        if (dig > maxdig) maxdig = dig;
    }
    const int64_t bl = o;
    SlipSolveArgs A; memset(&A, 0, sizeof A);
    int32_t *dblen = NULL, *dolen = NULL, *ddone = NULL; int64_t *dboff = NULL, *dooff = NULL; uint64_t *dbl = NULL, *dol = NULL;
    int32_t *xl = NULL, *hdone = NULL; uint64_t *xlimbs = NULL, *raw = NULL; int64_t *hooff = NULL;
    int rc = 0;
    /* x grows to about |b| * det: make room once (the kernel still reports GROW_X if this is short) */
    {
        int64_t want = 2 * ((int64_t) f->hs.c_maxdig + 2) + maxdig + 8;
        if (want > P->xcap) rc = grow_x_keep(f, want, n);
    }
    /* every right-hand side owns `ostride` limbs of the output slab */
    int64_t ostride = (int64_t) n * (((int64_t) f->hs.c_maxdig + maxdig) / 2 + 1) + 64;
#define A_(call) do { if (!rc) rc = (call); } while (0)
    A_(dev_alloc(&dblen, ne)); A_(dev_alloc(&dboff, ne)); A_(dev_alloc(&dbl, bl > 0 ? bl : 1));
    A_(dev_alloc(&dolen, ne)); A_(dev_alloc(&dooff, ne)); A_(dev_alloc(&dol, ostride * nrhs)); A_(dev_alloc(&ddone, nrhs));
#undef A_
    hdone = (int32_t *) calloc((size_t) nrhs, 4);
    if (!hdone) rc = rc ? rc : SLIP_HIP_OUT_OF_MEMORY;
    if (!rc && (hipMemcpy(dblen, hlen, (size_t) ne * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dboff, hoff, (size_t) ne * 8, hipMemcpyHostToDevice) != hipSuccess ||
                (bl > 0 && hipMemcpy(dbl, blimbs, (size_t) bl * 8, hipMemcpyHostToDevice) != hipSuccess) ||
                hipMemset(ddone, 0, (size_t) nrhs * 4) != hipSuccess))
        rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc) {
        SlipState *h = &f->hs;
        f->solve_ms = 0;
        A.nrhs = nrhs; A.blen = dblen; A.boff = dboff; A.blimbs = dbl; A.olen = dolen; A.ooff = dooff; A.olimbs = dol;
        A.ocap = ostride * nrhs; A.ostride = ostride;
        for (int guard = 0; !rc && guard < 64; guard++) {
            int e = launch_solve(f, A, ddone, stream);
            if (e) { rc = e; break; }
            if (hipMemcpy(hdone, ddone, (size_t) nrhs * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = SLIP_HIP_DEVICE_ERROR; break; }
            int all = 1;
            for (int32_t c = 0; c < nrhs; c++) if (!hdone[c]) all = 0;
            if (all) break;
            const int status = h->stop == INT64_MAX ? SLIPDEV_INTERNAL : (int)(h->stop & 0xFF);
            if (status == SLIPDEV_GROW_X) rc = grow_x_keep(f, (int64_t) P->xcap * 2, n);
            else if (status == SLIPDEV_GROW_U) {
                /* the output regions: twice the stride; finished right-hand sides are simply solved again */
                dev_free(dol); dol = NULL;
                ostride *= 2;
                rc = dev_alloc(&dol, ostride * nrhs);
                if (!rc && hipMemset(ddone, 0, (size_t) nrhs * 4) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
                A.olimbs = dol; A.ocap = ostride * nrhs; A.ostride = ostride;
            } else {
                fprintf(stderr, "slip_hip: solve kernel stopped with internal status %d (stop %lld)\n", status, (long long) h->stop);
                rc = SLIP_HIP_DEVICE_ERROR;
            }
        }
    }
    if (!rc) {
        /* gather the per-right-hand-side regions into the dense (rhs, position) slab of the ABI */
        xl = (int32_t *) malloc((size_t) ne * 4);
        hooff = (int64_t *) malloc((size_t) ne * 8);
        raw = (uint64_t *) malloc((size_t)(ostride * nrhs) * 8);
        if (!xl || !hooff || !raw) rc = SLIP_HIP_OUT_OF_MEMORY;
        if (!rc && (hipMemcpy(xl, dolen, (size_t) ne * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(hooff, dooff, (size_t) ne * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(raw, dol, (size_t)(ostride * nrhs) * 8, hipMemcpyDeviceToHost) != hipSuccess))
            rc = SLIP_HIP_DEVICE_ERROR;
        if (!rc) {
            int64_t nl = 0;
            for (int64_t t = 0; t < ne; t++) { const int32_t d = xl[t]; nl += ((d < 0 ? -d : d) + 1) >> 1; }
            xlimbs = (uint64_t *) malloc((size_t)(nl > 0 ? nl : 1) * 8);
            if (!xlimbs) rc = SLIP_HIP_OUT_OF_MEMORY;
            else {
                int64_t at = 0;
                for (int64_t t = 0; t < ne; t++) {
                    const int32_t d = xl[t], l = ((d < 0 ? -d : d) + 1) >> 1;
                    if (l) memcpy(xlimbs + at, raw + hooff[t], (size_t) l * 8);
                    at += l;
                    xl[t] = d < 0 ? -l : l;
                }
                *xlen_out = xl; *xlimbs_out = xlimbs; *xnl_out = nl;
                xl = NULL; xlimbs = NULL;
            }
        }
    }
    free(xl); free(xlimbs); free(hlen); free(hoff); free(hdone); free(raw); free(hooff);
    dev_free(dblen); dev_free(dboff); dev_free(dbl); dev_free(dolen); dev_free(dooff); dev_free(dol); dev_free(ddone);
    return rc;
}

extern "C" double slip_hip_factor_solve_ms(const slip_hip_factor *f) { return f ? f->solve_ms : 0.0; }

/* ---- subtree farm: multiply the committed factors by per-column scales ---- */
static void rescale_drop(slip_hip_factor *f)
{
    dev_free(f->rsLe); dev_free(f->rsUe); dev_free(f->rsLl); dev_free(f->rsUl); dev_free(f->rspiv);
    f->rsLe = f->rsUe = NULL; f->rsLl = f->rsUl = NULL; f->rspiv = NULL; f->rescaled = 0;
}

static int rescale_one(slip_hip_factor *f, int isL, int64_t nz, int64_t nl_alloc, const int32_t *sdig, const int32_t *dslen, const int64_t *dsoff,
                       const uint64_t *dslimbs, hipStream_t stream)
{
    SlipParams *P = &f->P;
    const int32_t K = f->hs.F;
    if (nz <= 0) return 0;
    SlipEnt *he = (SlipEnt *) malloc((size_t) nz * sizeof(SlipEnt));
    int32_t *hidx = (int32_t *) malloc((size_t) nz * 4), *hpinv = (int32_t *) malloc((size_t) f->n * 4);
    int64_t *hp = (int64_t *) malloc(((size_t) K + 1) * 8);
    if (!he || !hidx || !hpinv || !hp) { free(he); free(hidx); free(hpinv); free(hp); return SLIP_HIP_OUT_OF_MEMORY; }
    int rc = 0;
    if (hipMemcpy(he, isL ? P->Le : P->Ue, (size_t) nz * sizeof(SlipEnt), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(hidx, isL ? P->Li : P->Ui, (size_t) nz * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(hpinv, P->pinv.p_, (size_t) f->n * 4, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(hp, isL ? P->Lp : P->Up, ((size_t) K + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    /* lay the products out: digits(entry) + digits(scale) bound the product */
    int64_t o = 0;
    if (!rc) {
        int32_t c = 0;
        for (int64_t e = 0; e < nz; e++) {
            while (c + 1 <= K && hp[c + 1] <= e) c++;
            const int32_t sidx = isL ? c : hpinv[hidx[e]];
            if (sidx < 0 || sidx >= K) { rc = SLIP_HIP_INCORRECT_INPUT; break; }
            const int32_t la = he[e].len < 0 ? -he[e].len : he[e].len, lb = sdig[sidx];
            he[e].off = o;
            o += la ? (la + lb + 2) / 2 : 0;
        }
    }
    SlipEnt *de = NULL; uint64_t *dl = NULL;
    if (!rc) rc = dev_alloc(&de, nz);
    if (!rc) rc = dev_alloc(&dl, o > 0 ? o : 1);
    if (!rc && hipMemcpy(de, he, (size_t) nz * sizeof(SlipEnt), hipMemcpyHostToDevice) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc) {
        SlipRescaleArgs A; memset(&A, 0, sizeof A);
        A.ent = isL ? P->Le : P->Ue; A.limbs = isL ? P->Llimbs : P->Ulimbs; A.idx = isL ? P->Li : P->Ui; A.nz = nz;
        A.colp = isL ? P->Lp : NULL; A.ncols = K; A.pinv = P->pinv.p_;
        A.slen = dslen; A.soff = dsoff; A.slimbs = dslimbs; A.oent = de; A.olimbs = dl; A.pividx = f->rspiv; A.row_perm = P->row_perm.p_;
#ifndef SLIP_EMULATE
        int64_t blocks = (nz + 3) / 4; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(slip_rescale_kernel, dim3((unsigned) blocks), dim3(256), 0, stream, A);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
#else
        emu::launch(2, 128, [A]() { slip_rescale_body(A); }, 256 * 1024, 1);
#endif
    }
    if (!rc) {
        if (isL) { f->rsLe = de; f->rsLl = dl; f->rsLnl = o; }
        else     { f->rsUe = de; f->rsUl = dl; f->rsUnl = o; }
        de = NULL; dl = NULL;
    }
    if (de) dev_free(de);
    if (dl) dev_free(dl);
    (void) nl_alloc;
    free(he); free(hidx); free(hpinv); free(hp);
    return rc;
}

extern "C" int slip_hip_factor_rescale(slip_hip_factor *f, int32_t nscales, const int32_t *slen, const uint64_t *slimbs, void *stream_v)
{
    if (!f || !slen || !slimbs) return SLIP_HIP_INCORRECT_INPUT;
    hipStream_t stream = (hipStream_t) stream_v;
    const int32_t K = f->hs.F;
    if (K <= 0 || nscales != K) return SLIP_HIP_INCORRECT_INPUT;      /* one scale per committed column, no more, no fewer */
    /* scales: signed limb counts -> signed digit counts + limb offsets */
    int32_t *hd = (int32_t *) malloc((size_t) K * 4), *habs = (int32_t *) malloc((size_t) K * 4);
    int64_t *ho = (int64_t *) malloc((size_t) K * 8);
    if (!hd || !habs || !ho) { free(hd); free(habs); free(ho); return SLIP_HIP_OUT_OF_MEMORY; }
    int64_t o = 0;
    for (int32_t k = 0; k < K; k++) {
        int64_t l = slen[k] < 0 ? -(int64_t) slen[k] : slen[k];
        if (l == 0) { free(hd); free(habs); free(ho); return SLIP_HIP_INCORRECT_INPUT; }     /* a scale is a product of pivots: never zero */
        ho[k] = o;
        const uint64_t *src = slimbs + o;
        o += l;
        while (l > 0 && src[l - 1] == 0) l--;
        int32_t dig = (int32_t)(2 * l);
        if (l > 0 && (src[l - 1] >> 32) == 0) dig--;
        habs[k] = dig; hd[k] = slen[k] < 0 ? -dig : dig;
    }
    int32_t *dslen = NULL; int64_t *dsoff = NULL; uint64_t *dsl = NULL;
    int rc = 0;
    if (dev_alloc(&dslen, K) || dev_alloc(&dsoff, K) || dev_alloc(&dsl, o)) rc = SLIP_HIP_OUT_OF_MEMORY;
    if (!rc && (hipMemcpy(dslen, hd, (size_t) K * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dsoff, ho, (size_t) K * 8, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(dsl, slimbs, (size_t) o * 8, hipMemcpyHostToDevice) != hipSuccess)) rc = SLIP_HIP_DEVICE_ERROR;
    rescale_drop(f);
    if (!rc) rc = dev_alloc(&f->rspiv, K);
    if (!rc) rc = rescale_one(f, 1, f->hs.Lnz, f->hs.Lnl, habs, dslen, dsoff, dsl, stream);
    if (!rc) rc = rescale_one(f, 0, f->hs.Unz, f->hs.Unl, habs, dslen, dsoff, dsl, stream);
    if (!rc) {
        /* exact limb totals of the rescaled factors (download sizes its arrays from them) */
        SlipEnt *he = (SlipEnt *) malloc((size_t)(f->hs.Lnz > f->hs.Unz ? f->hs.Lnz : f->hs.Unz) * sizeof(SlipEnt));
        if (!he) rc = SLIP_HIP_OUT_OF_MEMORY;
        for (int pass = 0; pass < 2 && !rc; pass++) {
            const int64_t nz = pass == 0 ? f->hs.Lnz : f->hs.Unz;
            if (hipMemcpy(he, pass == 0 ? f->rsLe : f->rsUe, (size_t) nz * sizeof(SlipEnt), hipMemcpyDeviceToHost) != hipSuccess) { rc = SLIP_HIP_DEVICE_ERROR; break; }
            int64_t tot = 0;
            for (int64_t e = 0; e < nz; e++) { const int32_t d = he[e].len < 0 ? -he[e].len : he[e].len; tot += (d + 1) >> 1; }
            if (pass == 0) f->rsLexact = tot; else f->rsUexact = tot;
        }
        free(he);
        if (!rc) f->rescaled = 1;
    }
    if (rc) rescale_drop(f);
    dev_free(dslen); dev_free(dsoff); dev_free(dsl);
    free(hd); free(habs); free(ho);
    return rc;
}

/* diagnostic: per-phase shader cycles of the last run (zeros unless built with -DSLIP_PROFILE_PHASES) */
extern "C" int slip_hip_factor_phase_cycles(const slip_hip_factor *f, unsigned long long *out24)
{
    if (!f || !out24) return SLIP_HIP_INCORRECT_INPUT;
    for (int i = 0; i < 24; i++) out24[i] = f->hs.prof[i];
    return SLIP_HIP_OK;
}

/* diagnostic builds: wall-clock phase stamps of the heavy columns (more than 400 rows), 64 records of 32 words
 * (slot s = time of SLIP_STAMP(s); 29 = chain start, 30 = rows, 31 = column) */
extern "C" int slip_hip_factor_heavy_trace(const slip_hip_factor *f, int32_t *out2048)
{
    if (!f || !out2048) return SLIP_HIP_INCORRECT_INPUT;
    CK(hipMemcpy(out2048, f->P.dbg + 24 * (int64_t) f->n, 2048 * 4, hipMemcpyDeviceToHost));
    return SLIP_HIP_OK;
}

/* diagnostic builds: the per-column trace (8 words per column: commit-chain cycles, early flag, candidates computed, rows,
 * cycles of the sweep after the last frontier wait, of the early pass, of the publish, worker) */
extern "C" int slip_hip_factor_column_trace(const slip_hip_factor *f, int32_t *out, int32_t ncols)
{
    if (!f || !out || ncols <= 0 || ncols > f->n) return SLIP_HIP_INCORRECT_INPUT;
    CK(hipMemcpy(out, f->P.dbg, (size_t) ncols * 8 * 4, hipMemcpyDeviceToHost));
    /* word 8: the 100 MHz chip clock when the column's worker saw its turn (kept behind the n trace records) */
    CK(hipMemcpy(out + 8 * (int64_t) ncols, f->P.dbg + 8 * (int64_t) f->n, (size_t) ncols * 4, hipMemcpyDeviceToHost));
    /* words 9..16: cycles of the sub-steps of the commit chain */
    CK(hipMemcpy(out + 9 * (int64_t) ncols, f->P.dbg + 9 * (int64_t) f->n, (size_t) ncols * 8 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(out + 17 * (int64_t) ncols, f->P.dbg + 17 * (int64_t) f->n, (size_t) ncols * 4, hipMemcpyDeviceToHost));      /* path flags */
    CK(hipMemcpy(out + 18 * (int64_t) ncols, f->P.dbg + 18 * (int64_t) f->n, (size_t) ncols * 6 * 4, hipMemcpyDeviceToHost));  /* wall-clock time line, 6 words per column */

    return SLIP_HIP_OK;
}

/* diagnostic: raw words of the device debug area (placement of the workgroups of the last launch at 24 n + 2048: cu | sh | se | xcc) */
extern "C" int slip_hip_factor_debug_words(const slip_hip_factor *f, int64_t offset, int32_t count, int32_t *out)
{
    if (!f || !out || offset < 0 || count <= 0 || offset + count > 24 * (int64_t) f->n + 4096) return SLIP_HIP_INCORRECT_INPUT;
    CK(hipMemcpy(out, f->P.dbg + offset, (size_t) count * 4, hipMemcpyDeviceToHost));
    return SLIP_HIP_OK;
}

extern "C" int slip_hip_factor_info(const slip_hip_factor *f, slip_hip_info *o)
{
    if (!f || !o) return SLIP_HIP_INCORRECT_INPUT;
    const SlipState *h = &f->hs;
    o->n = f->n; o->K = h->F; o->status = f->last_status; o->window_end = f->window_end;
    o->lnz = h->Lnz; o->unz = h->Unz; o->l_limbs = f->rescaled ? f->rsLexact : h->Lnl_exact; o->u_limbs = f->rescaled ? f->rsUexact : h->Unl_exact;
    o->n_upd = (int64_t) h->c_upd; o->b_read = (int64_t) h->c_read; o->b_write = (int64_t) h->c_write;
    o->n_src = (int64_t) h->c_src; o->l_streamed = (int64_t) h->c_streamed;
    o->max_limbs = (int64_t)((h->c_maxdig + 1) / 2);
    o->kernel_ms = f->kernel_ms; o->launches = f->launches; o->xcap_digits = f->P.xcap;
    o->limb_macs = (int64_t) h->c_macs; o->workers = f->nworkers; o->waves = f->waves; o->lds_bytes = f->lds_words * 4;
    o->short_commits = (int32_t)(uint32_t) h->c_short; o->committer_commits = (int32_t)(h->c_short >> 32);
    o->farm_jobs = (int32_t)(uint32_t) h->c_farm; o->farm_items = (int32_t)(h->c_farm >> 32);
    o->engine_commits = (int32_t)(uint32_t) h->c_eng; o->engine_sources = (int32_t)(h->c_eng >> 32);
    o->retractions = (int32_t)(uint32_t) h->c_retract; o->reexports = (int32_t)(h->c_retract >> 32); o->pad2 = 0;
    return SLIP_HIP_OK;
}

/* entry records -> signed 64-bit limb counts, and the limbs gathered entry by entry into the back-to-back layout
 * of the ABI (slots of rows multiplied straight into the slab may leave unused limbs behind them) */
/* entries' limbs, scattered over a slab with gaps (an early commit reserves a column's region from bounds), packed back
 * to back on the device: one wavefront per entry, coalesced; the host then copies exactly the limbs it was promised */
#ifndef SLIP_EMULATE
__global__ void __launch_bounds__(256) slip_gather_kernel(const SlipEnt *ent, const uint64_t *limbs, const int64_t *ooff, uint64_t *out, int64_t nz)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t) gridDim.x * blockDim.x) >> 6;
    for (int64_t t = wave; t < nz; t += nwaves) {
        const int64_t l = ooff[t + 1] - ooff[t];
        const uint64_t *src = limbs + ent[t].off; uint64_t *dst = out + ooff[t];
        for (int64_t c = lane; c < l; c += 64) dst[c] = src[c];
    }
}
#endif

/* `expect`: the limbs the device counted for these entries (SlipState.Lnl_exact / Unl_exact, or the rescaled totals);
 * `cap`: what the caller's limbs_out can hold.  Nothing is written to limbs_out unless the entry records are consistent
 * with both and every entry lies inside the slab (nl_alloc limbs): device state never sizes a write into host memory. */
static int fetch_factor(int32_t *len_out, uint64_t *limbs_out, const SlipEnt *dev_ent, const uint64_t *dev_limbs, int64_t nz, int64_t nl_alloc,
                        int64_t expect, int64_t cap, int64_t *written)
{
    if (written) *written = 0;
    if (nz <= 0) return 0;
    SlipEnt *ent = (SlipEnt *) malloc((size_t) nz * sizeof(SlipEnt));
    if (!ent) return SLIP_HIP_OUT_OF_MEMORY;
    if (hipMemcpy(ent, dev_ent, (size_t) nz * sizeof(SlipEnt), hipMemcpyDeviceToHost) != hipSuccess) { free(ent); return SLIP_HIP_DEVICE_ERROR; }
    /* the records first: lengths, offsets and the total against what the device counted */
    int64_t total = 0;
    for (int64_t t = 0; t < nz; t++) {
        const int64_t d = ent[t].len < 0 ? -(int64_t) ent[t].len : ent[t].len, l = (d + 1) >> 1;
        if (d > (int64_t) 1 << 30 || ent[t].off < 0 || (l > 0 && ent[t].off + l > nl_alloc)) {
            fprintf(stderr, "slip_hip: entry record %lld is inconsistent (off %lld, %lld digits, slab %lld limbs)\n", (long long) t, (long long) ent[t].off, (long long) d, (long long) nl_alloc);
            free(ent); return SLIP_HIP_DEVICE_ERROR;
        }
        total += l;
    }
    if (total != expect) {
        fprintf(stderr, "slip_hip: the entry records hold %lld limbs, the device counted %lld\n", (long long) total, (long long) expect);
        free(ent); return SLIP_HIP_DEVICE_ERROR;
    }
    if (limbs_out && total > cap) { free(ent); return SLIP_HIP_INCORRECT_INPUT; }
    if (len_out)
        for (int64_t t = 0; t < nz; t++) {
            int32_t d = ent[t].len, a = d < 0 ? -d : d;
            a = (a + 1) >> 1;
            len_out[t] = d < 0 ? -a : a;
        }
    int rc = 0;
    if (limbs_out && total > 0) {
#ifndef SLIP_EMULATE
        /* one pass on the device, then one copy of exactly the packed limbs */
        int64_t *ooff = (int64_t *) malloc(((size_t) nz + 1) * 8), *d_ooff = NULL; uint64_t *d_out = NULL;
        if (!ooff) { free(ent); return SLIP_HIP_OUT_OF_MEMORY; }
        ooff[0] = 0;
        for (int64_t t = 0; t < nz; t++) { const int32_t d = ent[t].len; ooff[t + 1] = ooff[t] + (((d < 0 ? -d : d) + 1) >> 1); }
        if (dev_alloc(&d_ooff, nz + 1) || dev_alloc(&d_out, total)) rc = SLIP_HIP_OUT_OF_MEMORY;
        else if (hipMemcpy(d_ooff, ooff, ((size_t) nz + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
        else {
            int64_t blocks = (nz + 3) / 4; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(slip_gather_kernel, dim3((unsigned) blocks), dim3(256), 0, 0, dev_ent, dev_limbs, d_ooff, d_out, nz);
            if (hipGetLastError() != hipSuccess || hipMemcpy(limbs_out, d_out, (size_t) total * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
        }
        if (d_ooff) dev_free(d_ooff);
        if (d_out) dev_free(d_out);
        free(ooff);
#else
        uint64_t *raw = (uint64_t *) malloc((size_t)(nl_alloc > 0 ? nl_alloc : 1) * 8);
        if (!raw) rc = SLIP_HIP_OUT_OF_MEMORY;
        else if (hipMemcpy(raw, dev_limbs, (size_t) nl_alloc * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
        else {
            int64_t o = 0;
            for (int64_t t = 0; t < nz; t++) {
                const int32_t d = ent[t].len;
                const int64_t l = ((d < 0 ? -d : d) + 1) >> 1;
                if (l) memcpy(limbs_out + o, raw + ent[t].off, (size_t) l * 8);
                o += l;
            }
        }
        free(raw);
#endif
    }
    if (!rc && written) *written = total;
    free(ent);
    return rc;
}

extern "C" int slip_hip_factor_download(const slip_hip_factor *f,
                                        int64_t *Lp, int32_t *Li, int32_t *Llen, uint64_t *Llimbs, int64_t *L_limbs_inout,
                                        int64_t *Up, int32_t *Ui, int32_t *Ulen, uint64_t *Ulimbs, int64_t *U_limbs_inout,
                                        int32_t *rholen, uint64_t *rholimbs, int64_t *rho_limbs_inout,
                                        int32_t *pinv)
{
    if (!f) return SLIP_HIP_INCORRECT_INPUT;
    /* a limb array without its capacity cannot be written safely */
    if ((Llimbs && !L_limbs_inout) || (Ulimbs && !U_limbs_inout) || (rholimbs && !rho_limbs_inout)) return SLIP_HIP_INCORRECT_INPUT;
    /* after a device error the columns between the ready frontier and the commit frontier may be half written: nothing is handed out */
    if (f->last_status == SLIP_HIP_DEVICE_ERROR) return SLIP_HIP_DEVICE_ERROR;
    const SlipParams *P = &f->P;
    const SlipState *h = &f->hs;
    const int32_t K = h->F;
    int e;
    if (Lp) CK(hipMemcpy(Lp, P->Lp, ((size_t) K + 1) * 8, hipMemcpyDeviceToHost));
    if (Up) CK(hipMemcpy(Up, P->Up, ((size_t) K + 1) * 8, hipMemcpyDeviceToHost));
    if (Li && h->Lnz) CK(hipMemcpy(Li, P->Li, (size_t) h->Lnz * 4, hipMemcpyDeviceToHost));
    if (Ui && h->Unz) CK(hipMemcpy(Ui, P->Ui, (size_t) h->Unz * 4, hipMemcpyDeviceToHost));
    {
        int64_t wl = 0, wu = 0;
        if ((Llen || Llimbs) && (e = fetch_factor(Llen, Llimbs, f->rescaled ? f->rsLe : P->Le, f->rescaled ? f->rsLl : P->Llimbs, h->Lnz, f->rescaled ? f->rsLnl : P->Lcap_nl,
                                                  f->rescaled ? f->rsLexact : h->Lnl_exact, L_limbs_inout ? *L_limbs_inout : 0, &wl))) return e;
        if ((Ulen || Ulimbs) && (e = fetch_factor(Ulen, Ulimbs, f->rescaled ? f->rsUe : P->Ue, f->rescaled ? f->rsUl : P->Ulimbs, h->Unz, f->rescaled ? f->rsUnl : P->Ucap_nl,
                                                  f->rescaled ? f->rsUexact : h->Unl_exact, U_limbs_inout ? *U_limbs_inout : 0, &wu))) return e;
        if (L_limbs_inout) *L_limbs_inout = wl;
        if (U_limbs_inout) *U_limbs_inout = wu;
    }
    if (pinv) CK(hipMemcpy(pinv, P->pinv.p_, (size_t) f->n * 4, hipMemcpyDeviceToHost));
    if ((rholen || rholimbs) && K > 0) {
        /* the pivots live in the L slab: gather them through the pivot records (a rescaled copy: through the pivot entries) */
        SlipPiv *pr = (SlipPiv *) malloc((size_t) K * sizeof(SlipPiv));
        if (!pr) return SLIP_HIP_OUT_OF_MEMORY;
        if (hipMemcpy(pr, P->piv.p_, (size_t) K * sizeof(SlipPiv), hipMemcpyDeviceToHost) != hipSuccess) { free(pr); return SLIP_HIP_DEVICE_ERROR; }
        const uint64_t *Lsrc = P->Llimbs;
        if (f->rescaled) {
            int64_t *pi = (int64_t *) malloc((size_t) K * 8);
            SlipEnt pe;
            if (!pi) { free(pr); return SLIP_HIP_OUT_OF_MEMORY; }
            int bad = hipMemcpy(pi, f->rspiv, (size_t) K * 8, hipMemcpyDeviceToHost) != hipSuccess;
            for (int32_t k = 0; k < K && !bad; k++) {
                if (hipMemcpy(&pe, f->rsLe + pi[k], sizeof pe, hipMemcpyDeviceToHost) != hipSuccess) bad = 1;
                pr[k].off = pe.off; pr[k].len = pe.len;
            }
            free(pi);
            if (bad) { free(pr); return SLIP_HIP_DEVICE_ERROR; }
            Lsrc = f->rsLl;
        }
        int64_t o = 0, capl = rho_limbs_inout ? *rho_limbs_inout : 0;
        int rc = 0;
        for (int32_t k = 0; k < K && !rc; k++) {
            int32_t d = pr[k].len, l = ((d < 0 ? -d : d) + 1) >> 1;
            if (rholen) rholen[k] = d < 0 ? -l : l;
            if (pr[k].off < 0 || pr[k].off + l > (f->rescaled ? f->rsLnl : P->Lcap_nl)) { rc = SLIP_HIP_DEVICE_ERROR; break; }
            if (rholimbs) {
                if (o + l > capl) { rc = SLIP_HIP_INCORRECT_INPUT; break; }
                if (hipMemcpy(rholimbs + o, Lsrc + pr[k].off, (size_t) l * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
            }
            o += l;
        }
        if (rho_limbs_inout) *rho_limbs_inout = o;
        free(pr);
        if (rc) return rc;
    } else if (rho_limbs_inout) *rho_limbs_inout = 0;
    return SLIP_HIP_OK;
}

extern "C" int slip_hip_wave_op_test(int32_t op, int32_t nops, int32_t la, int32_t lb, int32_t W,
                                     const uint32_t *a, const uint32_t *b, uint32_t *out)
{
    if (nops <= 0 || la <= 0 || W <= 0 || !a || !out || (op != 3 && op != 13 && op != 14 && (!b || lb <= 0))) return SLIP_HIP_INCORRECT_INPUT;
    if (op >= 10 && W > 256) return SLIP_HIP_INCORRECT_INPUT;
    if (slip_hip_device_count() <= 0) return SLIP_HIP_DEVICE_ERROR;
    uint32_t *da = NULL, *db = NULL, *dout = NULL, *ds = NULL;
    const int64_t lbb = (lb > 0 && op != 14) ? lb : 1;
    if (dev_alloc(&da, (int64_t) nops * la) || dev_alloc(&db, (int64_t) nops * lbb) ||
        dev_alloc(&dout, (int64_t) nops * W) || dev_alloc(&ds, (int64_t) nops * 2 * (W + 1))) return SLIP_HIP_OUT_OF_MEMORY;
    CK(hipMemcpy(da, a, (size_t) nops * la * 4, hipMemcpyHostToDevice));
    if (b && lb > 0 && op != 14) CK(hipMemcpy(db, b, (size_t) nops * lb * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dout, 0, (size_t) nops * W * 4));
#ifndef SLIP_EMULATE
    hipLaunchKernelGGL(slip_wave_op_kernel, dim3(nops), dim3(64), 0, 0, op, la, lb, W, da, db, dout, ds);
    CK(hipGetLastError());
#else
    emu::launch(nops, 64, [=]() {
        const int blk = slip_block();
        const uint32_t *A = da + (int64_t) blk * la, *B = db + (int64_t) blk * lbb;
        uint32_t *O = dout + (int64_t) blk * W, *s0 = ds + (int64_t) blk * 2 * (W + 1), *s1 = s0 + W + 1;
        if (op == 0) wb_mul_lo(O, A, la, B, lb, W);
        else if (op == 1) wb_addsub(O, A, la, B, lb, W, 0);
        else if (op == 2) wb_addsub(O, A, la, B, lb, W, 1);
        else if (op == 3) wb_inv_extend(O, 0, W, A, la, s0, s1);
        else slip_reg_op_test(op, A, la, B, lb, W, O, s0);
    }, 256 * 1024, 1);
#endif
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout, (size_t) nops * W * 4, hipMemcpyDeviceToHost));
    dev_free(da); dev_free(db); dev_free(dout); dev_free(ds);
    return SLIP_HIP_OK;
}

#ifndef SLIP_EMULATE
/* development aid: average shader cycles of one wave-level primitive (see slip_wave_bench_kernel) */
extern "C" int slip_hip_wave_op_bench(int32_t op, int32_t la, int32_t lb, int32_t W, int32_t iters, int32_t nwaves,
                                      int32_t out_in_lds, unsigned long long *cycles_out)
{
    uint32_t *da = NULL, *db = NULL, *dout = NULL; unsigned long long *dc = NULL;
    if (dev_alloc(&da, (int64_t) nwaves * la) || dev_alloc(&db, (int64_t) nwaves * lb) ||
        dev_alloc(&dout, (int64_t) nwaves * 2 * (W + 2)) || dev_alloc(&dc, nwaves)) return SLIP_HIP_OUT_OF_MEMORY;
    CK(hipMemset(da, 0x5A, (size_t) nwaves * la * 4));
    CK(hipMemset(db, 0xC3, (size_t) nwaves * lb * 4));
    const size_t lds = (size_t) nwaves * 2 * (W + 2) * 4;
    CK(hipFuncSetAttribute((const void *) slip_wave_bench_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    hipLaunchKernelGGL(slip_wave_bench_kernel, dim3(1), dim3(64 * nwaves), lds, 0, op, la, lb, W, iters, out_in_lds, da, db, dout, dc);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(cycles_out, dc, (size_t) nwaves * 8, hipMemcpyDeviceToHost));
    dev_free(da); dev_free(db); dev_free(dout); dev_free(dc);
    return SLIP_HIP_OK;
}
#endif
