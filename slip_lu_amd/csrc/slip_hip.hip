/* slip_hip.hip -- libslip_hip.so: kernels' entry points and the host side of the
 * C ABI declared in include/slip_hip.h.
 *
 * Host responsibilities (the part of SLIP_LU/Source/SLIP_LU_factorize.c:58-211
 * that is not arithmetic): validate, upload A and q once, size the dense
 * scatter vector / L / U slabs in HBM, launch the column-loop kernel, and grow
 * a buffer and relaunch from the interrupted column when the kernel asks
 * (the reference doubles L/U at :200-211 and lets GMP realloc x).
 *
 * Built by hipcc for gfx950.  With -DSLIP_EMULATE (tests/emu, g++) the same
 * source runs the kernel lane-by-lane on the CPU for unit tests; that build is
 * never the product.
 */
#include "ref_lu_kernel.h"
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
#ifndef SLIP_EMULATE
extern "C" __global__ void __launch_bounds__(1024)
slip_factor_kernel(SlipDev *S)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t slip_lds[];
    slip_factor_columns(S, slip_lds);
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
    else wb_inv_extend(O, 0, W, A, la, s0, s1);
}
#else
static uint32_t slip_emu_lds[SLIP_LDS_MAX_WORDS + 1024];
#endif

/* ------------------------------------------------------------------ */
/* host state                                                          */
/* ------------------------------------------------------------------ */
struct slip_hip_factor {
    SlipDev h;            /* host mirror (device pointers inside) */
    SlipDev *d;           /* device copy the kernel works on      */
    int32_t n; int64_t annz, alimbs;
    int32_t waves, lds_words;
    int32_t last_status, window_end, launches;
    double kernel_ms;
    hipEvent_t ev0, ev1;
    /* owned device arrays that are not reachable through const pointers in h */
    int64_t *dAp; int32_t *dAi, *dAlen; int64_t *dAoff; uint64_t *dAlimbs; int32_t *dq;
};

#define CK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
        fprintf(stderr, "slip_hip: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
        return SLIP_HIP_DEVICE_ERROR; } } while (0)

template <class T> static int dev_alloc(T **p, int64_t count)
{
    void *q = NULL;
    if (hipMalloc(&q, (size_t)(count > 0 ? count : 1) * sizeof(T)) != hipSuccess) return SLIP_HIP_OUT_OF_MEMORY;
    *p = (T *) q;
    return 0;
}
template <class T> static int dev_grow(T **p, int64_t old_count, int64_t new_count)
{
    T *q = NULL;
    if (dev_alloc(&q, new_count)) return SLIP_HIP_OUT_OF_MEMORY;
    if (old_count > 0 && hipMemcpy(q, *p, (size_t) old_count * sizeof(T), hipMemcpyDeviceToDevice) != hipSuccess) return SLIP_HIP_DEVICE_ERROR;
    hipFree(*p);
    *p = q;
    return 0;
}

extern "C" const char *slip_hip_version(void) { return "slip_hip 0.1 (gfx950)"; }

extern "C" void slip_hip_default_options(slip_hip_options *o)
{
    /* SLIP_LU_internal.h:136-149: pivot = SLIP_TOL_SMALLEST, tol = 1 */
    o->pivot = 3; o->tol = 1.0; o->limb_cap = 0; o->waves = 0; o->lnz_hint = 0; o->unz_hint = 0;
}

extern "C" int slip_hip_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

extern "C" void slip_hip_free(void *p) { free(p); }

extern "C" int slip_hip_matgen(int32_t n, double density, int32_t bits, uint64_t seed,
                               int64_t **Ap, int32_t **Ai, int64_t **Ax)
{
    return slip_matgen_csc(n, density, bits, seed, Ap, Ai, Ax) ? SLIP_HIP_OUT_OF_MEMORY : SLIP_HIP_OK;
}

/* choose waves / LDS split for the current xcap */
static void plan_launch(slip_hip_factor *f)
{
    SlipDev *h = &f->h;
    h->wcap = h->xcap + 8;
    h->bm_words = (h->n + 31) / 32;
    h->bitmap_in_lds = h->bm_words <= SLIP_BITMAP_LDS_MAX_WORDS;
    int fixed = SLIP_LDS_BITMAP + (h->bitmap_in_lds ? h->bm_words : 0);
    int nw = f->waves;
    while (nw > 4 && fixed + (int64_t) nw * 3 * h->wcap > SLIP_LDS_MAX_WORDS) nw /= 2;
    h->scratch_in_lds = fixed + (int64_t) nw * 3 * h->wcap <= SLIP_LDS_MAX_WORDS;
    f->waves = nw;
    f->lds_words = fixed + (h->scratch_in_lds ? nw * 3 * h->wcap : 0);
}

static int alloc_x(slip_hip_factor *f, int32_t xcap)
{
    SlipDev *h = &f->h;
    if (h->xd) hipFree(h->xd);
    if (h->invd) hipFree(h->invd);
    if (h->gscratch) hipFree(h->gscratch);
    h->xd = NULL; h->invd = NULL; h->gscratch = NULL;
    xcap = (xcap + 1) & ~1;
    h->xcap = xcap; h->invcap = xcap + 8;
    if (dev_alloc(&h->xd, (int64_t) h->n * xcap)) return SLIP_HIP_OUT_OF_MEMORY;
    if (dev_alloc(&h->invd, (int64_t) h->n * h->invcap)) return SLIP_HIP_OUT_OF_MEMORY;
    if (hipMemset(h->invlen, 0, (size_t) h->n * 4) != hipSuccess) return SLIP_HIP_DEVICE_ERROR;
    plan_launch(f);
    if (dev_alloc(&h->gscratch, (int64_t) 16 * 3 * h->wcap)) return SLIP_HIP_OUT_OF_MEMORY;
    return 0;
}

extern "C" int slip_hip_factor_reset(slip_hip_factor *f)
{
    if (!f) return SLIP_HIP_INCORRECT_INPUT;
    SlipDev *h = &f->h;
    const int32_t n = f->n;
    int32_t *id = (int32_t *) malloc((size_t) n * 4);
    if (!id) return SLIP_HIP_OUT_OF_MEMORY;
    for (int32_t i = 0; i < n; i++) id[i] = i;
    CK(hipMemcpy(h->pinv, id, (size_t) n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(h->row_perm, id, (size_t) n * 4, hipMemcpyHostToDevice));
    free(id);
    CK(hipMemset(h->h, 0xFF, (size_t) n * 4));
    CK(hipMemset(h->invlen, 0, (size_t) n * 4));
    CK(hipMemset(h->Lp, 0, 8));
    CK(hipMemset(h->Up, 0, 8));
    h->Lnz = h->Lnl = h->Unz = h->Unl = 0;
    h->k_next = 0; h->status = 0; h->status_k = 0;
    h->c_upd = h->c_read = h->c_write = h->c_src = h->c_streamed = h->c_maxdig = 0;
    memset(h->prof, 0, sizeof h->prof);
    f->last_status = 0; f->window_end = 0; f->kernel_ms = 0; f->launches = 0;
    return SLIP_HIP_OK;
}

extern "C" void slip_hip_factor_destroy(slip_hip_factor *f)
{
    if (!f) return;
    SlipDev *h = &f->h;
    hipFree(f->dAp); hipFree(f->dAi); hipFree(f->dAlen); hipFree(f->dAoff); hipFree(f->dAlimbs); hipFree(f->dq);
    hipFree(h->pinv); hipFree(h->row_perm); hipFree(h->h); hipFree(h->xd); hipFree(h->xlen);
    hipFree(h->rho_off); hipFree(h->rho_len); hipFree(h->rho_bits); hipFree(h->rho_ctz);
    hipFree(h->invd); hipFree(h->invlen);
    hipFree(h->Lp); hipFree(h->Li); hipFree(h->Llen); hipFree(h->Loff); hipFree(h->Llimbs);
    hipFree(h->Up); hipFree(h->Ui); hipFree(h->Ulen); hipFree(h->Uoff); hipFree(h->Ulimbs);
    hipFree(h->pat); hipFree(h->gscratch); hipFree(h->gbitmap);
    hipFree(f->d);
    if (f->ev0) hipEventDestroy(f->ev0);
    if (f->ev1) hipEventDestroy(f->ev1);
    free(f);
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
    SlipDev *h = &f->h;
    f->n = n; f->annz = onz; f->alimbs = ol;
    f->waves = opt.waves > 0 ? opt.waves : 16;
    if (f->waves > 16) f->waves = 16;
    h->n = n; h->pivot_scheme = opt.pivot; h->limb_cap = opt.limb_cap;
    if (!(opt.tol > 0)) { h->tol_mode = 0; h->tol_m = 0; h->tol_e = 0; }
    else {
        int e; double fr = frexp(opt.tol, &e);            /* mpq_set_d takes the double exactly */
        h->tol_mode = 1; h->tol_m = (uint64_t) ldexp(fr, 53); h->tol_e = e - 53;
    }
    int rc = 0;
#define A_(call) do { if (!rc) rc = (call); } while (0)
    A_(dev_alloc(&f->dAp, (int64_t) n + 1)); A_(dev_alloc(&f->dAi, onz)); A_(dev_alloc(&f->dAlen, onz));
    A_(dev_alloc(&f->dAoff, onz)); A_(dev_alloc(&f->dAlimbs, ol)); A_(dev_alloc(&f->dq, n));
    A_(dev_alloc(&h->pinv, n)); A_(dev_alloc(&h->row_perm, n)); A_(dev_alloc(&h->h, n)); A_(dev_alloc(&h->xlen, n));
    A_(dev_alloc(&h->rho_off, n)); A_(dev_alloc(&h->rho_len, n)); A_(dev_alloc(&h->rho_bits, n)); A_(dev_alloc(&h->rho_ctz, n));
    A_(dev_alloc(&h->invlen, n)); A_(dev_alloc(&h->pat, n));
    A_(dev_alloc(&h->gbitmap, (int64_t)(n + 31) / 32 + 64));
    /* initial sizes: S->lnz/unz only size the first allocation in the reference too */
    h->Lcap_nz = opt.lnz_hint > 0 ? opt.lnz_hint : 4 * onz + n;
    h->Ucap_nz = opt.unz_hint > 0 ? opt.unz_hint : 4 * onz + n;
    if (h->Lcap_nz < n) h->Lcap_nz += n;
    if (h->Ucap_nz < n) h->Ucap_nz += n;
    const int32_t cap_digits = opt.limb_cap > 0 ? 2 * opt.limb_cap + 8 : 0;
    int32_t xcap0 = cap_digits > 0 ? cap_digits : (2 * maxdig + 8 > 16 ? 2 * maxdig + 8 : 16);
    h->Lcap_nl = h->Lcap_nz * (int64_t)(opt.limb_cap > 0 ? (opt.limb_cap + 1) / 2 + 1 : 2);
    h->Ucap_nl = h->Ucap_nz * 2;
    A_(dev_alloc(&h->Lp, (int64_t) n + 1)); A_(dev_alloc(&h->Li, h->Lcap_nz)); A_(dev_alloc(&h->Llen, h->Lcap_nz));
    A_(dev_alloc(&h->Loff, h->Lcap_nz)); A_(dev_alloc(&h->Llimbs, h->Lcap_nl));
    A_(dev_alloc(&h->Up, (int64_t) n + 1)); A_(dev_alloc(&h->Ui, h->Ucap_nz)); A_(dev_alloc(&h->Ulen, h->Ucap_nz));
    A_(dev_alloc(&h->Uoff, h->Ucap_nz)); A_(dev_alloc(&h->Ulimbs, h->Ucap_nl));
    A_(dev_alloc(&f->d, 1));
    if (!rc) rc = alloc_x(f, xcap0);
#undef A_
    if (!rc) {
        if (hipMemcpy(f->dAp, hAp, ((size_t) n + 1) * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAi, hAi, (size_t) onz * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAlen, hAlen, (size_t) onz * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAoff, hAoff, (size_t) onz * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dAlimbs, hAlimbs, (size_t) ol * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(f->dq, q, (size_t) n * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(h->xlen, 0, (size_t) n * 4) != hipSuccess)
            rc = SLIP_HIP_DEVICE_ERROR;
    }
    free(hAp); free(hAi); free(hAlen); free(hAoff); free(hAlimbs);
    h->Ap = f->dAp; h->Ai = f->dAi; h->Alen = f->dAlen; h->Aoff = f->dAoff; h->Alimbs = f->dAlimbs; h->q = f->dq;
    if (!rc && (hipEventCreate(&f->ev0) != hipSuccess || hipEventCreate(&f->ev1) != hipSuccess)) rc = SLIP_HIP_DEVICE_ERROR;
    if (!rc) rc = slip_hip_factor_reset(f);
    if (rc) { slip_hip_factor_destroy(f); return rc; }
    *out = f;
    return SLIP_HIP_OK;
}

static int launch_columns(slip_hip_factor *f, hipStream_t stream)
{
    CK(hipMemcpyAsync(f->d, &f->h, sizeof(SlipDev), hipMemcpyHostToDevice, stream));
    CK(hipEventRecord(f->ev0, stream));
#ifndef SLIP_EMULATE
    const size_t lds_bytes = (size_t) f->lds_words * 4;
    CK(hipFuncSetAttribute((const void *) slip_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds_bytes));
    hipLaunchKernelGGL(slip_factor_kernel, dim3(1), dim3(64 * f->waves), lds_bytes, stream, f->d);
    CK(hipGetLastError());
#else
    SlipDev *d = f->d;
    emu::launch(1, 64 * f->waves, [d]() { slip_factor_columns(d, slip_emu_lds); });
#endif
    CK(hipEventRecord(f->ev1, stream));
    CK(hipMemcpyAsync(&f->h, f->d, sizeof(SlipDev), hipMemcpyDeviceToHost, stream));
    CK(hipStreamSynchronize(stream));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, f->ev0, f->ev1));
    f->kernel_ms += ms;
    f->launches++;
    return 0;
}

extern "C" int slip_hip_factor_run(slip_hip_factor *f, int32_t kmax, void *stream_v)
{
    if (!f) return SLIP_HIP_INCORRECT_INPUT;
    hipStream_t stream = (hipStream_t) stream_v;
    SlipDev *h = &f->h;
    if (kmax <= 0 || kmax > f->n) kmax = f->n;
    f->kernel_ms = 0; f->launches = 0; f->window_end = 0;
    h->k_stop = kmax;
    int rc = SLIP_HIP_OK;
    while (h->k_next < kmax) {
        int e = launch_columns(f, stream);
        if (e) { rc = e; break; }
        if (h->status == SLIPDEV_OK) continue;
        if (h->status == SLIPDEV_SINGULAR) { rc = SLIP_HIP_SINGULAR; break; }
        if (h->status == SLIPDEV_WINDOW_END) { f->window_end = 1; break; }
        if (h->status == SLIPDEV_GROW_L) {
            /* the slab that ran out is the one to double (cf. slip_sparse_realloc.c) */
            int64_t nz = h->Lcap_nz * 2, nl = h->Lcap_nl * 2;
            if ((e = dev_grow(&h->Li, h->Lnz, nz)) || (e = dev_grow(&h->Llen, h->Lnz, nz)) ||
                (e = dev_grow(&h->Loff, h->Lnz, nz)) || (e = dev_grow(&h->Llimbs, h->Lnl, nl))) { rc = e; break; }
            h->Lcap_nz = nz; h->Lcap_nl = nl;
        } else if (h->status == SLIPDEV_GROW_U) {
            int64_t nz = h->Ucap_nz * 2, nl = h->Ucap_nl * 2;
            if ((e = dev_grow(&h->Ui, h->Unz, nz)) || (e = dev_grow(&h->Ulen, h->Unz, nz)) ||
                (e = dev_grow(&h->Uoff, h->Unz, nz)) || (e = dev_grow(&h->Ulimbs, h->Unl, nl))) { rc = e; break; }
            h->Ucap_nz = nz; h->Ucap_nl = nl;
        } else if (h->status == SLIPDEV_GROW_X) {
            if ((int64_t) h->xcap * 2 > (1 << 28)) { rc = SLIP_HIP_OUT_OF_MEMORY; break; }
            if ((e = alloc_x(f, h->xcap * 2))) { rc = e; break; }
        } else { rc = SLIP_HIP_DEVICE_ERROR; break; }
    }
    f->last_status = rc;
    return rc;
}

/* diagnostic: per-phase shader cycles of the last run (zeros unless built with -DSLIP_PROFILE_PHASES) */
extern "C" int slip_hip_factor_phase_cycles(const slip_hip_factor *f, unsigned long long *out12)
{
    if (!f || !out12) return SLIP_HIP_INCORRECT_INPUT;
    for (int i = 0; i < 12; i++) out12[i] = f->h.prof[i];
    return SLIP_HIP_OK;
}

extern "C" int slip_hip_factor_info(const slip_hip_factor *f, slip_hip_info *o)
{
    if (!f || !o) return SLIP_HIP_INCORRECT_INPUT;
    const SlipDev *h = &f->h;
    o->n = f->n; o->K = h->k_next; o->status = f->last_status; o->window_end = f->window_end;
    o->lnz = h->Lnz; o->unz = h->Unz; o->l_limbs = h->Lnl; o->u_limbs = h->Unl;
    o->n_upd = (int64_t) h->c_upd; o->b_read = (int64_t) h->c_read; o->b_write = (int64_t) h->c_write;
    o->n_src = (int64_t) h->c_src; o->l_streamed = (int64_t) h->c_streamed;
    o->max_limbs = (int64_t)((h->c_maxdig + 1) / 2);
    o->kernel_ms = f->kernel_ms; o->launches = f->launches; o->xcap_digits = h->xcap;
    return SLIP_HIP_OK;
}

static int fetch_lens(int32_t *dst, const int32_t *dev, int64_t cnt)
{
    if (hipMemcpy(dst, dev, (size_t) cnt * 4, hipMemcpyDeviceToHost) != hipSuccess) return SLIP_HIP_DEVICE_ERROR;
    for (int64_t t = 0; t < cnt; t++) {             /* digits -> 64-bit limbs, sign kept */
        int32_t d = dst[t], a = d < 0 ? -d : d;
        a = (a + 1) >> 1;
        dst[t] = d < 0 ? -a : a;
    }
    return 0;
}

extern "C" int slip_hip_factor_download(const slip_hip_factor *f,
                                        int64_t *Lp, int32_t *Li, int32_t *Llen, uint64_t *Llimbs,
                                        int64_t *Up, int32_t *Ui, int32_t *Ulen, uint64_t *Ulimbs,
                                        int32_t *rholen, uint64_t *rholimbs, int64_t *rho_limbs_inout,
                                        int32_t *pinv)
{
    if (!f) return SLIP_HIP_INCORRECT_INPUT;
    const SlipDev *h = &f->h;
    const int32_t K = h->k_next;
    if (Lp) CK(hipMemcpy(Lp, h->Lp, ((size_t) K + 1) * 8, hipMemcpyDeviceToHost));
    if (Up) CK(hipMemcpy(Up, h->Up, ((size_t) K + 1) * 8, hipMemcpyDeviceToHost));
    if (Li) CK(hipMemcpy(Li, h->Li, (size_t) h->Lnz * 4, hipMemcpyDeviceToHost));
    if (Ui) CK(hipMemcpy(Ui, h->Ui, (size_t) h->Unz * 4, hipMemcpyDeviceToHost));
    if (Llen && fetch_lens(Llen, h->Llen, h->Lnz)) return SLIP_HIP_DEVICE_ERROR;
    if (Ulen && fetch_lens(Ulen, h->Ulen, h->Unz)) return SLIP_HIP_DEVICE_ERROR;
    if (Llimbs) CK(hipMemcpy(Llimbs, h->Llimbs, (size_t) h->Lnl * 8, hipMemcpyDeviceToHost));
    if (Ulimbs) CK(hipMemcpy(Ulimbs, h->Ulimbs, (size_t) h->Unl * 8, hipMemcpyDeviceToHost));
    if (pinv) CK(hipMemcpy(pinv, h->pinv, (size_t) f->n * 4, hipMemcpyDeviceToHost));
    if (rholen || rholimbs) {
        /* the pivots live in the L slab: gather them on the host side */
        int32_t *rl = (int32_t *) malloc((size_t)(K > 0 ? K : 1) * 4);
        int64_t *ro = (int64_t *) malloc((size_t)(K > 0 ? K : 1) * 8);
        if (!rl || !ro) { free(rl); free(ro); return SLIP_HIP_OUT_OF_MEMORY; }
        if (fetch_lens(rl, h->rho_len, K) ||
            hipMemcpy(ro, h->rho_off, (size_t) K * 8, hipMemcpyDeviceToHost) != hipSuccess) { free(rl); free(ro); return SLIP_HIP_DEVICE_ERROR; }
        int64_t o = 0, capl = rho_limbs_inout ? *rho_limbs_inout : 0;
        int rc = 0;
        for (int32_t k = 0; k < K && !rc; k++) {
            int32_t l = rl[k] < 0 ? -rl[k] : rl[k];
            if (rholen) rholen[k] = rl[k];
            if (rholimbs) {
                if (o + l > capl) { rc = SLIP_HIP_INCORRECT_INPUT; break; }
                if (hipMemcpy(rholimbs + o, h->Llimbs + ro[k], (size_t) l * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = SLIP_HIP_DEVICE_ERROR;
            }
            o += l;
        }
        if (rho_limbs_inout) *rho_limbs_inout = o;
        free(rl); free(ro);
        if (rc) return rc;
    }
    return SLIP_HIP_OK;
}

extern "C" int slip_hip_wave_op_test(int32_t op, int32_t nops, int32_t la, int32_t lb, int32_t W,
                                     const uint32_t *a, const uint32_t *b, uint32_t *out)
{
    if (nops <= 0 || la <= 0 || W <= 0 || !a || !out || (op != 3 && (!b || lb <= 0))) return SLIP_HIP_INCORRECT_INPUT;
    if (slip_hip_device_count() <= 0) return SLIP_HIP_DEVICE_ERROR;
    uint32_t *da = NULL, *db = NULL, *dout = NULL, *ds = NULL;
    const int64_t lbb = lb > 0 ? lb : 1;
    if (dev_alloc(&da, (int64_t) nops * la) || dev_alloc(&db, (int64_t) nops * lbb) ||
        dev_alloc(&dout, (int64_t) nops * W) || dev_alloc(&ds, (int64_t) nops * 2 * (W + 1))) return SLIP_HIP_OUT_OF_MEMORY;
    CK(hipMemcpy(da, a, (size_t) nops * la * 4, hipMemcpyHostToDevice));
    if (b && lb > 0) CK(hipMemcpy(db, b, (size_t) nops * lb * 4, hipMemcpyHostToDevice));
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
        else wb_inv_extend(O, 0, W, A, la, s0, s1);
    });
#endif
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, dout, (size_t) nops * W * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dout); hipFree(ds);
    return SLIP_HIP_OK;
}
