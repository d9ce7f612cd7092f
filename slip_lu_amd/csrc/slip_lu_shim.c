/* slip_lu_shim.c -- libslip_lu_hip.so: the reference's SLIP_LU_factorize entry point
 * (SLIP_LU/Include/SLIP_LU.h:854-863) served by the HIP path.
 *
 * Host C only: converts the caller's mpz_t CSC into the limb-slab arrays of
 * include/slip_hip.h, runs the factorisation on the GPU, and materialises L, U,
 * rhos, pinv exactly as the reference hands them back
 * (SLIP_LU/Source/SLIP_LU_factorize.c:170-301):
 *   - L, U arrive from SLIP_create_sparse() (p = i = x = NULL) and leave owning
 *     p[n+1], i[nz], x[nz] with nzmax == nz (slip_sparse_collapse.c:28-32);
 *   - every x[p] is a genuine mpz_t initialised with mpz_init2(bits+2) and set
 *     through GMP, so SLIP_delete_sparse -> mpz_clear frees it with whatever
 *     allocator the caller installed (SLIP_initialize_expert.c:95);
 *   - L->i, U->i hold the PERMUTED row positions (the relabel of :293-301);
 *   - rhos[k] (already mpz_init-ed by the caller) is assigned with GMP;
 *   - errors: SLIP_INCORRECT_INPUT for missing arguments (:48-52), SLIP_SINGULAR
 *     (slip_get_smallest_pivot.c:93-96), SLIP_OUT_OF_MEMORY; a HIP failure maps
 *     to SLIP_OUT_OF_MEMORY (there is no CPU fallback).
 * Index arrays come from SLIP_calloc when the reference library is linked in
 * (so a MATLAB-aware allocator is honoured), from calloc otherwise.
 */
#include "../../include/SLIP_LU_hip.h"
#include "../../include/slip_hip.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdio.h>

extern void *SLIP_calloc(size_t n, size_t size) __attribute__((weak));

static void *shim_calloc(size_t n, size_t size)
{
    if (n == 0) n = 1;
    return SLIP_calloc ? SLIP_calloc(n, size) : calloc(n, size);
}

static void set_from_limbs(mpz_t z, int32_t slen, const uint64_t *limbs)
{
    int32_t l = slen < 0 ? -slen : slen;
    if (l == 0) { mpz_set_ui(z, 0); return; }
    mpz_import(z, (size_t) l, -1, 8, 0, 0, limbs);
    if (slen < 0) mpz_neg(z, z);
}

/* fill a SLIP_sparse from slab arrays; returns SLIP_OK or SLIP_OUT_OF_MEMORY */
static SLIP_info build_factor(SLIP_sparse *M, int32_t n, int64_t nz, const int64_t *p, const int32_t *ids,
                              const int32_t *len, const uint64_t *limbs, const int32_t *pinv)
{
    M->m = n; M->n = n; M->nz = (int32_t) nz; M->nzmax = (int32_t) nz;
    M->p = (int32_t *) shim_calloc((size_t) n + 1, sizeof(int32_t));
    M->i = (int32_t *) shim_calloc((size_t) nz, sizeof(int32_t));
    M->x = (mpz_t *) shim_calloc((size_t) nz, sizeof(mpz_t));
    if (!M->p || !M->i || !M->x) return SLIP_OUT_OF_MEMORY;
    for (int32_t k = 0; k <= n; k++) M->p[k] = (int32_t) p[k];
    int64_t o = 0;
    for (int64_t t = 0; t < nz; t++) {
        int32_t l = len[t] < 0 ? -len[t] : len[t];
        size_t bits = 0;
        if (l) bits = 64 * (size_t)(l - 1) + (64 - (size_t) __builtin_clzll(limbs[o + l - 1]));
        mpz_init2(M->x[t], (bits ? bits : 1) + 2);           /* SLIP_LU_factorize.c:239-241 */
        set_from_limbs(M->x[t], len[t], limbs + o);
        M->i[t] = pinv[ids[t]];                              /* :293-301 */
        o += l;
    }
    return SLIP_OK;
}

SLIP_info SLIP_hip_LU_factorize(SLIP_sparse *L, SLIP_sparse *U, SLIP_sparse *A, SLIP_LU_analysis *S,
                                mpz_t *rhos, int32_t *pinv, SLIP_options *option)
{
    if (!A || !L || !U || !S || !rhos || !pinv || !option || !A->p || !A->x || !A->i || !S->q)
        return SLIP_INCORRECT_INPUT;
    const int32_t n = A->n;
    if (n <= 0 || A->p[n] < 1) return SLIP_INCORRECT_INPUT;
    const int64_t annz = A->p[n];
    SLIP_info ret = SLIP_OUT_OF_MEMORY;
    slip_hip_factor *f = NULL;
    int64_t *Ap = NULL, *Lp = NULL, *Up = NULL;
    int32_t *Alen = NULL, *Li = NULL, *Ui = NULL, *Llen = NULL, *Ulen = NULL, *rholen = NULL;
    uint64_t *Alimbs = NULL, *Llimbs = NULL, *Ulimbs = NULL, *rholimbs = NULL;
    slip_hip_info info;
    slip_hip_options opt;
    int rc;

    /* SLIP_HIP_SHIM_TIMING=1: where the call spends its time, one line on stderr (conversions are one pass each way) */
    struct timespec t0_, t1_, t2_, t3_, t4_;
    clock_gettime(CLOCK_MONOTONIC, &t0_);
    /* ---- A: mpz_t -> limb slab ---- */
    int64_t nl = 0;
    for (int64_t t = 0; t < annz; t++) nl += (int64_t) mpz_size(A->x[t]);
    Ap = (int64_t *) malloc(((size_t) n + 1) * 8);
    Alen = (int32_t *) malloc((size_t) annz * 4);
    Alimbs = (uint64_t *) malloc((size_t)(nl ? nl : 1) * 8);
    if (!Ap || !Alen || !Alimbs) goto done;
    for (int32_t j = 0; j <= n; j++) Ap[j] = A->p[j];
    {
        int64_t o = 0;
        for (int64_t t = 0; t < annz; t++) {
            size_t l = mpz_size(A->x[t]);
            if (l) memcpy(Alimbs + o, mpz_limbs_read(A->x[t]), l * 8);
            Alen[t] = mpz_sgn(A->x[t]) < 0 ? -(int32_t) l : (int32_t) l;
            o += (int64_t) l;
        }
    }

    clock_gettime(CLOCK_MONOTONIC, &t1_);
    /* ---- factorise on the GPU ---- */
    slip_hip_default_options(&opt);
    opt.pivot = (int32_t) option->pivot;
    opt.tol = option->tol;
    opt.lnz_hint = S->lnz; opt.unz_hint = S->unz;
    rc = slip_hip_factor_create(&f, n, Ap, A->i, Alen, Alimbs, S->q, &opt);
    if (rc == SLIP_HIP_OK) rc = slip_hip_factor_run(f, 0, NULL);
    if (rc != SLIP_HIP_OK) {
        ret = rc == SLIP_HIP_SINGULAR ? SLIP_SINGULAR
            : rc == SLIP_HIP_INCORRECT_INPUT ? SLIP_INCORRECT_INPUT : SLIP_OUT_OF_MEMORY;
        goto done;
    }
    clock_gettime(CLOCK_MONOTONIC, &t2_);
    slip_hip_factor_info(f, &info);
    if (info.lnz > INT32_MAX || info.unz > INT32_MAX) goto done;      /* int32 CSC of the reference */
    Lp = (int64_t *) malloc(((size_t) n + 1) * 8); Up = (int64_t *) malloc(((size_t) n + 1) * 8);
    Li = (int32_t *) malloc((size_t) info.lnz * 4); Ui = (int32_t *) malloc((size_t) info.unz * 4);
    Llen = (int32_t *) malloc((size_t) info.lnz * 4); Ulen = (int32_t *) malloc((size_t) info.unz * 4);
    Llimbs = (uint64_t *) malloc((size_t)(info.l_limbs ? info.l_limbs : 1) * 8);
    Ulimbs = (uint64_t *) malloc((size_t)(info.u_limbs ? info.u_limbs : 1) * 8);
    rholen = (int32_t *) malloc((size_t) n * 4);
    rholimbs = (uint64_t *) malloc((size_t)(info.l_limbs ? info.l_limbs : 1) * 8);
    if (!Lp || !Up || !Li || !Ui || !Llen || !Ulen || !Llimbs || !Ulimbs || !rholen || !rholimbs) goto done;
    {
        /* every limb array goes with its capacity: the library checks the device's records against them before it writes */
        int64_t rcap = info.l_limbs ? info.l_limbs : 1, lcap = info.l_limbs ? info.l_limbs : 1, ucap = info.u_limbs ? info.u_limbs : 1;
        if (slip_hip_factor_download(f, Lp, Li, Llen, Llimbs, &lcap, Up, Ui, Ulen, Ulimbs, &ucap, rholen, rholimbs, &rcap, pinv) != SLIP_HIP_OK)
            goto done;
        if (lcap != info.l_limbs || ucap != info.u_limbs) goto done;
    }

    clock_gettime(CLOCK_MONOTONIC, &t3_);
    /* ---- hand the results over the way the reference does ---- */
    ret = build_factor(L, n, info.lnz, Lp, Li, Llen, Llimbs, pinv);
    if (ret == SLIP_OK) ret = build_factor(U, n, info.unz, Up, Ui, Ulen, Ulimbs, pinv);
    if (ret == SLIP_OK) {
        int64_t o = 0;
        for (int32_t k = 0; k < n; k++) {
            set_from_limbs(rhos[k], rholen[k], rholimbs + o);
            o += rholen[k] < 0 ? -rholen[k] : rholen[k];
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t4_);
    if (getenv("SLIP_HIP_SHIM_TIMING")) {
#define MS_(a, b) (((b).tv_sec - (a).tv_sec) * 1e3 + ((b).tv_nsec - (a).tv_nsec) * 1e-6)
        fprintf(stderr, "slip_lu_hip timing: A to slabs %.3f ms, upload+factorise %.3f ms (kernel %.3f ms), download %.3f ms, "
                        "L/U/rhos to mpz %.3f ms (%lld + %lld entries, %lld limbs)\n",
                MS_(t0_, t1_), MS_(t1_, t2_), info.kernel_ms, MS_(t2_, t3_), MS_(t3_, t4_),
                (long long) info.lnz, (long long) info.unz, (long long)(info.l_limbs + info.u_limbs));
#undef MS_
    }
done:
    if (f) slip_hip_factor_destroy(f);
    free(Ap); free(Alen); free(Alimbs); free(Lp); free(Up); free(Li); free(Ui); free(Llen); free(Ulen);
    free(Llimbs); free(Ulimbs); free(rholen); free(rholimbs);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * SLIP_LU_solve (SLIP_LU/Include/SLIP_LU.h:941-949; SLIP_LU/Source/SLIP_LU_solve.c:41-86) on the GPU:
 * the caller's L, U (permuted row positions, mpz_t values), pinv and the dense right-hand sides go to
 * the device as limb slabs, slip_hip_factor_solve runs the forward substitution, the scaling by
 * det = rhos[n-1] and the back substitution, and x[p][k] = numerator / det is formed with GMP exactly
 * as slip_array_div.c:36-49 does (mpq_set_num, mpq_div: canonical fractions).  x must arrive from
 * SLIP_create_mpq_mat (initialised), as in SLIP_solve_mpq.c; b is not modified.
 * ------------------------------------------------------------------------------------------------ */
static int sparse_to_slab(const SLIP_sparse *M, int32_t n, const int32_t *rowperm,
                          int64_t **p_out, int32_t **i_out, int32_t **len_out, uint64_t **limbs_out)
{
    const int64_t nz = M->p[n];
    int64_t nl = 0;
    *p_out = NULL; *i_out = NULL; *len_out = NULL; *limbs_out = NULL;
    if (nz < n) return 2;                                /* a factor holds at least its diagonal */
    for (int64_t t = 0; t < nz; t++) nl += (int64_t) mpz_size(M->x[t]);
    int64_t *p = (int64_t *) malloc(((size_t) n + 1) * 8);
    int32_t *ids = (int32_t *) malloc((size_t) nz * 4), *len = (int32_t *) malloc((size_t) nz * 4);
    uint64_t *limbs = (uint64_t *) malloc((size_t)(nl ? nl : 1) * 8);
    *p_out = p; *i_out = ids; *len_out = len; *limbs_out = limbs;
    if (!p || !ids || !len || !limbs) return 1;
    for (int32_t k = 0; k <= n; k++) p[k] = M->p[k];
    int64_t o = 0;
    for (int64_t t = 0; t < nz; t++) {
        if (M->i[t] < 0 || M->i[t] >= n) return 2;
        ids[t] = rowperm[M->i[t]];                       /* permuted position -> original row id */
        size_t l = mpz_size(M->x[t]);
        if (l) memcpy(limbs + o, mpz_limbs_read(M->x[t]), l * 8);
        len[t] = mpz_sgn(M->x[t]) < 0 ? -(int32_t) l : (int32_t) l;
        o += (int64_t) l;
    }
    return 0;
}

SLIP_info SLIP_hip_LU_solve(mpq_t **x, SLIP_dense *b, const mpz_t *rhos, const SLIP_sparse *L,
                            const SLIP_sparse *U, const int32_t *pinv)
{
    if (!x || !b || !rhos || !pinv || !L || !U || !b->x
        || !L->p || !L->i || !L->x || !U->p || !U->i || !U->x)
        return SLIP_INCORRECT_INPUT;                      /* SLIP_LU_solve.c:51-55 */
    const int32_t n = L->n, nrhs = b->n;
    if (n <= 0 || nrhs <= 0) return SLIP_INCORRECT_INPUT;
    SLIP_info ret = SLIP_OUT_OF_MEMORY;
    int32_t *rowperm = (int32_t *) malloc((size_t) n * 4), *blen = NULL, *Li = NULL, *Ui = NULL, *Llen = NULL, *Ulen = NULL;
    int64_t *Lp = NULL, *Up = NULL;
    uint64_t *blimbs = NULL, *Llimbs = NULL, *Ulimbs = NULL;
    int32_t *xlen = NULL; uint64_t *xlimbs = NULL; int64_t xnl = 0;
    slip_hip_factor *f = NULL;
    mpq_t det2; mpz_t num;
    int have_gmp = 0, rc;
    if (!rowperm) goto done;
    for (int32_t i = 0; i < n; i++) { if (pinv[i] < 0 || pinv[i] >= n) { ret = SLIP_INCORRECT_INPUT; goto done; } rowperm[pinv[i]] = i; }
    rc = sparse_to_slab(L, n, rowperm, &Lp, &Li, &Llen, &Llimbs);
    if (!rc) rc = sparse_to_slab(U, n, rowperm, &Up, &Ui, &Ulen, &Ulimbs);
    if (rc) { if (rc == 2) ret = SLIP_INCORRECT_INPUT; goto done; }
    /* b: dense, right-hand side k, row i at k*n+i (b->x[i][k], SLIP_LU_solve.c:68-75) */
    {
        int64_t nl = 0;
        for (int32_t i = 0; i < n; i++) for (int32_t k = 0; k < nrhs; k++) nl += (int64_t) mpz_size(b->x[i][k]);
        blen = (int32_t *) malloc((size_t) n * nrhs * 4);
        blimbs = (uint64_t *) malloc((size_t)(nl ? nl : 1) * 8);
        if (!blen || !blimbs) goto done;
        int64_t o = 0;
        for (int32_t k = 0; k < nrhs; k++)
            for (int32_t i = 0; i < n; i++) {
                size_t l = mpz_size(b->x[i][k]);
                if (l) memcpy(blimbs + o, mpz_limbs_read(b->x[i][k]), l * 8);
                blen[(int64_t) k * n + i] = mpz_sgn(b->x[i][k]) < 0 ? -(int32_t) l : (int32_t) l;
                o += (int64_t) l;
            }
    }
    rc = slip_hip_factor_from_factors(&f, n, Lp, Li, Llen, Llimbs, Up, Ui, Ulen, Ulimbs, pinv, NULL);
    if (rc == SLIP_HIP_OK) rc = slip_hip_factor_solve(f, nrhs, blen, blimbs, &xlen, &xlimbs, &xnl, NULL);
    if (rc != SLIP_HIP_OK) { ret = rc == SLIP_HIP_INCORRECT_INPUT ? SLIP_INCORRECT_INPUT : SLIP_OUT_OF_MEMORY; goto done; }
    /* x = b2 / det (slip_array_div.c:36-49) */
    mpq_init(det2); mpz_init(num); have_gmp = 1;
    mpq_set_num(det2, rhos[n - 1]);
    {
        int64_t o = 0;
        for (int32_t k = 0; k < nrhs; k++)
            for (int32_t p = 0; p < n; p++) {
                const int32_t sl = xlen[(int64_t) k * n + p];
                set_from_limbs(num, sl, xlimbs + o);
                o += sl < 0 ? -sl : sl;
                mpq_set_num(x[p][k], num);
                mpz_set_ui(mpq_denref(x[p][k]), 1);
                mpq_div(x[p][k], x[p][k], det2);
            }
    }
    ret = SLIP_OK;
done:
    if (have_gmp) { mpq_clear(det2); mpz_clear(num); }
    if (f) slip_hip_factor_destroy(f);
    slip_hip_free(xlen); slip_hip_free(xlimbs);
    free(rowperm); free(blen); free(blimbs); free(Lp); free(Li); free(Llen); free(Llimbs); free(Up); free(Ui); free(Ulen); free(Ulimbs);
    return ret;
}

SLIP_info SLIP_LU_solve(mpq_t **x, SLIP_dense *b, const mpz_t *rhos, const SLIP_sparse *L,
                        const SLIP_sparse *U, const int32_t *pinv)
{
    return SLIP_hip_LU_solve(x, b, rhos, L, U, pinv);
}

/* the reference's symbol: interposes when this library is linked ahead of libsliplu */
SLIP_info SLIP_LU_factorize(SLIP_sparse *L, SLIP_sparse *U, SLIP_sparse *A, SLIP_LU_analysis *S,
                            mpz_t *rhos, int32_t *pinv, SLIP_options *option)
{
    return SLIP_hip_LU_factorize(L, U, A, S, rhos, pinv, option);
}
