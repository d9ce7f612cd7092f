/* SLIP_LU_hip.h -- the GMP-typed drop-in of the hot path (libslip_lu_hip.so).
 *
 * Exports the reference's own entry point
 *
 *     SLIP_info SLIP_LU_factorize(SLIP_sparse *L, SLIP_sparse *U, SLIP_sparse *A,
 *                                 SLIP_LU_analysis *S, mpz_t *rhos, int32_t *pinv,
 *                                 SLIP_options *option);
 *
 * with the prototype, argument meaning, ownership and error codes of
 * cjh10644/SLIP_LU, SLIP_LU/Include/SLIP_LU.h:854-863 (implementation replaced:
 * SLIP_LU/Source/SLIP_LU_factorize.c:36-314).  A program built against the
 * reference's SLIP_LU.h needs NO source change: link libslip_lu_hip.so ahead of
 * (or instead of the SLIP_LU_factorize.o inside) the reference library; every
 * other SLIP_* symbol keeps coming from the reference.  See INTEGRATION.md.
 *
 * The same function is also exported as SLIP_hip_LU_factorize, for processes
 * that load both libraries and want to call either explicitly (the tests do).
 *
 * SLIP_LU_solve (SLIP_LU.h:941-949) is served the same way (SLIP_hip_LU_solve below).
 *
 * When the reference's SLIP_LU.h has been included first, this header only adds
 * the aliases; otherwise it declares layout-compatible mirrors of the four types
 * the call touches (SLIP_LU.h:160-168 SLIP_info, :212-223 SLIP_options,
 * :246-256 SLIP_sparse, :308-316 SLIP_LU_analysis).
 */
#ifndef SLIP_LU_HIP_H
#define SLIP_LU_HIP_H

#include <stdint.h>
#include <gmp.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SLIP_LU_VERSION          /* the reference's header is not in scope */
typedef enum {
    SLIP_OK = 0, SLIP_OUT_OF_MEMORY = -1, SLIP_SINGULAR = -2,
    SLIP_INCORRECT_INPUT = -3, SLIP_INCORRECT = -4
} SLIP_info;

typedef struct SLIP_options {
    int32_t pivot;               /* SLIP_pivot: 0 smallest .. 3 tol-smallest (default) .. 5 largest */
    int32_t order;               /* SLIP_col_order (used by SLIP_LU_analyze only) */
    double  tol;
    int32_t print_level;
    uint64_t prec;
    int32_t SLIP_MPFR_ROUND;     /* mpfr_rnd_t */
} SLIP_options;

typedef struct {
    int32_t m, n, nzmax, nz;
    int32_t *p, *i;
    mpz_t *x;
    mpq_t scale;
} SLIP_sparse;

typedef struct {
    int32_t *q;
    int32_t lnz, unz;
} SLIP_LU_analysis;

typedef struct {                 /* SLIP_LU.h:277-284 */
    int32_t m, n;
    mpz_t **x;                   /* x[i][k]: row i of right-hand side k */
    mpq_t scale;
} SLIP_dense;

SLIP_info SLIP_LU_solve(mpq_t **x, SLIP_dense *b, const mpz_t *rhos, const SLIP_sparse *L,
                        const SLIP_sparse *U, const int32_t *pinv);

SLIP_info SLIP_LU_factorize(SLIP_sparse *L, SLIP_sparse *U, SLIP_sparse *A, SLIP_LU_analysis *S,
                            mpz_t *rhos, int32_t *pinv, SLIP_options *option);
#endif

SLIP_info SLIP_hip_LU_factorize(SLIP_sparse *L, SLIP_sparse *U, SLIP_sparse *A, SLIP_LU_analysis *S,
                                mpz_t *rhos, int32_t *pinv, SLIP_options *option);

/* SLIP_LU_solve (SLIP_LU.h:941-949; SLIP_LU_solve.c:41-86) on the GPU: forward substitution, scaling by
 * det = rhos[n-1] and back substitution run in slip_hip_factor_solve on the uploaded L, U; the rational
 * x = b2/det is formed on the host exactly as slip_array_div.c does.  Same arguments, ownership and
 * error codes as the reference; also exported under the reference's own name. */
SLIP_info SLIP_hip_LU_solve(mpq_t **x, SLIP_dense *b, const mpz_t *rhos, const SLIP_sparse *L,
                            const SLIP_sparse *U, const int32_t *pinv);

#ifdef __cplusplus
}
#endif
#endif /* SLIP_LU_HIP_H */
