/* ref_lu_oracle.h -- C interface of the CPU restatement (oracle/ref_lu_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path and, where the
 * compiled reference (oracle/_ref) is unavailable, the "port" CPU baseline of
 * bench.py.  Nothing in the product path may call into this.
 */
#ifndef REF_LU_ORACLE_H
#define REF_LU_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: the reference's SLIP_info (SLIP_LU/Include/SLIP_LU.h:160-168) */
enum { ORC_OK = 0, ORC_OUT_OF_MEMORY = -1, ORC_SINGULAR = -2, ORC_INCORRECT_INPUT = -3 };

/* counter slots */
enum { ORC_N_UPD = 0, ORC_B_READ = 1, ORC_B_WRITE = 2, ORC_N_SRC = 3, ORC_L_STREAMED = 4,
       ORC_MAXLIMBS = 5, ORC_K_DONE = 6, ORC_LIMB_MACS = 7, ORC_NCOUNTERS = 8 };

typedef struct {
    int32_t  n, K, status;       /* K = columns committed                      */
    int64_t  lnz, unz;           /* entries in L(:,0:K), U(:,0:K)              */
    int64_t *Lp, *Up;            /* K+1 column pointers                        */
    int32_t *Li, *Ui;            /* ORIGINAL row ids, in the reference's order */
    int32_t *Llen, *Ulen;        /* sign * (number of 64-bit limbs); 0 = zero  */
    int64_t  Lnl, Unl;           /* total limbs                                */
    uint64_t *Llimbs, *Ulimbs;   /* little-endian limbs, entries back to back  */
    int32_t *rholen; uint64_t *rholimbs; int64_t rhonl;
    int32_t *pinv;               /* n, state after column K-1                  */
    int64_t  counters[ORC_NCOUNTERS];
    double   seconds;            /* wall time of the column loop               */
} orc_result;

/* Factorise columns [0,K) of A*Q.  A is CSC (Ap has n+1 entries), values as
 * (Alen[p] = sign*limbs, limbs back to back in Alimbs).  q = column order.
 * pivot = 0..5 (SLIP_pivot), tol as SLIP_options.tol.
 * Kmax <= 0 or > n: all columns.  cap > 0: stop BEFORE the first column that
 * holds a value of more than cap limbs.  Returns NULL on allocation failure. */
orc_result *orc_factorize(int32_t n, const int64_t *Ap, const int32_t *Ai,
                          const int32_t *Alen, const uint64_t *Alimbs,
                          const int32_t *q, int32_t pivot, double tol,
                          int32_t Kmax, int32_t cap);
void orc_free(orc_result *r);

/* REF forward/back substitution (SLIP_LU_solve.c:41-86): b is n-by-nrhs dense,
 * column major, entry (i,c) at b[c*n+i] as (blen, limbs back to back).
 * Needs a complete factorisation (K == n).  Returns numerators xnum (same
 * layout, row order = permuted order before SLIP_permute_x) over the common
 * denominator rhos[n-1]; the caller canonicalises.  out arrays malloc'ed. */
int orc_solve(const orc_result *f, int32_t nrhs, const int32_t *blen, const uint64_t *blimbs,
              int32_t **xlen_out, uint64_t **xlimbs_out, int64_t *xnl_out);

/* single big-integer operations, exposed for unit tests of the HIP limb kernels */
int orc_ipge(/* x_i */ int32_t xl, const uint64_t *x, /* rho_j */ int32_t rjl, const uint64_t *rj,
             /* L_m */ int32_t ll, const uint64_t *l, /* x_j */ int32_t xjl, const uint64_t *xj,
             /* rho_{j-1} (len 0: no division) */ int32_t rpl, const uint64_t *rp,
             /* history: multiply by hm then divide by hd first (len 0: skip) */
             int32_t hml, const uint64_t *hm, int32_t hdl, const uint64_t *hd,
             int32_t *outlen, uint64_t *out, int32_t outcap);

int orc_matgen(int32_t n, double density, int32_t bits, uint64_t seed, int64_t **Ap, int32_t **Ai, int64_t **Ax);
void orc_free_ptr(void *p);

#ifdef __cplusplus
}
#endif
#endif
