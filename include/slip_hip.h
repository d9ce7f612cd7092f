/* slip_hip.h -- C ABI of the MI355X-native REF sparse LU hot path (libslip_hip.so).
 *
 * Plain pointers and sizes only: no GMP, no torch, no C++ types.  Big integers
 * cross this boundary as "limb slabs": per entry a signed limb count
 * (sign * number of 64-bit limbs, 0 = zero) and the limbs little-endian, back
 * to back -- the same words GMP keeps behind an mpz_t (_mp_size, _mp_d).
 *
 * What each entry point replaces in cjh10644/SLIP_LU (reference file:line):
 *
 *   slip_hip_factor_create/run/download  <->  SLIP_LU_factorize
 *        SLIP_LU/Include/SLIP_LU.h:854-863, SLIP_LU/Source/SLIP_LU_factorize.c:36-314
 *        (with slip_REF_triangular_solve.c:65-265, slip_reach.c, slip_dfs.c,
 *         slip_sort_xi.c and slip_get_pivot.c:30-183 inside the column loop)
 *   slip_hip_options                     <->  SLIP_options.pivot / .tol
 *        SLIP_LU/Include/SLIP_LU.h:212-223; defaults SLIP_LU_internal.h:136-149
 *   slip_hip_factor_solve                <->  the integer core of SLIP_LU_solve
 *        SLIP_LU/Source/SLIP_LU_solve.c:41-86 (slip_forward_sub.c, slip_array_mul.c, slip_back_sub.c)
 *   status codes                         <->  SLIP_info, SLIP_LU.h:160-168
 *
 * The GMP-typed drop-in  SLIP_LU_factorize(L,U,A,S,rhos,pinv,option)  built on
 * top of this ABI is declared in include/SLIP_LU_hip.h (libslip_lu_hip.so).
 *
 * Threading: like the reference, one factorisation per handle at a time; the
 * calls block until the device work is complete.
 */
#ifndef SLIP_HIP_H
#define SLIP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SLIP_info values (SLIP_LU.h:160-168), plus one code kept outside them */
#define SLIP_HIP_OK               0
#define SLIP_HIP_OUT_OF_MEMORY   (-1)
#define SLIP_HIP_SINGULAR        (-2)
#define SLIP_HIP_INCORRECT_INPUT (-3)
#define SLIP_HIP_DEVICE_ERROR    (-100)   /* HIP runtime failure (no GPU, launch error) */

typedef struct slip_hip_options {
    int32_t pivot;        /* SLIP_pivot 0..5; default 3 = SLIP_TOL_SMALLEST            */
    double  tol;          /* SLIP_options.tol; default 1.0                             */
    int32_t limb_cap;     /* > 0: column-window mode -- stop BEFORE the first column   */
                          /*      that holds a value of more than limb_cap limbs       */
    int32_t waves;        /* waves per workgroup (0 = default 8)                        */
    int64_t lnz_hint;     /* initial capacity of L / U in entries (0 = 4*nnz(A)+n),    */
    int64_t unz_hint;     /*   cf. SLIP_LU_analysis.lnz/.unz; both grow on demand      */
    int32_t workers;      /* column workers (workgroups of a launch; each owns a private */
                          /*   dense vector): 0 = as many as can be resident            */
    int32_t reserved;     /* diagnostics: bit 0 = no early commit, bit 1 = no committer workgroup, bit 2 = no helping with long update queues, bit 3 = no full packages (chain engine off) */
} slip_hip_options;

typedef struct slip_hip_info {
    int32_t n;
    int32_t K;            /* columns committed so far                                  */
    int32_t status;       /* SLIP_HIP_* of the last run                                */
    int32_t window_end;   /* 1: run stopped at the limb cap (column K not committed)   */
    int64_t lnz, unz;     /* entries of L(:,0:K), U(:,0:K) (pivots in both)            */
    int64_t l_limbs, u_limbs;
    /* algorithmic counters of SURVEY.md 8(d), counted by the device */
    int64_t n_upd, b_read, b_write, n_src, l_streamed, max_limbs;
    double  kernel_ms;    /* device time of the factorisation kernels of the last run  */
    int32_t launches;     /* kernel launches of the last run (regrows relaunch)        */
    int32_t xcap_digits;  /* current stride of the dense scatter vectors, 32-bit digits */
    int64_t limb_macs;    /* sum over the IPGE updates of l(L_m)*l(x_j) + l(x_i)*l(rho_jn), 64-bit limbs */
    int32_t workers, waves;   /* launch shape: column workers x wavefronts each          */
    int32_t lds_bytes;        /* dynamic LDS per worker                                   */
    int32_t short_commits;    /* columns whose pivot was published by the short commit chain (diagnostic) */
    int32_t committer_commits;         /* ... of those, by the committer workgroup */
    int32_t farm_jobs, farm_items, pad2;   /* update queues opened to helpers; items helpers ran (diagnostic) */
    int32_t engine_commits, engine_sources; /* columns committed by the committer's chain engine from FULL packages; late sources it applied */
    int32_t retractions, reexports;        /* packages a worker took back because a source arrived; packages exported again */
} slip_hip_info;

typedef struct slip_hip_factor slip_hip_factor;

void slip_hip_default_options(slip_hip_options *opt);

/* number of HIP devices visible (0 if none / runtime missing) */
int slip_hip_device_count(void);

/* Upload A (CSC: Ap[n+1], Ai, per-entry signed limb counts Alen, limbs back to
 * back in Alimbs, entry order as in the reference's SLIP_sparse: unsorted rows
 * allowed, a duplicated row keeps the LAST value as slip_get_column.c:22 does)
 * and the column order q[n] (SLIP_LU_analysis.q).  All inputs are host
 * pointers; they are copied. */
int slip_hip_factor_create(slip_hip_factor **out, int32_t n,
                           const int64_t *Ap, const int32_t *Ai,
                           const int32_t *Alen, const uint64_t *Alimbs,
                           const int32_t *q, const slip_hip_options *opt);

/* Forget all columns: back to k = 0 with A and q still resident. */
int slip_hip_factor_reset(slip_hip_factor *f);

/* Factorise columns [K, kmax) (kmax <= 0 or > n: n).  stream: a hipStream_t
 * passed as void* (NULL = default stream).  Returns SLIP_HIP_OK when kmax (or
 * the limb cap) was reached, SLIP_HIP_SINGULAR, SLIP_HIP_OUT_OF_MEMORY, ... */
int slip_hip_factor_run(slip_hip_factor *f, int32_t kmax, void *stream);

int slip_hip_factor_info(const slip_hip_factor *f, slip_hip_info *info);

/* Copy the factors to host arrays sized from slip_hip_factor_info:
 *   Lp[K+1], Li[lnz], Llen[lnz], Llimbs[l_limbs]   (same for U)
 *   rholen[K], rholimbs[sum |rholen|], pinv[n]
 * Row indices are ORIGINAL row ids in the reference's entry order
 * (SLIP_LU_factorize.c:226-263); apply pinv for the final relabel (:293-301).
 * Any pointer may be NULL to skip that array.  Every limb array travels with its
 * capacity: *_limbs_inout: in = capacity of the array in limbs, out = limbs written
 * (required whenever the array is given).  The entry records on the device are checked
 * against the device's own limb counters and against the slab before anything is
 * copied: an inconsistency is SLIP_HIP_DEVICE_ERROR, a short array SLIP_HIP_INCORRECT_INPUT;
 * in neither case is a limb array written. */
int slip_hip_factor_download(const slip_hip_factor *f,
                             int64_t *Lp, int32_t *Li, int32_t *Llen, uint64_t *Llimbs, int64_t *L_limbs_inout,
                             int64_t *Up, int32_t *Ui, int32_t *Ulen, uint64_t *Ulimbs, int64_t *U_limbs_inout,
                             int32_t *rholen, uint64_t *rholimbs, int64_t *rho_limbs_inout,
                             int32_t *pinv);

/* REF forward/back substitution on the factors still resident in HBM -- the arithmetic of
 * SLIP_LU_solve (SLIP_LU/Source/SLIP_LU_solve.c:41-86: b2 = P b, slip_forward_sub.c:61-158,
 * slip_array_mul.c:19 by det = rhos[n-1], slip_back_sub.c:36-52).  Needs the complete
 * factorisation (K == n).  b is dense, nrhs columns of n entries in ORIGINAL row order:
 * blen[c*n+i] = signed limb count, limbs back to back in blimbs in that order.  The result is
 * what SLIP_LU_solve leaves in x before SLIP_permute_x / the division by det*scale
 * (SLIP_solve_double etc. do those on the host): integer numerators over det, entry c*n+p for
 * pivot POSITION p (x_final[q[p]] = xnum[p] / det), same signed-limb-slab form.  *xlen_out and
 * *xlimbs_out are malloc'ed; release with slip_hip_free. */
int slip_hip_factor_solve(slip_hip_factor *f, int32_t nrhs, const int32_t *blen, const uint64_t *blimbs,
                          int32_t **xlen_out, uint64_t **xlimbs_out, int64_t *xnl_out, void *stream);
/* A handle around factors the CALLER already holds (what SLIP_LU_solve is given, SLIP_LU.h:941-949), for
 * slip_hip_factor_solve only (run/reset refuse it).  L, U in the form slip_hip_factor_download produces:
 * column pointers, ORIGINAL row ids in the reference's entry order (the pivot LAST in every U column,
 * slip_back_sub.c:43), signed limb counts, limbs back to back; pinv[n].  The pivots rho_k are read from L
 * (the entry of L(:,k) in the row with pinv == k).  Host pointers, copied. */
int slip_hip_factor_from_factors(slip_hip_factor **out, int32_t n,
                                 const int64_t *Lp, const int32_t *Li, const int32_t *Llen, const uint64_t *Llimbs,
                                 const int64_t *Up, const int32_t *Ui, const int32_t *Ulen, const uint64_t *Ulimbs,
                                 const int32_t *pinv, const slip_hip_options *opt);
/* device time of the solve kernels of the last slip_hip_factor_solve, milliseconds */
double slip_hip_factor_solve_ms(const slip_hip_factor *f);

/* Subtree farm (no counterpart in the reference, which has no parallelism): multiply the K committed
 * columns by per-column scales on the device -- L(:,k) and rho[k] by scale[k], an entry of U in the row whose pivot sits
 * at position p by scale[p] -- where scale[k] is the product of the pivots the OTHER independent blocks had produced when
 * global column k was eliminated (slip_lu_amd/parallel.py: subtree_scales).  scale[k]: signed limb counts slen[nscales], limbs back
 * to back; nscales must equal the K committed columns (SLIP_HIP_INCORRECT_INPUT otherwise).  The rescaled copy is what slip_hip_factor_download / _info then serve, until the next reset (or rescale);
 * the handle's own factors are untouched, so run / solve keep working on the local values. */
int slip_hip_factor_rescale(slip_hip_factor *f, int32_t nscales, const int32_t *slen, const uint64_t *slimbs, void *stream);

/* Subtree farm, last step ("completed L columns gathered", then the separator columns): the first K columns
 * of THIS matrix's factorisation are given -- the blocks' columns, factorised on other handles / ranks and rescaled -- and
 * slip_hip_factor_run continues with column K exactly as SLIP_LU_factorize.c:190-264 does from k = K.  L, U: the K columns in
 * the form slip_hip_factor_download produces (column pointers Lp[K+1] / Up[K+1], ORIGINAL row ids in the reference's entry
 * order, the pivot LAST in every U column, signed limb counts, limbs back to back); piv_row[K]: the pivot row of every
 * column, from which the row permutation is replayed (slip_get_pivot.c:164-176).  The handle is reset first; K = 0 is a
 * reset.  Host pointers, copied.  Inconsistent input (a pivot row that is already pivotal or missing from its column, an
 * empty U column) is SLIP_HIP_INCORRECT_INPUT. */
int slip_hip_factor_set_prefix(slip_hip_factor *f, int32_t K,
                               const int64_t *Lp, const int32_t *Li, const int32_t *Llen, const uint64_t *Llimbs,
                               const int64_t *Up, const int32_t *Ui, const int32_t *Ulen, const uint64_t *Ulimbs,
                               const int32_t *piv_row);

void slip_hip_factor_destroy(slip_hip_factor *f);
/* Device buffers of destroyed handles are kept in a process-level pool for the next handle (the drop-in SLIP_LU_factorize
 * creates and destroys one per call): at most SLIP_HIP_POOL_MB megabytes (environment, default 32768; 0 = no pool).  This
 * gives them all back to the runtime. */
void slip_hip_pool_release(void);

/* Triplet files <-> limb slabs, host only.  read: what SLIP_tripread + SLIP_build_sparse_trip_mpz
 * produce (SLIP_LU/Demo/demos.c:245-331, SLIP_LU/Source/slip_trip_to_mat.c:23-69: "m n nz" then nz lines "i j value",
 * indices 1-based unless the first entry holds a 0, columns by counting sort in file order, duplicates kept), as the
 * arrays slip_hip_factor_create takes; they are malloc'ed, release with slip_hip_free.  Malformed input is
 * SLIP_HIP_INCORRECT_INPUT as in the reference.  write: the inverse (1-based, decimal), readable by either. */
int slip_hip_read_triplet(const char *path, int32_t *n_out, int64_t **Ap, int32_t **Ai, int32_t **Alen, uint64_t **Alimbs,
                          int64_t *nlimbs_out);
int slip_hip_write_triplet(const char *path, int32_t n, const int64_t *Ap, const int32_t *Ai, const int32_t *Alen,
                           const uint64_t *Alimbs);

/* Deterministic synthetic CSC generator of the benchmark configs
 * (slip_matgen.h): arrays are malloc'ed, release with slip_hip_free. */
int slip_hip_matgen(int32_t n, double density, int32_t bits, uint64_t seed,
                    int64_t **Ap, int32_t **Ai, int64_t **Ax);
void slip_hip_free(void *p);

/* Wave-level limb kernels (wave_bigint.h) run in isolation on the device, one
 * wavefront per operation, for the parity unit tests.  Operands are arrays of
 * 32-bit digits.  op: 0 = low product a*b mod B^W, 1 = a+b mod B^W,
 * 2 = a-b mod B^W, 3 = inverse of odd a modulo B^W (b unused); 10..13 = the same four
 * on the register-resident primitives (wave_bigint_reg.h, W <= 256), 14 = a >> lb bits.
 * out receives nops*W digits. */
int slip_hip_wave_op_test(int32_t op, int32_t nops, int32_t la, int32_t lb, int32_t W,
                          const uint32_t *a, const uint32_t *b, uint32_t *out);

/* Diagnostic builds (-DSLIP_PROFILE_PHASES) only: shader cycles thread 0 of every worker spent per phase of its
 * columns during the last run (summed over the workers); all zero in the product build. */
int slip_hip_factor_phase_cycles(const slip_hip_factor *f, unsigned long long *out24);   /* 24 slots */

const char *slip_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLIP_HIP_H */
