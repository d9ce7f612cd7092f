/* dropin_driver.c -- the drop-in acceptance test, in the shape of the reference's own demo
 * (SLIP_LU/Demo/SLIPLU.c:152-367): a program written against the REFERENCE's public header and
 * library whose SLIP_LU_factorize call is served by whichever library comes first on the link
 * line -- libslip_lu_hip.so (HIP path; it serves SLIP_LU_solve as well) or the reference itself.  TEST ONLY; compiled here, where
 * the reference's headers exist (tests/dropin/Makefile), the binaries travel to the GPU box.
 *
 *   dropin_driver <triplet file> [pivot] [nrhs]
 *   dropin_driver --errors            (the error paths, in the shape of Tcov/cov_test.c:342-345,466-472,674-678,751-763)
 * prints: status of SLIP_check_solution (exact A x == b), then the demo's report numbers
 *   sum bits(rhos) / sum bits(L)+bits(U)-bits(rhos) / L->nz+U->nz-n     (SLIPLU.c:339-367)
 * and a FNV-1a hash over pinv, L, U (permuted ids, values) and rhos for a cheap equality check.
 */
#include "SLIP_LU.h"
#include <string.h>

#define OK(call) do { SLIP_info s_ = (call); if (s_ != SLIP_OK) { printf("ERROR %d at %s\n", (int) s_, #call); return 1; } } while (0)

static uint64_t fnv(uint64_t h, const void *p, size_t n)
{
    const unsigned char *c = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ULL; }
    return h;
}
static uint64_t fnv_mpz(uint64_t h, const mpz_t z)
{
    int sg = mpz_sgn(z); h = fnv(h, &sg, sizeof sg);
    size_t l = mpz_size(z);
    for (size_t t = 0; t < l; t++) { mp_limb_t v = mpz_getlimbn(z, t); h = fnv(h, &v, sizeof v); }
    return h;
}

/* Error paths of the two replaced entry points, as the reference's coverage test drives them
 * (Tcov/cov_test.c:674-678 NULL arguments, :466-472 a singular matrix under three pivot schemes,
 * :751-763 a corrupted b must fail SLIP_check_solution): only return codes are printed, so the
 * output of the HIP build must equal the reference build's. */
static int error_paths(void)
{
    SLIP_initialize();
    printf("null_all factorize=%d solve=%d\n",
           (int) SLIP_LU_factorize(NULL, NULL, NULL, NULL, NULL, NULL, NULL),
           (int) SLIP_LU_solve(NULL, NULL, NULL, NULL, NULL, NULL));
    /* 4x4, columns 1 and 3 equal: singular whatever the pivoting does */
    const int32_t n = 4, nz = 10;
    int32_t I[10] = {0, 1, 0, 2, 1, 3, 0, 2, 3, 2}, J[10] = {0, 0, 1, 1, 2, 2, 3, 3, 0, 2};
    long V[10] = {3, -7, 5, 11, 2, 9, 5, 11, 4, 6};
    mpz_t *xv = SLIP_create_mpz_array(nz);
    for (int32_t p = 0; p < nz; p++) mpz_set_si(xv[p], V[p]);
    SLIP_options *option = SLIP_create_default_options();
    const int schemes[3] = {(int) SLIP_TOL_SMALLEST, (int) SLIP_LARGEST, (int) SLIP_FIRST_NONZERO};
    for (int s = 0; s < 3; s++) {
        SLIP_sparse *A = SLIP_create_sparse(), *L = SLIP_create_sparse(), *U = SLIP_create_sparse();
        SLIP_LU_analysis *S = SLIP_create_LU_analysis(n + 1);
        mpz_t *rhos = SLIP_create_mpz_array(n);
        int32_t *pinv = (int32_t *) SLIP_malloc((size_t) n * sizeof(int32_t));
        option->pivot = (SLIP_pivot) schemes[s];
        OK(SLIP_build_sparse_trip_mpz(A, I, J, xv, n, nz));
        OK(SLIP_LU_analyze(S, A, option));
        /* one missing argument at a time (SLIP_LU_factorize.c:48-52) */
        printf("pivot %d: null_L=%d null_A=%d null_S=%d null_rhos=%d null_pinv=%d null_opt=%d", schemes[s],
               (int) SLIP_LU_factorize(NULL, U, A, S, rhos, pinv, option), (int) SLIP_LU_factorize(L, U, NULL, S, rhos, pinv, option),
               (int) SLIP_LU_factorize(L, U, A, NULL, rhos, pinv, option), (int) SLIP_LU_factorize(L, U, A, S, NULL, pinv, option),
               (int) SLIP_LU_factorize(L, U, A, S, rhos, NULL, option), (int) SLIP_LU_factorize(L, U, A, S, rhos, pinv, NULL));
        printf(" singular=%d\n", (int) SLIP_LU_factorize(L, U, A, S, rhos, pinv, option));
        /* whatever came back is the caller's: the reference's destructors must cope with it */
        SLIP_delete_sparse(&L); SLIP_delete_sparse(&U); SLIP_delete_sparse(&A);
        SLIP_delete_mpz_array(&rhos, n); SLIP_delete_LU_analysis(&S); SLIP_free(pinv);
    }
    SLIP_delete_mpz_array(&xv, nz);
    SLIP_free(option);
    SLIP_finalize();
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) { printf("usage: dropin_driver <triplet> [pivot] [nrhs] | --errors\n"); return 2; }
    if (!strcmp(argv[1], "--errors")) return error_paths();
    const int32_t nrhs = argc > 3 ? atoi(argv[3]) : 1;
    SLIP_initialize();
    FILE *f = fopen(argv[1], "r");
    if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
    int32_t m, n, nz;
    if (fscanf(f, "%d %d %d", &m, &n, &nz) != 3) return 2;
    int32_t *I = (int32_t *) malloc((size_t) nz * 4), *J = (int32_t *) malloc((size_t) nz * 4);
    mpz_t *xv = SLIP_create_mpz_array(nz);
    for (int32_t p = 0; p < nz; p++) {
        if (gmp_fscanf(f, "%d %d %Zd", &I[p], &J[p], xv[p]) != 3) return 2;
        I[p]--; J[p]--;
    }
    fclose(f);
    SLIP_sparse *A = SLIP_create_sparse(), *L = SLIP_create_sparse(), *U = SLIP_create_sparse();
    SLIP_dense *b = SLIP_create_dense();
    SLIP_options *option = SLIP_create_default_options();
    if (argc > 2) option->pivot = (SLIP_pivot) atoi(argv[2]);
    OK(SLIP_build_sparse_trip_mpz(A, I, J, xv, n, nz));

    /* b: a deterministic integer right-hand side */
    mpz_t **bm = SLIP_create_mpz_mat(n, nrhs);
    for (int32_t i = 0; i < n; i++)
        for (int32_t c = 0; c < nrhs; c++) mpz_set_si(bm[i][c], ((long)((i * 2654435761u) % 2001) - 1000) * (c % 2 ? -3 : 1) + c);
    OK(SLIP_build_dense_mpz(b, bm, n, nrhs));

    mpz_t *rhos = SLIP_create_mpz_array(n);
    int32_t *pinv = (int32_t *) SLIP_malloc((size_t) n * sizeof(int32_t));
    mpq_t **x = SLIP_create_mpq_mat(n, nrhs);
    SLIP_LU_analysis *S = SLIP_create_LU_analysis(n + 1);

    OK(SLIP_LU_analyze(S, A, option));
    OK(SLIP_LU_factorize(L, U, A, S, rhos, pinv, option));        /* <-- the replaced call */
    OK(SLIP_LU_solve(x, b, rhos, L, U, pinv));                    /* <-- replaced too (HIP forward/back substitution) */
    uint64_t hx = 1469598103934665603ULL;                         /* the exact rational solution, before permute */
    for (int32_t i = 0; i < n; i++)
        for (int32_t c = 0; c < nrhs; c++) { hx = fnv_mpz(hx, mpq_numref(x[i][c])); hx = fnv_mpz(hx, mpq_denref(x[i][c])); }
    OK(SLIP_permute_x(x, n, nrhs, S));
    SLIP_info check = SLIP_check_solution(A, x, b);
    /* a corrupted right-hand side must be noticed (Tcov/cov_test.c:751-763) */
    mpz_add_ui(b->x[0][0], b->x[0][0], 1000);
    const SLIP_info check_bad = SLIP_check_solution(A, x, b);
    mpz_sub_ui(b->x[0][0], b->x[0][0], 1000);

    size_t brho = 0, blu = 0;
    uint64_t h = 1469598103934665603ULL;
    h = fnv(h, pinv, (size_t) n * 4);
    for (int32_t k = 0; k < n; k++) { brho += mpz_sizeinbase(rhos[k], 2); h = fnv_mpz(h, rhos[k]); }
    h = fnv(h, L->p, ((size_t) n + 1) * 4); h = fnv(h, L->i, (size_t) L->nz * 4);
    h = fnv(h, U->p, ((size_t) n + 1) * 4); h = fnv(h, U->i, (size_t) U->nz * 4);
    for (int32_t t = 0; t < L->nz; t++) { blu += mpz_sizeinbase(L->x[t], 2); h = fnv_mpz(h, L->x[t]); }
    for (int32_t t = 0; t < U->nz; t++) { blu += mpz_sizeinbase(U->x[t], 2); h = fnv_mpz(h, U->x[t]); }
    printf("check=%d check_corrupt=%d nzmaxL=%d nzL=%d nzmaxU=%d nzU=%d report %zu %zu %d hash %016llx xhash %016llx\n", (int) check, (int) check_bad,
           L->nzmax, L->nz, U->nzmax, U->nz, brho, blu - brho, L->nz + U->nz - n, (unsigned long long) h, (unsigned long long) hx);

    /* the caller owns everything: free through the reference's own destructors */
    SLIP_delete_sparse(&A); SLIP_delete_sparse(&L); SLIP_delete_sparse(&U);
    SLIP_delete_dense(&b); SLIP_delete_mpz_mat(&bm, n, nrhs); SLIP_delete_mpq_mat(&x, n, nrhs);
    SLIP_delete_mpz_array(&rhos, n); SLIP_delete_mpz_array(&xv, nz); SLIP_delete_LU_analysis(&S);
    SLIP_free(pinv); SLIP_free(option); free(I); free(J);
    SLIP_finalize();
    return check == SLIP_OK ? 0 : 1;
}
