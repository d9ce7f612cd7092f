/* ref_driver.c -- driver around the UNMODIFIED compiled reference (oracle/_ref).
 *
 * TEST INFRASTRUCTURE ONLY.  Links oracle/_ref/libsliplu_ref.a, which is built
 * by oracle/Makefile from the reference sources where they lie under
 * /root/reference (nothing of the reference is copied into this repository).
 * Uses:
 *   - golden fixtures for tests/golden (L, U, rhos, pinv, q of the reference)
 *   - the CPU baseline of bench.py ("cpu_baseline.kind": "reference")
 *   - the algorithmic-byte counters of SURVEY.md section 8(d)
 *
 * Modes
 *   full   <input> <out.slab> [pivot order tol]   SLIP_LU_analyze + SLIP_LU_factorize
 *   window <input> <out.slab> K cap [pivot order tol]
 *          replays the reference's column loop (SLIP_LU_factorize.c:190-264)
 *          through its extern internals slip_REF_triangular_solve
 *          (SLIP_LU_internal.h:700) and slip_get_pivot (:513) for at most K
 *          columns, stopping BEFORE the first column that holds a value of
 *          more than `cap` 64-bit limbs (cap <= 0: no cap); counts the
 *          algorithmic bytes while doing so.
 *   order  <input> <out.slab> [order]             only the column ordering q
 *   solve  <input> <out.slab> [pivot order tol]   full factorisation, then SLIP_LU_solve on the deterministic
 *          right-hand side b_i = ((i*2654435761) mod 2001) - 1000 (one column); dumps the rational solution
 *          (before SLIP_permute_x) as numerator / denominator limb lists "xnum*", "xden*"
 * <input> is  trip:<path>  (the reference's triplet text, Demo/demos.c:245-331)
 *         or  gen:<n>,<density>,<bits>,<seed>     (slip_matgen.h)
 * Optional trailing argument  q:<file.slab>  takes q from a slab file instead
 * of running the ordering.
 *
 * All row indices are dumped as ORIGINAL row ids (the reference relabels
 * L->i, U->i through pinv at SLIP_LU_factorize.c:293-301; that is undone here).
 */
#include "SLIP_LU_internal.h"
#include <time.h>
#include "slabio.h"
#include "../slip_lu_amd/csrc/slip_matgen.h"

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); exit(2); } while (0)

static int64_t zlimbs(const mpz_t z) { return (int64_t) mpz_size(z); }

/* ---- flatten an array of mpz into (signed limb count, limbs) ---- */
typedef struct { int32_t *len; uint64_t *limbs; int64_t n, nl, cap; } flat_t;

static void flat_init(flat_t *f, int64_t n)
{
    f->n = 0; f->nl = 0; f->cap = 1024;
    f->len = (int32_t *) malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));
    f->limbs = (uint64_t *) malloc((size_t) f->cap * 8);
}
static void flat_push(flat_t *f, const mpz_t z)
{
    int64_t l = zlimbs(z);
    if (f->nl + l > f->cap) {
        while (f->nl + l > f->cap) f->cap *= 2;
        f->limbs = (uint64_t *) realloc(f->limbs, (size_t) f->cap * 8);
    }
    for (int64_t t = 0; t < l; t++) f->limbs[f->nl + t] = (uint64_t) mpz_getlimbn(z, t);
    f->len[f->n++] = (int32_t)(mpz_sgn(z) < 0 ? -l : l);
    f->nl += l;
}

/* ---- input ---- */
static SLIP_sparse *read_input(const char *spec)
{
    SLIP_sparse *A = SLIP_create_sparse();
    if (!A) DIE("out of memory");
    if (!strncmp(spec, "trip:", 5)) {
        FILE *f = fopen(spec + 5, "r");
        if (!f) DIE("cannot open %s", spec + 5);
        int32_t m, n, nz;
        if (fscanf(f, "%d %d %d", &m, &n, &nz) != 3) DIE("bad header");
        int32_t *I = malloc((size_t) nz * 4), *J = malloc((size_t) nz * 4);
        mpz_t *x = SLIP_create_mpz_array(nz);
        int dec = 1;
        for (int32_t p = 0; p < nz; p++) {
            if (gmp_fscanf(f, "%d %d %Zd", &I[p], &J[p], x[p]) != 3) DIE("bad entry %d", p);
            if (p == 0 && (I[0] == 0 || J[0] == 0)) dec = 0;   /* as Demo/demos.c:293-302 */
            I[p] -= dec; J[p] -= dec;
        }
        fclose(f);
        if (SLIP_build_sparse_trip_mpz(A, I, J, x, n, nz) != SLIP_OK) DIE("build failed");
        free(I); free(J); SLIP_delete_mpz_array(&x, nz);
    } else if (!strncmp(spec, "gen:", 4)) {
        int n, bits; double d; unsigned long long seed;
        if (sscanf(spec + 4, "%d,%lf,%d,%llu", &n, &d, &bits, &seed) != 4) DIE("bad gen spec");
        int64_t *Ap; int32_t *Ai; int64_t *Ax;
        if (slip_matgen_csc(n, d, bits, seed, &Ap, &Ai, &Ax)) DIE("matgen failed");
        int64_t nz = Ap[n];
        if (nz > INT32_MAX) DIE("nnz overflow");
        int32_t *p32 = malloc(((size_t) n + 1) * 4);
        for (int j = 0; j <= n; j++) p32[j] = (int32_t) Ap[j];
        mpz_t *x = SLIP_create_mpz_array((int32_t) nz);
        for (int64_t t = 0; t < nz; t++) mpz_set_si(x[t], (long) Ax[t]);
        if (SLIP_build_sparse_ccf_mpz(A, p32, Ai, x, n, (int32_t) nz) != SLIP_OK) DIE("build failed");
        free(p32); free(Ap); free(Ai); free(Ax); SLIP_delete_mpz_array(&x, (int32_t) nz);
    } else DIE("input must be trip:<path> or gen:n,d,bits,seed");
    return A;
}

/* read int32 array "q" out of a slab file (minimal reader) */
static int load_q(const char *path, int32_t *q, int32_t n)
{
    FILE *f = fopen(path, "rb"); if (!f) return -1;
    char magic[8]; if (fread(magic, 1, 8, f) != 8) { fclose(f); return -1; }
    for (;;) {
        char nm[24]; int32_t hdr[2]; int64_t cnt;
        if (fread(nm, 1, 24, f) != 24) break;
        if (fread(hdr, 4, 2, f) != 2 || fread(&cnt, 8, 1, f) != 1) break;
        size_t esz = hdr[0] == SLAB_I32 ? 4 : 8, bytes = esz * (size_t) cnt;
        if (!strcmp(nm, "q") && hdr[0] == SLAB_I32 && cnt >= n) {
            int ok = fread(q, 4, (size_t) n, f) == (size_t) n; fclose(f); return ok ? 0 : -1;
        }
        fseek(f, (long)(bytes + (8 - bytes % 8) % 8), SEEK_CUR);
    }
    fclose(f); return -1;
}

static void dump_A(FILE *out, SLIP_sparse *A)
{
    flat_t fa; flat_init(&fa, A->nz);
    for (int32_t p = 0; p < A->nz; p++) flat_push(&fa, A->x[p]);
    slab_put(out, "Ap", SLAB_I32, A->p, A->n + 1);
    slab_put(out, "Ai", SLAB_I32, A->i, A->nz);
    slab_put(out, "Alen", SLAB_I32, fa.len, fa.n);
    slab_put(out, "Alimbs", SLAB_U64, fa.limbs, fa.nl);
}

static void dump_factor(FILE *out, const char *pfx, SLIP_sparse *M, int32_t K, int32_t nz,
                        const int32_t *row_of /* NULL: ids already original */)
{
    char nm[24];
    flat_t fm; flat_init(&fm, nz);
    int32_t *ids = malloc((size_t)(nz > 0 ? nz : 1) * 4);
    for (int32_t p = 0; p < nz; p++) {
        flat_push(&fm, M->x[p]);
        ids[p] = row_of ? row_of[M->i[p]] : M->i[p];
    }
    int32_t *pp = malloc(((size_t) K + 1) * 4);
    for (int32_t k = 0; k < K; k++) pp[k] = M->p[k];
    pp[K] = nz;
    snprintf(nm, sizeof nm, "%sp", pfx);     slab_put(out, nm, SLAB_I32, pp, K + 1);
    snprintf(nm, sizeof nm, "%si", pfx);     slab_put(out, nm, SLAB_I32, ids, nz);
    snprintf(nm, sizeof nm, "%slen", pfx);   slab_put(out, nm, SLAB_I32, fm.len, fm.n);
    snprintf(nm, sizeof nm, "%slimbs", pfx); slab_put(out, nm, SLAB_U64, fm.limbs, fm.nl);
    free(ids); free(pp); free(fm.len); free(fm.limbs);
}

int main(int argc, char **argv)
{
    if (argc < 4) DIE("usage: ref_driver full|window|order <input> <out.slab> ...");
    const char *mode = argv[1], *qfile = NULL;
    if (!strncmp(argv[argc - 1], "q:", 2)) { qfile = argv[argc - 1] + 2; argc--; }
    SLIP_initialize();
    SLIP_sparse *A = read_input(argv[2]);
    int32_t n = A->n;
    SLIP_options *opt = SLIP_create_default_options();
    int is_window = !strcmp(mode, "window"), is_order = !strcmp(mode, "order"), is_solve = !strcmp(mode, "solve");
    int32_t K = n, cap = 0;
    int ai = 4;
    if (is_window) { if (argc < 6) DIE("window needs K cap"); K = atoi(argv[4]); cap = atoi(argv[5]); ai = 6; }
    if (is_order) { if (argc > ai) opt->order = (SLIP_col_order) atoi(argv[ai]); }
    else {
        if (argc > ai)     opt->pivot = (SLIP_pivot) atoi(argv[ai]);
        if (argc > ai + 1) opt->order = (SLIP_col_order) atoi(argv[ai + 1]);
        if (argc > ai + 2) opt->tol = atof(argv[ai + 2]);
    }
    if (K > n || K <= 0) K = n;

    SLIP_LU_analysis *S = SLIP_create_LU_analysis(n + 1);
    double t0 = now_s();
    if (SLIP_LU_analyze(S, A, opt) != SLIP_OK) DIE("analyze failed");
    double t_sym = now_s() - t0;
    if (qfile && load_q(qfile, S->q, n)) DIE("cannot load q from %s", qfile);

    FILE *out = slab_open(argv[3]);
    if (!out) DIE("cannot write %s", argv[3]);
    int32_t meta_n = n;
    slab_put(out, "n", SLAB_I32, &meta_n, 1);
    slab_put(out, "q", SLAB_I32, S->q, n);
    if (is_order) { dump_A(out, A); fclose(out); return 0; }

    mpz_t *rhos = SLIP_create_mpz_array(n);
    int32_t *pinv = malloc((size_t) n * 4);
    SLIP_sparse *L = SLIP_create_sparse(), *U = SLIP_create_sparse();
    int64_t counters[8] = {0};   /* N_upd, B_read, B_write, N_src, L_streamed, maxlimbs, K_done, status */
    double t_factor;
    int32_t Kdone = n, lnz = 0, unz = 0;
    SLIP_info ok = SLIP_OK;

    if (!is_window) {
        t0 = now_s();
        ok = SLIP_LU_factorize(L, U, A, S, rhos, pinv, opt);
        t_factor = now_s() - t0;
        if (ok != SLIP_OK) fprintf(stderr, "SLIP_LU_factorize returned %d\n", (int) ok);
        lnz = L->nz; unz = U->nz;
        counters[7] = ok;
        if (ok != SLIP_OK) Kdone = 0;
    } else {
        /* ---- replay of SLIP_LU_factorize.c:190-264 through the extern internals ---- */
        int32_t *pivs = malloc((size_t) n * 4), *h = malloc((size_t) n * 4),
                *xi = malloc((size_t) 2 * n * 4), *row_perm = malloc((size_t) n * 4);
        slip_reset_int_array(pivs, n); slip_reset_int_array(h, n);
        int32_t hint = cap > 0 ? 64 * (cap + 2) : 4096;
        mpz_t *x = slip_create_mpz_array2(n, hint);
        for (int32_t i = 0; i < n; i++) pinv[i] = row_perm[i] = i;
        if (slip_sparse_alloc2(L, n, n, S->lnz) != SLIP_OK) DIE("alloc L");
        if (slip_sparse_alloc2(U, n, n, S->unz) != SLIP_OK) DIE("alloc U");
        double t_acc = 0;
        int32_t k;
        for (k = 0; k < K; k++) {
            /* timed: only what SLIP_LU_factorize.c:190-264 itself does per column (realloc check, the triangular
             * solve, the pivot search, the L/U split); the driver's own cap scan and counter loops run untimed */
            double tk = now_s();
            L->p[k] = lnz; U->p[k] = unz;
            int32_t col = S->q[k], top, pivot;
            if (lnz + n > L->nzmax) { L->nz = lnz; if (slip_sparse_realloc(L) != SLIP_OK) DIE("realloc L"); }
            if (unz + n > U->nzmax) { U->nz = unz; if (slip_sparse_realloc(U) != SLIP_OK) DIE("realloc U"); }
            ok = slip_REF_triangular_solve(&top, L, A, k, xi, S->q, rhos, pinv, row_perm, h, x);
            t_acc += now_s() - tk;
            if (ok != SLIP_OK) break;
            /* cap test on the finished column, BEFORE it is committed */
            if (cap > 0) {
                int over = 0;
                for (int32_t j = top; j < n; j++) if (zlimbs(x[xi[j]]) > cap) { over = 1; break; }
                if (over) break;
            }
            /* counters of SURVEY.md 8(d): uses pinv BEFORE the pivot swap of column k */
            for (int32_t p = A->p[col]; p < A->p[col + 1]; p++) counters[1] += 4 + 8 * zlimbs(A->x[p]);
            for (int32_t j = top; j < n; j++) {
                int32_t r = xi[j], jnew = pinv[r];
                if (jnew >= k || mpz_sgn(x[r]) == 0) continue;
                counters[3]++;
                counters[1] += 8 * zlimbs(rhos[jnew]) + (jnew >= 1 ? 8 * zlimbs(rhos[jnew - 1]) : 0);
                for (int32_t m = L->p[jnew]; m < (jnew + 1 == k ? lnz : L->p[jnew + 1]); m++) {
                    counters[4]++;
                    counters[1] += 4 + 8 * zlimbs(L->x[m]);
                    if (pinv[L->i[m]] > jnew && mpz_sgn(L->x[m]) != 0) counters[0]++;
                }
            }
            tk = now_s();
            ok = slip_get_pivot(&pivot, x, pivs, n, top, xi, opt->pivot, col, k, rhos, pinv, row_perm, opt->tol);
            if (ok != SLIP_OK) break;
            const int64_t lnz0 = lnz, unz0 = unz;
            for (int32_t j = top; j < n; j++) {
                int32_t jnew = xi[j], loc = pinv[jnew];
                size_t size = mpz_sizeinbase(x[jnew], 2);
                if (loc <= k) { U->i[unz] = jnew; mpz_init2(U->x[unz], size + 2); mpz_set(U->x[unz], x[jnew]); unz++; }
                if (loc >= k) { L->i[lnz] = jnew; mpz_init2(L->x[lnz], size + 2); mpz_set(L->x[lnz], x[jnew]); lnz++; }
            }
            t_acc += now_s() - tk;
            for (int64_t t = unz0; t < unz; t++) { int64_t l = zlimbs(U->x[t]); if (l > counters[5]) counters[5] = l; counters[2] += 4 + 8 * l; }
            for (int64_t t = lnz0; t < lnz; t++) { int64_t l = zlimbs(L->x[t]); if (l > counters[5]) counters[5] = l; counters[2] += 4 + 8 * l; }
            counters[2] += 8 * zlimbs(rhos[k]);
        }
        Kdone = k; t_factor = t_acc;
        L->nz = lnz; U->nz = unz;
        counters[7] = ok;
        free(pivs); free(h); free(xi); free(row_perm);
    }
    counters[6] = Kdone;

    /* inverse of pinv, to undo the reference's final relabel in full mode */
    int32_t *row_of = NULL;
    if (!is_window && ok == SLIP_OK) {
        row_of = malloc((size_t) n * 4);
        for (int32_t i = 0; i < n; i++) row_of[pinv[i]] = i;
    }
    slab_put(out, "K", SLAB_I32, &Kdone, 1);
    slab_put(out, "pinv", SLAB_I32, pinv, n);
    if (Kdone > 0) {
        dump_factor(out, "L", L, Kdone, lnz, row_of);
        dump_factor(out, "U", U, Kdone, unz, row_of);
        flat_t fr; flat_init(&fr, Kdone);
        for (int32_t k = 0; k < Kdone; k++) flat_push(&fr, rhos[k]);
        slab_put(out, "rholen", SLAB_I32, fr.len, fr.n);
        slab_put(out, "rholimbs", SLAB_U64, fr.limbs, fr.nl);
    }
    if (is_solve && ok == SLIP_OK) {
        /* SLIP_LU_solve (SLIP_LU_solve.c:41-86) wants the relabelled factors SLIP_LU_factorize returned */
        SLIP_dense *b = SLIP_create_dense();
        mpz_t **bm = SLIP_create_mpz_mat(n, 1);
        for (int32_t i = 0; i < n; i++) mpz_set_si(bm[i][0], (long)(((unsigned) i * 2654435761u) % 2001u) - 1000);
        if (SLIP_build_dense_mpz(b, bm, n, 1) != SLIP_OK) DIE("build b");
        mpq_t **x = SLIP_create_mpq_mat(n, 1);
        double ts = now_s();
        SLIP_info sok = SLIP_LU_solve(x, b, rhos, L, U, pinv);
        ts = now_s() - ts;
        if (sok != SLIP_OK) DIE("SLIP_LU_solve returned %d", (int) sok);
        flat_t fn, fd; flat_init(&fn, n); flat_init(&fd, n);
        for (int32_t i = 0; i < n; i++) { flat_push(&fn, mpq_numref(x[i][0])); flat_push(&fd, mpq_denref(x[i][0])); }
        slab_put(out, "xnumlen", SLAB_I32, fn.len, fn.n); slab_put(out, "xnumlimbs", SLAB_U64, fn.limbs, fn.nl);
        slab_put(out, "xdenlen", SLAB_I32, fd.len, fd.n); slab_put(out, "xdenlimbs", SLAB_U64, fd.limbs, fd.nl);
        slab_put(out, "solve_seconds", SLAB_F64, &ts, 1);
    }
    slab_put(out, "counters", SLAB_I64, counters, 8);
    double tm[2] = { t_factor, t_sym };
    slab_put(out, "timing", SLAB_F64, tm, 2);
    fclose(out);
    fprintf(stderr, "ref_driver: mode=%s n=%d K=%d lnz=%d unz=%d L+U-K=%d t_factor=%.6f s t_sym=%.6f s status=%d\n",
            mode, n, Kdone, lnz, unz, lnz + unz - Kdone, t_factor, t_sym, (int) ok);
    return 0;
}
