/* ref_lu_oracle.c -- CPU restatement of the reference's REF sparse LU hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker for the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg); the product
 * never links or calls it.
 *
 * It restates, in plain C99 with its own sign-magnitude big integers (64-bit
 * limbs, no GMP), the algorithm of cjh10644/SLIP_LU:
 *   column loop            SLIP_LU/Source/SLIP_LU_factorize.c:190-264
 *   reach (DFS on G(L))    SLIP_LU/Source/slip_reach.c:19-51, slip_dfs.c:19-81
 *   pattern sort           SLIP_LU/Source/slip_sort_xi.c:27-48
 *   scatter / reset        SLIP_LU/Source/slip_REF_triangular_solve.c:105-119
 *   history + IPGE sweep   SLIP_LU/Source/slip_REF_triangular_solve.c:124-259
 *   pivot search           SLIP_LU/Source/slip_get_pivot.c:30-183,
 *                          slip_get_smallest_pivot.c:25-101 (and largest / nonzero)
 *   forward/back solve     SLIP_LU/Source/slip_forward_sub.c:33-164,
 *                          slip_back_sub.c:21-54, SLIP_LU_solve.c:41-86
 * The arithmetic lives in GMP in the reference (libgmp, not vendored, version
 * unpinned; call sites SLIP_gmp.c:626-735); exact integer results do not
 * depend on the GMP version, so mul / submul / divexact are restated here with
 * schoolbook multiplication and Jebelean exact division.
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks this file bit-for-bit
 * against L, U, rhos, pinv dumped from the compiled reference (oracle/_ref,
 * built by oracle/Makefile from /root/reference) for the reference's own
 * ExampleMats, and against the algorithmic-byte anchors of SURVEY.md 8(d).
 */
#include "ref_lu_oracle.h"
#include "../slip_lu_amd/csrc/slip_matgen.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ */
/* big integers: sign-magnitude, little-endian 64-bit limbs            */
/* ------------------------------------------------------------------ */
typedef struct { uint64_t *d; int32_t alloc; int32_t size; } bz;   /* size<0: negative */

static int bz_oom = 0;

static void bz_init(bz *a) { a->d = NULL; a->alloc = 0; a->size = 0; }
static void bz_clear(bz *a) { free(a->d); a->d = NULL; a->alloc = 0; a->size = 0; }
static int bz_len(const bz *a) { return a->size < 0 ? -a->size : a->size; }
static int bz_sgn(const bz *a) { return (a->size > 0) - (a->size < 0); }

static void bz_reserve(bz *a, int n)
{
    if (n <= a->alloc) return;
    int na = a->alloc ? a->alloc : 2;
    while (na < n) na *= 2;
    uint64_t *p = (uint64_t *) realloc(a->d, (size_t) na * 8);
    if (!p) { bz_oom = 1; return; }
    a->d = p; a->alloc = na;
}
static void bz_set(bz *r, const bz *a)
{
    if (r == a) return;
    int l = bz_len(a);
    bz_reserve(r, l); if (bz_oom) return;
    if (l) memcpy(r->d, a->d, (size_t) l * 8);
    r->size = a->size;
}
static void bz_set_limbs(bz *r, int32_t slen, const uint64_t *limbs)
{
    int l = slen < 0 ? -slen : slen;
    bz_reserve(r, l); if (bz_oom) return;
    if (l) memcpy(r->d, limbs, (size_t) l * 8);
    while (l > 0 && r->d[l - 1] == 0) l--;
    r->size = slen < 0 ? -l : l;
}
static void bz_swap(bz *a, bz *b) { bz t = *a; *a = *b; *b = t; }

static int mag_cmp(const uint64_t *a, int la, const uint64_t *b, int lb)
{
    if (la != lb) return la > lb ? 1 : -1;
    for (int i = la - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i] ? 1 : -1;
    return 0;
}
static int bz_cmpabs(const bz *a, const bz *b) { return mag_cmp(a->d, bz_len(a), b->d, bz_len(b)); }

static int64_t bz_bits(const bz *a)
{
    int l = bz_len(a);
    if (!l) return 0;
    return 64 * (int64_t)(l - 1) + (64 - __builtin_clzll(a->d[l - 1]));
}

/* r = |a| * |b| (magnitudes), schoolbook; r must not alias */
static void mag_mul(uint64_t *r, const uint64_t *a, int la, const uint64_t *b, int lb)
{
    memset(r, 0, (size_t)(la + lb) * 8);
    for (int i = 0; i < la; i++) {
        uint64_t carry = 0, ai = a[i];
        for (int j = 0; j < lb; j++) {
            u128 t = (u128) ai * b[j] + r[i + j] + carry;
            r[i + j] = (uint64_t) t; carry = (uint64_t)(t >> 64);
        }
        r[i + lb] = carry;
    }
}

/* r = a * b  (mpz_mul, SLIP_gmp.c:626) */
static void bz_mul(bz *r, const bz *a, const bz *b)
{
    int la = bz_len(a), lb = bz_len(b);
    if (!la || !lb) { r->size = 0; return; }
    bz t; bz_init(&t); bz_reserve(&t, la + lb); if (bz_oom) return;
    mag_mul(t.d, a->d, la, b->d, lb);
    int l = la + lb; while (l > 0 && t.d[l - 1] == 0) l--;
    t.size = ((a->size < 0) != (b->size < 0)) ? -l : l;
    bz_swap(r, &t); bz_clear(&t);
}

/* r = r + s*|p| where p has lp limbs, s = +1/-1 applied on top of sign handling */
static void bz_addmag(bz *r, const uint64_t *p, int lp, int psign)
{
    if (!lp) return;
    int lr = bz_len(r);
    if (!lr) { bz_reserve(r, lp); if (bz_oom) return; memcpy(r->d, p, (size_t) lp * 8); r->size = psign < 0 ? -lp : lp; return; }
    int rs = bz_sgn(r);
    if (rs == psign) {                      /* magnitudes add */
        int l = lr > lp ? lr : lp;
        bz_reserve(r, l + 1); if (bz_oom) return;
        uint64_t carry = 0;
        for (int i = 0; i < l; i++) {
            u128 t = (u128)(i < lr ? r->d[i] : 0) + (i < lp ? p[i] : 0) + carry;
            r->d[i] = (uint64_t) t; carry = (uint64_t)(t >> 64);
        }
        if (carry) r->d[l++] = carry;
        r->size = rs < 0 ? -l : l;
    } else {                                /* magnitudes subtract */
        int c = mag_cmp(r->d, lr, p, lp);
        if (c == 0) { r->size = 0; return; }
        const uint64_t *big = c > 0 ? r->d : p, *small = c > 0 ? p : r->d;
        int lb = c > 0 ? lr : lp, ls = c > 0 ? lp : lr;
        bz_reserve(r, lb); if (bz_oom) return;
        if (c > 0) big = r->d; else small = r->d;     /* reserve may have moved r->d */
        uint64_t borrow = 0;
        for (int i = 0; i < lb; i++) {
            uint64_t s = i < ls ? small[i] : 0, b = big[i];
            uint64_t d1 = b - s, br1 = b < s;
            uint64_t d2 = d1 - borrow, br2 = d1 < borrow;
            r->d[i] = d2; borrow = br1 | br2;
        }
        int l = lb; while (l > 0 && r->d[l - 1] == 0) l--;
        int sign = c > 0 ? rs : psign;
        r->size = sign < 0 ? -l : l;
    }
}

/* r = r - a*b  (mpz_submul, SLIP_gmp.c:709) */
static void bz_submul(bz *r, const bz *a, const bz *b)
{
    int la = bz_len(a), lb = bz_len(b);
    if (!la || !lb) return;
    uint64_t *p = (uint64_t *) malloc((size_t)(la + lb) * 8);
    if (!p) { bz_oom = 1; return; }
    mag_mul(p, a->d, la, b->d, lb);
    int lp = la + lb; while (lp > 0 && p[lp - 1] == 0) lp--;
    int psign = ((a->size < 0) != (b->size < 0)) ? -1 : 1;
    bz_addmag(r, p, lp, -psign);
    free(p);
}

static uint64_t inv64(uint64_t d)           /* d odd: d^-1 mod 2^64 (Newton) */
{
    uint64_t x = d;                         /* 3 correct bits */
    for (int i = 0; i < 6; i++) x *= 2 - d * x;
    return x;
}

/* r = a / d, the division being exact (mpz_divexact, SLIP_gmp.c:728).
 * Jebelean's exact division from the least significant limb. */
static void bz_divexact(bz *r, const bz *a, const bz *d)
{
    int la = bz_len(a), ld = bz_len(d);
    if (!la) { r->size = 0; return; }
    /* strip the trailing zeros the divisor has (the dividend has at least as many) */
    int zl = 0; while (d->d[zl] == 0) zl++;
    int zb = __builtin_ctzll(d->d[zl]);
    int ldd = ld - zl, laa = la - zl;
    uint64_t *dd = (uint64_t *) malloc((size_t) ldd * 8), *t = (uint64_t *) malloc((size_t) laa * 8);
    if (!dd || !t) { bz_oom = 1; free(dd); free(t); return; }
    for (int i = 0; i < ldd; i++) {
        uint64_t lo = d->d[zl + i] >> zb;
        uint64_t hi = (zb && i + 1 < ldd) ? d->d[zl + i + 1] << (64 - zb) : 0;
        dd[i] = lo | hi;
    }
    for (int i = 0; i < laa; i++) {
        uint64_t lo = a->d[zl + i] >> zb;
        uint64_t hi = (zb && i + 1 < laa) ? a->d[zl + i + 1] << (64 - zb) : 0;
        t[i] = lo | hi;
    }
    while (ldd > 0 && dd[ldd - 1] == 0) ldd--;
    while (laa > 0 && t[laa - 1] == 0) laa--;
    int lq = laa - ldd + 1;
    if (lq < 1) lq = 1;
    uint64_t dinv = inv64(dd[0]);
    bz q; bz_init(&q); bz_reserve(&q, lq); if (bz_oom) { free(dd); free(t); return; }
    for (int i = 0; i < lq; i++) {
        uint64_t qi = t[i] * dinv;
        q.d[i] = qi;
        /* t[i .. lq) -= qi * dd, only the low lq limbs matter */
        int lim = lq - i < ldd ? lq - i : ldd;
        uint64_t borrow = 0;
        for (int j = 0; j < lim; j++) {
            u128 prod = (u128) qi * dd[j] + borrow;
            uint64_t lo = (uint64_t) prod, hi = (uint64_t)(prod >> 64);
            uint64_t tv = t[i + j];
            t[i + j] = tv - lo;
            borrow = hi + (tv < lo);
        }
        for (int j = i + lim; j < lq && borrow; j++) {
            uint64_t tv = t[j];
            t[j] = tv - borrow;
            borrow = tv < borrow;
        }
    }
    int l = lq; while (l > 0 && q.d[l - 1] == 0) l--;
    q.size = ((a->size < 0) != (d->size < 0)) ? -l : l;
    bz_swap(r, &q); bz_clear(&q);
    free(dd); free(t);
}

/* ------------------------------------------------------------------ */
/* growing CSC of big integers (L and U under construction)            */
/* ------------------------------------------------------------------ */
typedef struct { int64_t nz, cap; int64_t *p; int32_t *i; bz *x; } bcsc;

static int bcsc_init(bcsc *M, int32_t n)
{
    M->nz = 0; M->cap = 4 * (int64_t) n + 16;
    M->p = (int64_t *) calloc((size_t) n + 1, 8);
    M->i = (int32_t *) malloc((size_t) M->cap * 4);
    M->x = (bz *) calloc((size_t) M->cap, sizeof(bz));
    return (M->p && M->i && M->x) ? 0 : -1;
}
static int bcsc_room(bcsc *M, int64_t extra)
{
    if (M->nz + extra <= M->cap) return 0;
    int64_t nc = M->cap; while (nc < M->nz + extra) nc *= 2;
    int32_t *ni = (int32_t *) realloc(M->i, (size_t) nc * 4);
    if (!ni) return -1;
    M->i = ni;
    bz *nx = (bz *) realloc(M->x, (size_t) nc * sizeof(bz));
    if (!nx) return -1;
    memset(nx + M->cap, 0, (size_t)(nc - M->cap) * sizeof(bz));
    M->x = nx; M->cap = nc;
    return 0;
}
static void bcsc_free(bcsc *M)
{
    if (M->x) for (int64_t t = 0; t < M->cap; t++) free(M->x[t].d);
    free(M->x); free(M->i); free(M->p);
}

/* ------------------------------------------------------------------ */
/* reach of A(:,col) in the graph of L: slip_reach.c / slip_dfs.c      */
/* (a visited[] array replaces the reference's sign-flip marks on L->p) */
/* ------------------------------------------------------------------ */
static int32_t reach(int32_t n, int32_t k, const bcsc *L, const int64_t *Ap, const int32_t *Ai,
                     int32_t col, int32_t *xi, int32_t *pstack, const int32_t *pinv, char *mark)
{
    int32_t top = n;
    for (int64_t p = Ap[col]; p < Ap[col + 1]; p++) {
        int32_t start = Ai[p];
        if (mark[start]) continue;
        int32_t head = 0;
        xi[0] = start;
        while (head >= 0) {
            int32_t j = xi[head], jnew = pinv[j];
            /* L(:,jnew) exists only for jnew < k (slip_dfs.c:48-58 reads an empty range otherwise) */
            if (!mark[j]) { mark[j] = 1; pstack[head] = (jnew < k) ? (int32_t) L->p[jnew] : 0; }
            int done = 1;
            int32_t p2 = (jnew < k) ? (int32_t) L->p[jnew + 1] : 0;
            for (int32_t q = pstack[head]; q < p2; q++) {
                int32_t i = L->i[q];
                if (mark[i]) continue;
                pstack[head] = q;
                xi[++head] = i;
                done = 0;
                break;
            }
            if (done) { head--; xi[--top] = j; }
        }
    }
    for (int32_t p = top; p < n; p++) mark[xi[p]] = 0;
    return top;
}

static int cmp_i32(const void *a, const void *b)
{
    int32_t x = *(const int32_t *) a, y = *(const int32_t *) b;
    return (x > y) - (x < y);
}

/* ------------------------------------------------------------------ */
/* exact comparison  |num| / |den|  >=  tol   (slip_get_pivot.c:100-118) */
/* tol is a double, taken exactly (mpq_set_d)                           */
/* ------------------------------------------------------------------ */
static void bz_shl(bz *r, const bz *a, int64_t sh)
{
    int la = bz_len(a);
    if (!la) { r->size = 0; return; }
    int wl = (int)(sh / 64), bl = (int)(sh % 64);
    bz t; bz_init(&t); bz_reserve(&t, la + wl + 1); if (bz_oom) return;
    memset(t.d, 0, (size_t)(la + wl + 1) * 8);
    for (int i = 0; i < la; i++) {
        t.d[i + wl] |= a->d[i] << bl;
        if (bl) t.d[i + wl + 1] |= a->d[i] >> (64 - bl);
    }
    int l = la + wl + 1; while (l > 0 && t.d[l - 1] == 0) l--;
    t.size = l;
    bz_swap(r, &t); bz_clear(&t);
}
static int ratio_ge_tol(const bz *num, const bz *den, double tol)
{
    /* tol = m * 2^e with integer m >= 0 */
    if (!(tol > 0)) return 1;                      /* ratio >= 0 >= tol */
    int e; double fr = frexp(tol, &e);             /* tol = fr * 2^e, fr in [0.5,1) */
    uint64_t m = (uint64_t) ldexp(fr, 53); e -= 53;
    bz mm, lhs, rhs; bz_init(&mm); bz_init(&lhs); bz_init(&rhs);
    uint64_t ml = m; bz_set_limbs(&mm, 1, &ml);
    bz an, ad; an = *num; ad = *den; an.size = bz_len(num); ad.size = bz_len(den);
    /* |num| >= m * 2^e * |den|  */
    bz_mul(&rhs, &mm, &ad);
    if (e >= 0) { bz_shl(&rhs, &rhs, e); bz_set(&lhs, &an); }
    else        { bz_shl(&lhs, &an, -(int64_t) e); }
    int c = bz_cmpabs(&lhs, &rhs);
    bz_clear(&mm); bz_clear(&lhs); bz_clear(&rhs);
    return c >= 0;
}

/* smallest / largest / first eligible entry of the pattern:
 * slip_get_smallest_pivot.c:25-101, slip_get_largest_pivot.c, slip_get_nonzero_pivot.c */
static int32_t pick(int kind, const bz *x, const int32_t *pivs, int32_t n, int32_t top, const int32_t *xi)
{
    int32_t pivot = -1;
    for (int32_t t = top; t < n; t++) {
        int32_t r = xi[t];
        if (pivs[r] >= 0 || bz_sgn(&x[r]) == 0) continue;
        if (pivot < 0) { pivot = r; if (kind == 2) break; continue; }
        int c = bz_cmpabs(&x[pivot], &x[r]);
        if (kind == 0 && c > 0) pivot = r;          /* strictly smaller wins (first stays on ties) */
        if (kind == 1 && c < 0) pivot = r;          /* strictly larger wins */
    }
    return pivot;
}

/* ------------------------------------------------------------------ */
/* the factorisation                                                   */
/* ------------------------------------------------------------------ */
static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void flatten(const bcsc *M, int32_t K, int64_t **p_out, int32_t **i_out, int32_t **len_out,
                    uint64_t **limbs_out, int64_t *nl_out)
{
    int64_t nz = M->nz, nl = 0;
    for (int64_t t = 0; t < nz; t++) nl += bz_len(&M->x[t]);
    int64_t *p = (int64_t *) malloc(((size_t) K + 1) * 8);
    int32_t *ii = (int32_t *) malloc((size_t)(nz ? nz : 1) * 4), *len = (int32_t *) malloc((size_t)(nz ? nz : 1) * 4);
    uint64_t *limbs = (uint64_t *) malloc((size_t)(nl ? nl : 1) * 8);
    for (int32_t k = 0; k <= K; k++) p[k] = k < K ? M->p[k] : nz;
    int64_t o = 0;
    for (int64_t t = 0; t < nz; t++) {
        int l = bz_len(&M->x[t]);
        ii[t] = M->i[t]; len[t] = M->x[t].size;
        if (l) memcpy(limbs + o, M->x[t].d, (size_t) l * 8);
        o += l;
    }
    *p_out = p; *i_out = ii; *len_out = len; *limbs_out = limbs; *nl_out = nl;
}

orc_result *orc_factorize(int32_t n, const int64_t *Ap, const int32_t *Ai,
                          const int32_t *Alen, const uint64_t *Alimbs,
                          const int32_t *q, int32_t pivot_scheme, double tol,
                          int32_t Kmax, int32_t cap)
{
    orc_result *R = (orc_result *) calloc(1, sizeof(orc_result));
    if (!R) return NULL;
    R->n = n;
    if (n <= 0 || !Ap || !Ai || !Alen || !Alimbs || !q) { R->status = ORC_INCORRECT_INPUT; return R; }
    bz_oom = 0;
    int32_t K = (Kmax <= 0 || Kmax > n) ? n : Kmax;
    int64_t annz = Ap[n];

    /* offsets of A's values in the limb slab */
    int64_t *Aoff = (int64_t *) malloc(((size_t) annz + 1) * 8);
    int32_t *pivs = (int32_t *) malloc((size_t) n * 4), *h = (int32_t *) malloc((size_t) n * 4),
            *xi = (int32_t *) malloc((size_t) 2 * n * 4), *row_perm = (int32_t *) malloc((size_t) n * 4),
            *pinv = (int32_t *) malloc((size_t) n * 4);
    char *mark = (char *) calloc((size_t) n, 1);
    bz *x = (bz *) calloc((size_t) n, sizeof(bz)), *rhos = (bz *) calloc((size_t) n, sizeof(bz));
    bcsc L, U; memset(&L, 0, sizeof L); memset(&U, 0, sizeof U);
    if (!Aoff || !pivs || !h || !xi || !row_perm || !pinv || !mark || !x || !rhos ||
        bcsc_init(&L, n) || bcsc_init(&U, n)) { R->status = ORC_OUT_OF_MEMORY; goto done; }
    Aoff[0] = 0;
    for (int64_t p = 0; p < annz; p++) Aoff[p + 1] = Aoff[p] + (Alen[p] < 0 ? -Alen[p] : Alen[p]);
    for (int32_t i = 0; i < n; i++) { pivs[i] = -1; h[i] = -1; pinv[i] = i; row_perm[i] = i; }

    int64_t *C = R->counters;
    double t0 = now_s();
    int32_t k;
    for (k = 0; k < K; k++) {
        int32_t col = q[k];
        L.p[k] = L.nz; U.p[k] = U.nz;
        if (bcsc_room(&L, n) || bcsc_room(&U, n)) { R->status = ORC_OUT_OF_MEMORY; break; }

        /* ---- slip_REF_triangular_solve.c:96-119: pattern, sort, reset, scatter ---- */
        int32_t top = reach(n, k, &L, Ap, Ai, col, xi, xi + n, pinv, mark);
        for (int32_t t = top; t < n; t++) xi[t] = pinv[xi[t]];
        qsort(xi + top, (size_t)(n - top), 4, cmp_i32);
        for (int32_t t = top; t < n; t++) xi[t] = row_perm[xi[t]];
        for (int32_t t = top; t < n; t++) { x[xi[t]].size = 0; h[xi[t]] = -1; }
        x[col].size = 0;
        for (int64_t p = Ap[col]; p < Ap[col + 1]; p++) bz_set_limbs(&x[Ai[p]], Alen[p], Alimbs + Aoff[p]);

        /* ---- slip_REF_triangular_solve.c:124-259: the sweep ---- */
        int64_t c_upd = 0, c_read = 0, c_src = 0, c_str = 0, c_mac = 0;
        for (int64_t p = Ap[col]; p < Ap[col + 1]; p++) c_read += 4 + 8 * (Aoff[p + 1] - Aoff[p]);
        for (int32_t t = top; t < n; t++) {
            int32_t j = xi[t], jnew = pinv[j];
            if (bz_sgn(&x[j]) == 0) continue;
            if (jnew < k) {
                if (h[j] < jnew - 1) {                                   /* :139-149 */
                    bz_mul(&x[j], &x[j], &rhos[jnew - 1]);
                    if (h[j] > -1) bz_divexact(&x[j], &x[j], &rhos[h[j]]);
                }
                c_src++;
                c_read += 8 * bz_len(&rhos[jnew]) + (jnew >= 1 ? 8 * bz_len(&rhos[jnew - 1]) : 0);
                for (int64_t m = L.p[jnew]; m < L.p[jnew + 1]; m++) {    /* :156-241 */
                    int32_t i = L.i[m], inew = pinv[i];
                    c_str++; c_read += 4 + 8 * bz_len(&L.x[m]);
                    if (inew <= jnew) continue;
                    if (bz_sgn(&L.x[m]) == 0) continue;
                    c_upd++;
                    c_mac += (int64_t) bz_len(&L.x[m]) * bz_len(&x[j]) + (int64_t) bz_len(&x[i]) * bz_len(&rhos[jnew]);
                    if (bz_sgn(&x[i]) == 0) {                            /* :175-196 */
                        bz_submul(&x[i], &L.x[m], &x[j]);
                        if (jnew >= 1) bz_divexact(&x[i], &x[i], &rhos[jnew - 1]);
                    } else if (jnew < 1) {                               /* :205-213 */
                        bz_mul(&x[i], &x[i], &rhos[0]);
                        bz_submul(&x[i], &L.x[m], &x[j]);
                    } else {                                             /* :218-237 */
                        if (h[i] < jnew - 1) {
                            bz_mul(&x[i], &x[i], &rhos[jnew - 1]);
                            if (h[i] > -1) bz_divexact(&x[i], &x[i], &rhos[h[i]]);
                        }
                        bz_mul(&x[i], &x[i], &rhos[jnew]);
                        bz_submul(&x[i], &L.x[m], &x[j]);
                        bz_divexact(&x[i], &x[i], &rhos[jnew - 1]);
                    }
                    h[i] = jnew;
                }
            } else if (h[j] < k - 1) {                                   /* :248-257 */
                bz_mul(&x[j], &x[j], &rhos[k - 1]);
                if (h[j] > -1) bz_divexact(&x[j], &x[j], &rhos[h[j]]);
            }
            if (bz_oom) break;
        }
        if (bz_oom) { R->status = ORC_OUT_OF_MEMORY; break; }

        /* ---- column window: stop BEFORE a column holding a value above the cap ---- */
        if (cap > 0) {
            int over = 0;
            for (int32_t t = top; t < n; t++) if (bz_len(&x[xi[t]]) > cap) { over = 1; break; }
            if (over) break;
        }
        C[ORC_N_UPD] += c_upd; C[ORC_B_READ] += c_read; C[ORC_N_SRC] += c_src;
        C[ORC_L_STREAMED] += c_str; C[ORC_LIMB_MACS] += c_mac;

        /* ---- slip_get_pivot.c:58-182 ---- */
        int32_t pivot;
        int diag_ok = bz_sgn(&x[col]) != 0 && pivs[col] < 0;
        switch (pivot_scheme) {
        case 0: pivot = pick(0, x, pivs, n, top, xi); break;
        case 1: pivot = diag_ok ? col : pick(0, x, pivs, n, top, xi); break;
        case 2: pivot = pick(2, x, pivs, n, top, xi); break;
        case 3: pivot = pick(0, x, pivs, n, top, xi);
                if (pivot >= 0 && diag_ok && ratio_ge_tol(&x[pivot], &x[col], tol)) pivot = col;
                break;
        case 4: pivot = pick(1, x, pivs, n, top, xi);
                if (pivot >= 0 && diag_ok && ratio_ge_tol(&x[col], &x[pivot], tol)) pivot = col;
                break;
        default: pivot = pick(1, x, pivs, n, top, xi); break;
        }
        if (pivot < 0) { R->status = ORC_SINGULAR; break; }
        {   /* :164-176 */
            int32_t intermed = pinv[pivot], intermed2 = row_perm[k];
            row_perm[k] = pivot; row_perm[intermed] = intermed2;
            pinv[pivot] = k; pinv[intermed2] = intermed;
            pivs[pivot] = 1;
            bz_set(&rhos[k], &x[pivot]);
        }

        /* ---- SLIP_LU_factorize.c:226-263: split the pattern into U(:,k) and L(:,k) ---- */
        for (int32_t t = top; t < n; t++) {
            int32_t r = xi[t], loc = pinv[r];
            int l = bz_len(&x[r]);
            if (l > C[ORC_MAXLIMBS]) C[ORC_MAXLIMBS] = l;
            if (loc <= k) { U.i[U.nz] = r; bz_set(&U.x[U.nz], &x[r]); U.nz++; C[ORC_B_WRITE] += 4 + 8 * l; }
            if (loc >= k) { L.i[L.nz] = r; bz_set(&L.x[L.nz], &x[r]); L.nz++; C[ORC_B_WRITE] += 4 + 8 * l; }
        }
        C[ORC_B_WRITE] += 8 * bz_len(&rhos[k]);
        if (bz_oom) { R->status = ORC_OUT_OF_MEMORY; break; }
    }
    R->seconds = now_s() - t0;
    R->K = k;
    C[ORC_K_DONE] = k;
    L.p[k] = L.nz; U.p[k] = U.nz;      /* a column interrupted mid-way committed nothing */

    /* ---- results ---- */
    R->lnz = L.nz; R->unz = U.nz;
    flatten(&L, k, &R->Lp, &R->Li, &R->Llen, &R->Llimbs, &R->Lnl);
    flatten(&U, k, &R->Up, &R->Ui, &R->Ulen, &R->Ulimbs, &R->Unl);
    {
        int64_t nl = 0; for (int32_t t = 0; t < k; t++) nl += bz_len(&rhos[t]);
        R->rholen = (int32_t *) malloc((size_t)(k ? k : 1) * 4);
        R->rholimbs = (uint64_t *) malloc((size_t)(nl ? nl : 1) * 8);
        int64_t o = 0;
        for (int32_t t = 0; t < k; t++) {
            int l = bz_len(&rhos[t]);
            R->rholen[t] = rhos[t].size;
            if (l) memcpy(R->rholimbs + o, rhos[t].d, (size_t) l * 8);
            o += l;
        }
        R->rhonl = nl;
    }
    R->pinv = pinv; pinv = NULL;

done:
    if (x) for (int32_t i = 0; i < n; i++) free(x[i].d);
    if (rhos) for (int32_t i = 0; i < n; i++) free(rhos[i].d);
    free(x); free(rhos); free(Aoff); free(pivs); free(h); free(xi); free(row_perm); free(pinv); free(mark);
    bcsc_free(&L); bcsc_free(&U);
    return R;
}

void orc_free(orc_result *r)
{
    if (!r) return;
    free(r->Lp); free(r->Up); free(r->Li); free(r->Ui); free(r->Llen); free(r->Ulen);
    free(r->Llimbs); free(r->Ulimbs); free(r->rholen); free(r->rholimbs); free(r->pinv);
    free(r);
}

/* ------------------------------------------------------------------ */
/* REF triangular solves: SLIP_LU_solve.c:41-86                        */
/* ------------------------------------------------------------------ */
int orc_solve(const orc_result *f, int32_t nrhs, const int32_t *blen, const uint64_t *blimbs,
              int32_t **xlen_out, uint64_t **xlimbs_out, int64_t *xnl_out)
{
    int32_t n = f->n;
    if (f->K != n || f->status != ORC_OK || nrhs <= 0) return ORC_INCORRECT_INPUT;
    bz_oom = 0;
    /* rebuild big-integer views of L, U (permuted row positions), rhos */
    int64_t *Lo = (int64_t *) malloc(((size_t) f->lnz + 1) * 8), *Uo = (int64_t *) malloc(((size_t) f->unz + 1) * 8),
            *Ro = (int64_t *) malloc(((size_t) n + 1) * 8);
    bz *b2 = (bz *) calloc((size_t) n * nrhs, sizeof(bz));
    int32_t *h = (int32_t *) malloc((size_t) n * 4);
    Lo[0] = Uo[0] = Ro[0] = 0;
    for (int64_t t = 0; t < f->lnz; t++) Lo[t + 1] = Lo[t] + abs(f->Llen[t]);
    for (int64_t t = 0; t < f->unz; t++) Uo[t + 1] = Uo[t] + abs(f->Ulen[t]);
    for (int32_t t = 0; t < n; t++) Ro[t + 1] = Ro[t] + abs(f->rholen[t]);
#define VIEW(v, lenarr, off, limbs, t) bz v; v.d = (uint64_t *)(limbs) + (off)[t]; v.size = (lenarr)[t]; v.alloc = 0
    /* b2[pinv[i]] = b[i]   (SLIP_LU_solve.c:68-75) */
    {
        int64_t o = 0;
        for (int32_t c = 0; c < nrhs; c++)
            for (int32_t i = 0; i < n; i++) {
                int32_t sl = blen[(int64_t) c * n + i];
                bz_set_limbs(&b2[(int64_t) c * n + f->pinv[i]], sl, blimbs + o);
                o += sl < 0 ? -sl : sl;
            }
    }
    for (int32_t c = 0; c < nrhs; c++) {
        bz *x = b2 + (int64_t) c * n;
        /* forward substitution, slip_forward_sub.c:61-158 */
        for (int32_t i = 0; i < n; i++) h[i] = -1;
        for (int32_t i = 0; i < n; i++) {
            if (bz_sgn(&x[i]) == 0) continue;
            if (h[i] < i - 1) {
                VIEW(r1, f->rholen, Ro, f->rholimbs, i - 1);
                bz_mul(&x[i], &x[i], &r1);
                if (h[i] > -1) { VIEW(rh, f->rholen, Ro, f->rholimbs, h[i]); bz_divexact(&x[i], &x[i], &rh); }
            }
            for (int64_t m = f->Lp[i]; m < f->Lp[i + 1]; m++) {
                int32_t mnew = f->pinv[f->Li[m]];
                VIEW(lm, f->Llen, Lo, f->Llimbs, m);
                if (bz_sgn(&lm) == 0 || mnew <= i) continue;
                if (bz_sgn(&x[mnew]) == 0) {
                    bz_submul(&x[mnew], &lm, &x[i]);
                    if (i > 0) { VIEW(r1, f->rholen, Ro, f->rholimbs, i - 1); bz_divexact(&x[mnew], &x[mnew], &r1); }
                } else {
                    if (h[mnew] < i - 1) {
                        VIEW(r1, f->rholen, Ro, f->rholimbs, i - 1);
                        bz_mul(&x[mnew], &x[mnew], &r1);
                        if (h[mnew] > -1) { VIEW(rh, f->rholen, Ro, f->rholimbs, h[mnew]); bz_divexact(&x[mnew], &x[mnew], &rh); }
                    }
                    VIEW(ri, f->rholen, Ro, f->rholimbs, i);
                    bz_mul(&x[mnew], &x[mnew], &ri);
                    bz_submul(&x[mnew], &lm, &x[i]);
                    if (i > 0) { VIEW(r1, f->rholen, Ro, f->rholimbs, i - 1); bz_divexact(&x[mnew], &x[mnew], &r1); }
                }
                h[mnew] = i;
            }
        }
        /* x *= det  (slip_array_mul.c:19) */
        { VIEW(det, f->rholen, Ro, f->rholimbs, n - 1); for (int32_t i = 0; i < n; i++) bz_mul(&x[i], &x[i], &det); }
        /* back substitution, slip_back_sub.c:36-52 (pivot is the LAST entry of U(:,j)) */
        for (int32_t j = n - 1; j >= 0; j--) {
            if (bz_sgn(&x[j]) == 0) continue;
            VIEW(ujj, f->Ulen, Uo, f->Ulimbs, f->Up[j + 1] - 1);
            bz_divexact(&x[j], &x[j], &ujj);
            for (int64_t t = f->Up[j]; t < f->Up[j + 1] - 1; t++) {
                VIEW(u, f->Ulen, Uo, f->Ulimbs, t);
                if (bz_sgn(&u) == 0) continue;
                bz_submul(&x[f->pinv[f->Ui[t]]], &u, &x[j]);
            }
        }
    }
#undef VIEW
    int64_t nl = 0;
    for (int64_t t = 0; t < (int64_t) n * nrhs; t++) nl += bz_len(&b2[t]);
    int32_t *xl = (int32_t *) malloc((size_t) n * nrhs * 4);
    uint64_t *xlimbs = (uint64_t *) malloc((size_t)(nl ? nl : 1) * 8);
    int64_t o = 0;
    for (int64_t t = 0; t < (int64_t) n * nrhs; t++) {
        int l = bz_len(&b2[t]);
        xl[t] = b2[t].size;
        if (l) memcpy(xlimbs + o, b2[t].d, (size_t) l * 8);
        o += l;
        free(b2[t].d);
    }
    free(b2); free(h); free(Lo); free(Uo); free(Ro);
    *xlen_out = xl; *xlimbs_out = xlimbs; *xnl_out = nl;
    return bz_oom ? ORC_OUT_OF_MEMORY : ORC_OK;
}

/* ------------------------------------------------------------------ */
/* one IPGE update in isolation, for unit tests of the HIP limb kernels */
/* x <- ((x*hm/hd) * rho_j - l*xj) / rho_p                             */
/* ------------------------------------------------------------------ */
int orc_ipge(int32_t xl, const uint64_t *x, int32_t rjl, const uint64_t *rj,
             int32_t ll, const uint64_t *l, int32_t xjl, const uint64_t *xj,
             int32_t rpl, const uint64_t *rp,
             int32_t hml, const uint64_t *hm, int32_t hdl, const uint64_t *hd,
             int32_t *outlen, uint64_t *out, int32_t outcap)
{
    bz_oom = 0;
    bz X, RJ, Lm, XJ, RP, HM, HD;
    bz_init(&X); bz_init(&RJ); bz_init(&Lm); bz_init(&XJ); bz_init(&RP); bz_init(&HM); bz_init(&HD);
    bz_set_limbs(&X, xl, x); bz_set_limbs(&RJ, rjl, rj); bz_set_limbs(&Lm, ll, l);
    bz_set_limbs(&XJ, xjl, xj); bz_set_limbs(&RP, rpl, rp); bz_set_limbs(&HM, hml, hm); bz_set_limbs(&HD, hdl, hd);
    if (bz_sgn(&HM)) bz_mul(&X, &X, &HM);
    if (bz_sgn(&HD)) bz_divexact(&X, &X, &HD);
    if (bz_sgn(&RJ)) bz_mul(&X, &X, &RJ);
    bz_submul(&X, &Lm, &XJ);
    if (bz_sgn(&RP)) bz_divexact(&X, &X, &RP);
    int n = bz_len(&X), rc = 0;
    if (n > outcap) rc = -1; else { if (n) memcpy(out, X.d, (size_t) n * 8); *outlen = X.size; }
    bz_clear(&X); bz_clear(&RJ); bz_clear(&Lm); bz_clear(&XJ); bz_clear(&RP); bz_clear(&HM); bz_clear(&HD);
    return rc;
}

/* the benchmark's synthetic matrix (slip_matgen.h), so oracle-side tests need no HIP library */
int orc_matgen(int32_t n, double density, int32_t bits, uint64_t seed, int64_t **Ap, int32_t **Ai, int64_t **Ax)
{
    return slip_matgen_csc(n, density, bits, seed, Ap, Ai, Ax);
}
void orc_free_ptr(void *p) { free(p); }
