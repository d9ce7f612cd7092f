/* slip_matgen.h -- deterministic synthetic integer CSC generator (C99, header-only).
 *
 * Produces the "random n-by-n, density d, |a_ij| < 2^b" matrices named by
 * BASELINE.json's configs:
 *   - PRNG: splitmix64(seed)
 *   - the diagonal (j,j) is always present
 *   - round(d*n*n) - n off-diagonal positions drawn uniformly, duplicates
 *     (and diagonal hits) dropped
 *   - values uniform in [1, 2^b) with a random sign, assigned in column-major
 *     (column, then ascending row) order AFTER the pattern is fixed
 * Output is CSC with sorted, duplicate-free columns; values fit int64.
 *
 * The same routine feeds the HIP path, the CPU restatement (oracle/) and the
 * compiled reference (oracle/_ref/ref_driver), so all three factor the
 * identical matrix.
 */
#ifndef SLIP_MATGEN_H
#define SLIP_MATGEN_H

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t s; } slip_rng;

static inline uint64_t slip_rng_next(slip_rng *r)
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static int slip_matgen_cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/* Number of off-diagonal draws the generator makes for (n, density). */
static inline int64_t slip_matgen_draws(int32_t n, double density)
{
    double t = floor(density * (double)n * (double)n + 0.5) - (double)n;
    return t > 0 ? (int64_t)t : 0;
}

/* Generate the matrix.  On success returns 0 and sets *Ap (n+1 int64),
 * *Ai (nnz int32), *Ax (nnz int64); caller frees each with free().
 * Returns -1 on allocation failure or bad arguments. */
static int slip_matgen_csc(int32_t n, double density, int32_t bits, uint64_t seed,
                           int64_t **Ap_out, int32_t **Ai_out, int64_t **Ax_out)
{
    if (n <= 0 || bits < 1 || bits > 62) return -1;
    slip_rng rng; rng.s = seed;
    int64_t draws = slip_matgen_draws(n, density);
    int64_t cap = draws + n;
    uint64_t *key = (uint64_t *)malloc((size_t)cap * sizeof(uint64_t));
    if (!key) return -1;
    int64_t cnt = 0;
    for (int32_t j = 0; j < n; j++) key[cnt++] = ((uint64_t)j << 32) | (uint32_t)j;
    for (int64_t t = 0; t < draws; t++) {
        uint64_t r = slip_rng_next(&rng);
        uint32_t i = (uint32_t)(((r >> 32) * (uint64_t)n) >> 32);
        uint32_t j = (uint32_t)(((r & 0xffffffffULL) * (uint64_t)n) >> 32);
        if (i == j) continue;
        key[cnt++] = ((uint64_t)j << 32) | i;
    }
    qsort(key, (size_t)cnt, sizeof(uint64_t), slip_matgen_cmp_u64);
    int64_t nnz = 0;
    for (int64_t t = 0; t < cnt; t++)
        if (t == 0 || key[t] != key[t - 1]) key[nnz++] = key[t];

    int64_t *Ap = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
    int32_t *Ai = (int32_t *)malloc((size_t)nnz * sizeof(int32_t));
    int64_t *Ax = (int64_t *)malloc((size_t)nnz * sizeof(int64_t));
    if (!Ap || !Ai || !Ax) { free(key); free(Ap); free(Ai); free(Ax); return -1; }
    uint64_t span = ((uint64_t)1 << bits) - 1;      /* values in [1, 2^bits) */
    for (int64_t t = 0; t < nnz; t++) {
        uint32_t j = (uint32_t)(key[t] >> 32);
        Ai[t] = (int32_t)(key[t] & 0xffffffffULL);
        Ap[j + 1]++;
        uint64_t r = slip_rng_next(&rng);
        int64_t mag = (int64_t)(1 + (r >> 1) % span);
        Ax[t] = (r & 1) ? -mag : mag;
    }
    for (int32_t j = 0; j < n; j++) Ap[j + 1] += Ap[j];
    free(key);
    *Ap_out = Ap; *Ai_out = Ai; *Ax_out = Ax;
    return 0;
}

#endif /* SLIP_MATGEN_H */
