/* wave_bigint.h -- wave64-cooperative multi-digit integer primitives (gfx950).
 *
 * One wavefront works on one big integer.  A number is an array of 32-bit
 * digits, little endian (memory-identical to GMP's 64-bit limbs on this
 * little-endian target), living in LDS or in global memory.  Lane t of a
 * 64-digit chunk owns digit 64*r + t; products are formed column-wise (lane =
 * output digit) in a 96-bit per-lane accumulator and the carries are resolved
 * across the wave with one __shfl_up / __ballot carry-lookahead step per
 * chunk -- no serial ripple.
 *
 * All arithmetic the REF-LU update needs is expressed as arithmetic modulo
 * B^W (B = 2^32): the IPGE numerator is only ever needed modulo B^W because
 * the exact division by a pivot is done 2-adically (Jebelean / Hensel):
 *      q = ((t mod 2^(32W+z)) >> z) * inv(d >> z)   (mod B^W),   z = ctz(d)
 * with inv() the inverse of the pivot's odd part modulo B^W, computed once
 * per pivot by Newton iteration and cached.  Signs are carried separately;
 * a final two's-complement test recovers the sign of a difference.
 *
 * These replace the reference's GMP calls mpz_mul / mpz_submul / mpz_divexact
 * (SLIP_LU/Source/SLIP_gmp.c:626,709,728) on the hot path
 * (SLIP_LU/Source/slip_REF_triangular_solve.c:139-257).
 *
 * Every function is called by all 64 lanes of a wave with wave-uniform
 * arguments and ends with the data it wrote visible to the whole wave.
 */
#ifndef SLIP_WAVE_BIGINT_H
#define SLIP_WAVE_BIGINT_H

#include "wave_shim.h"

typedef uint32_t dig_t;

/* Carry-lookahead over one 64-lane chunk.
 * G bit c: lane c generated a carry (into lane c+1); P bit c: lane c holds
 * 0xFFFFFFFF and would propagate an arriving carry.  A generating lane is
 * never propagating (callers guarantee it), so the arrivals are
 *     A = (P + (G << 1)) ^ P
 * (binary addition ripples a generated carry through the run of propagating
 * lanes above it).  *cout = carry out of lane 63. */
SLIP_DEV uint64_t wb_carry_arrivals(uint64_t G, uint64_t P, uint32_t *cout)
{
    uint64_t Gs = G << 1;
    uint64_t sum = P + Gs;
    *cout = (uint32_t)((G >> 63) | (sum < P ? 1u : 0u));
    return sum ^ P;
}

/* out[0..W) = (a[0..la) * b[0..lb)) mod B^W.   out must not overlap a or b. */
SLIP_DEV void wb_mul_lo(dig_t *out, const dig_t *a, int la, const dig_t *b, int lb, int W)
{
    const int lane = slip_lane();
    if (la > W) la = W;
    if (lb > W) lb = W;
    if (la > lb) { const dig_t *t = a; a = b; b = t; int tl = la; la = lb; lb = tl; }
    if (la == 0) {
        for (int c = lane; c < W; c += SLIP_WAVE) out[c] = 0;
        slip_wave_sync();
        return;
    }
    uint32_t cin = 0, pm63 = 0, ph62 = 0, ph63 = 0;
    const int ncols = la + lb;                 /* columns >= ncols-1 hold carries only */
    for (int base = 0; base < W; base += SLIP_WAVE) {
        const int c = base + lane;
        uint32_t lo = 0, mid = 0, hi = 0;
        if (c < W && c < ncols - 1) {
            int ilo = c - lb + 1; if (ilo < 0) ilo = 0;
            int ihi = c < la - 1 ? c : la - 1;
            for (int i = ilo; i <= ihi; i++) {
                uint64_t p = (uint64_t) a[i] * b[c - i];
                uint64_t s = (uint64_t) lo + (uint32_t) p;
                lo = (uint32_t) s;
                s = (uint64_t) mid + (uint32_t)(p >> 32) + (uint32_t)(s >> 32);
                mid = (uint32_t) s;
                hi += (uint32_t)(s >> 32);
            }
        }
        /* digit c collects lo_c + mid_{c-1} + hi_{c-2} */
        uint32_t m1 = slip_shfl_up_u32(mid, 1);
        uint32_t h2 = slip_shfl_up_u32(hi, 2);
        if (lane == 0) { m1 = pm63; h2 = ph62; }
        if (lane == 1) { h2 = ph63; }
        uint64_t s = (uint64_t) lo + m1 + h2 + (lane == 0 ? cin : 0u);
        uint32_t d = (uint32_t) s, e = (uint32_t)(s >> 32);          /* e <= 3 */
        uint32_t e1 = slip_shfl_up_u32(e, 1);
        if (lane == 0) e1 = 0;
        uint64_t s2 = (uint64_t) d + e1;
        uint32_t d2 = (uint32_t) s2, g = (uint32_t)(s2 >> 32);       /* g => d2 <= 2 */
        uint64_t G = slip_ballot(g != 0), P = slip_ballot(d2 == 0xFFFFFFFFu);
        uint32_t cout;
        uint64_t A = wb_carry_arrivals(G, P, &cout);
        if (c < W) out[c] = d2 + (uint32_t)((A >> lane) & 1);
        pm63 = slip_shfl_u32(mid, 63);
        ph62 = slip_shfl_u32(hi, 62);
        ph63 = slip_shfl_u32(hi, 63);
        cin = slip_shfl_u32(e, 63) + cout;
    }
    slip_wave_sync();
}

/* out[0..W) = (x +/- y) mod B^W, x and y zero-extended from lx, ly digits.
 * out may be x or y themselves (same indexing), not a shifted alias.
 * x == NULL: x is the one-digit constant x0. */
SLIP_DEV void wb_addsub(dig_t *out, const dig_t *x, int lx, const dig_t *y, int ly, int W, int sub,
                        uint32_t x0 = 0)
{
    const int lane = slip_lane();
    uint32_t cin = sub ? 1u : 0u;
    for (int base = 0; base < W; base += SLIP_WAVE) {
        const int c = base + lane;
        uint32_t xv = x ? ((c < lx && c < W) ? x[c] : 0u) : (c == 0 ? x0 : 0u);
        uint32_t yv = (c < ly && c < W) ? y[c] : 0u;
        if (sub && c < W) yv = ~yv;
        uint64_t s = (uint64_t) xv + yv + (lane == 0 ? cin : 0u);
        uint32_t d = (uint32_t) s, g = (uint32_t)(s >> 32);
        /* lane 0 with cin may both generate and hold all-ones; it receives no
         * arrival itself, so the lookahead identity still holds */
        uint64_t G = slip_ballot(g != 0), P = slip_ballot(d == 0xFFFFFFFFu && c < W);
        uint32_t cout;
        uint64_t A = wb_carry_arrivals(G, P, &cout);
        if (c < W) out[c] = d + (uint32_t)((A >> lane) & 1);
        cin = cout;
    }
    slip_wave_sync();
}

/* dst[0..W) = (src >> shift)[0..W), src zero-extended from ls digits; no overlap */
SLIP_DEV void wb_copy_shr(dig_t *dst, const dig_t *src, int ls, int shift, int W)
{
    const int lane = slip_lane();
    const int sw = shift >> 5, sb = shift & 31;
    for (int c = lane; c < W; c += SLIP_WAVE) {
        int idx = c + sw;
        uint32_t lo = idx < ls ? src[idx] : 0u;
        uint32_t hi = idx + 1 < ls ? src[idx + 1] : 0u;
        dst[c] = sb ? ((lo >> sb) | (hi << (32 - sb))) : lo;
    }
    slip_wave_sync();
}

/* dst[0..W) = (src << shift)[0..W), src zero-extended from ls digits; no overlap */
SLIP_DEV void wb_copy_shl(dig_t *dst, const dig_t *src, int ls, int shift, int W)
{
    const int lane = slip_lane();
    const int sw = shift >> 5, sb = shift & 31;
    for (int c = lane; c < W; c += SLIP_WAVE) {
        int idx = c - sw;
        uint32_t lo = (idx >= 0 && idx < ls) ? src[idx] : 0u;
        uint32_t below = (idx - 1 >= 0 && idx - 1 < ls) ? src[idx - 1] : 0u;
        dst[c] = sb ? ((lo << sb) | (below >> (32 - sb))) : lo;
    }
    slip_wave_sync();
}

/* number of significant digits of x[0..W) */
SLIP_DEV int wb_len(const dig_t *x, int W)
{
    const int lane = slip_lane();
    for (int base = ((W - 1) >> 6) << 6; base >= 0; base -= SLIP_WAVE) {
        const int c = base + lane;
        uint64_t nz = slip_ballot(c < W && x[c] != 0);
        if (nz) return base + 64 - slip_clz64(nz);
    }
    return 0;
}

/* number of trailing zero bits of x (la >= 1 significant digits, x != 0) */
SLIP_DEV int wb_ctz(const dig_t *x, int la)
{
    const int lane = slip_lane();
    for (int base = 0; base < la; base += SLIP_WAVE) {
        const int c = base + lane;
        uint32_t v = c < la ? x[c] : 0u;
        uint64_t nz = slip_ballot(v != 0);
        if (nz) {
            int t = slip_ctz64(nz);
            uint32_t w = slip_shfl_u32(v, t);
            return 32 * (base + t) + slip_ctz32(w);
        }
    }
    return 0;
}

/* bit length of a normalised la-digit number */
SLIP_DEV int wb_bits(const dig_t *x, int la)
{
    if (la == 0) return 0;
    return 32 * la - slip_clz32(x[la - 1]);
}

/* compare magnitudes of normalised numbers: -1, 0, +1 */
SLIP_DEV int wb_cmp(const dig_t *a, int la, const dig_t *b, int lb)
{
    if (la != lb) return la > lb ? 1 : -1;
    const int lane = slip_lane();
    for (int base = ((la - 1) >> 6) << 6; base >= 0 && la > 0; base -= SLIP_WAVE) {
        const int c = base + lane;
        uint32_t av = c < la ? a[c] : 0u, bv = c < la ? b[c] : 0u;
        uint64_t df = slip_ballot(av != bv);
        if (df) {
            int t = 63 - slip_clz64(df);
            uint32_t at = slip_shfl_u32(av, t), bt = slip_shfl_u32(bv, t);
            return at > bt ? 1 : -1;
        }
    }
    return 0;
}

/* copy la digits, append one zero digit when la is odd (whole 64-bit limbs) */
SLIP_DEV void wb_copy_pad(dig_t *dst, const dig_t *src, int la)
{
    const int lane = slip_lane();
    const int lw = (la + 1) & ~1;
    for (int c = lane; c < lw; c += SLIP_WAVE) dst[c] = c < la ? src[c] : 0u;
    slip_wave_sync();
}

/* inverse of an odd 32-bit digit modulo 2^32 */
SLIP_DEV uint32_t wb_inv32(uint32_t d)
{
    uint32_t x = d;                          /* correct to 3 bits */
    x *= 2u - d * x; x *= 2u - d * x; x *= 2u - d * x; x *= 2u - d * x;
    return x;
}

/* Extend inv (an inverse of the odd number dodd modulo B^have, have >= 0) to
 * modulo B^want by Newton steps  v <- v * (2 - d*v).  e, t: two scratch
 * buffers of `want` digits.  The low `have` digits of inv are not rewritten. */
SLIP_DEV void wb_inv_extend(dig_t *inv, int have, int want, const dig_t *dodd, int ld, dig_t *e, dig_t *t)
{
    const int lane = slip_lane();
    if (have == 0) {
        if (lane == 0) inv[0] = wb_inv32(dodd[0]);
        slip_wave_sync();
        have = 1;
    }
    while (have < want) {
        /* halving chain down from `want`: the last (most expensive) step doubles exactly onto the target
         * instead of repeating a full-length step for a few extra digits */
        int m2 = want;
        while (m2 > 2 * have) m2 = (m2 + 1) >> 1;
        wb_mul_lo(e, dodd, ld < m2 ? ld : m2, inv, have, m2);
        wb_addsub(e, (const dig_t *) 0, 1, e, m2, m2, 1, 2u);      /* e = 2 - d*v */
        wb_mul_lo(t, inv, have, e, m2, m2);
        for (int c = have + lane; c < m2; c += SLIP_WAVE) inv[c] = t[c];
        slip_wave_sync();
        have = m2;
    }
}

#endif /* SLIP_WAVE_BIGINT_H */
