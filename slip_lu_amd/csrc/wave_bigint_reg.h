/* wave_bigint_reg.h -- register-resident wave64 big-integer arithmetic (gfx950).
 *
 * Same arithmetic as wave_bigint.h (everything modulo B^W, B = 2^32; exact
 * division by a pivot = multiplication by the cached 2-adic inverse), but the
 * numbers stay in VGPRs: a WR<D> holds up to 64*D digits, lane l owning digits
 * l, 64+l, 128+l, ...  The schoolbook product keeps one operand as a
 * wave-wide shift register: at step i every lane multiplies the broadcast
 * digit a_i (v_readlane -> SGPR) by the B digit that has been shifted i lanes
 * towards it (v_mov_dpp wave_shr:1, one VALU op per chunk and step, no LDS
 * crossbar, no memory traffic in the loop) and accumulates into a 96-bit
 * per-lane column sum (v_mad_u64_u32 + carry).  Carries are resolved per
 * 64-digit chunk with DPP shifts and one ballot carry-lookahead.
 *
 * Replaces mpz_mul / mpz_submul / mpz_divexact (SLIP_LU/Source/SLIP_gmp.c:626,709,728)
 * for the multi-limb updates of slip_REF_triangular_solve.c:139-257.
 */
#ifndef SLIP_WAVE_BIGINT_REG_H
#define SLIP_WAVE_BIGINT_REG_H

#include "wave_bigint.h"

template <int D> struct WR { uint32_t d[D]; };

template <int D> SLIP_DEV WR<D> wr_zero(void)
{
    WR<D> x;
#pragma unroll
    for (int q = 0; q < D; q++) x.d[q] = 0;
    return x;
}

/* digits [0, len) of p (len is clipped to 64*D), zero above */
template <int D> SLIP_DEV WR<D> wr_load(const dig_t *p, int len)
{
    const int lane = slip_lane();
    WR<D> x;
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; x.d[q] = c < len ? p[c] : 0u; }
    return x;
}

/* write digits [0, count) (count <= 64*D) */
template <int D> SLIP_DEV void wr_store(dig_t *p, const WR<D> &x, int count)
{
    const int lane = slip_lane();
#pragma unroll
    for (int q = 0; q < D; q++) { const int c = 64 * q + lane; if (c < count) p[c] = x.d[q]; }
}

/* keep digits [0, W), clear the rest */
template <int D> SLIP_DEV WR<D> wr_mask(const WR<D> &x, int W)
{
    const int lane = slip_lane();
    WR<D> r;
#pragma unroll
    for (int q = 0; q < D; q++) r.d[q] = (64 * q + lane < W) ? x.d[q] : 0u;
    return r;
}

/* number of significant digits */
template <int D> SLIP_DEV int wr_len(const WR<D> &x)
{
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {
        const uint64_t nz = slip_ballot(x.d[q] != 0);
        if (nz) return 64 * q + 64 - slip_clz64(nz);
    }
    return 0;
}

/* digit c of x as a wave-uniform value */
template <int D> SLIP_DEV uint32_t wr_digit(const WR<D> &x, int c)
{
    uint32_t v = 0;
#pragma unroll
    for (int q = 0; q < D; q++) if ((c >> 6) == q) v = slip_readlane(x.d[q], c & 63);
    return v;
}

/* column sums (lo,mid,hi per lane and chunk) -> digits */
template <int D> SLIP_DEV WR<D> wr_normalise(const uint32_t *lo, const uint32_t *mid, const uint32_t *hi)
{
    const int lane = slip_lane();
    WR<D> out;
    uint32_t cin = 0, pm63 = 0, ph62 = 0, ph63 = 0;
#pragma unroll
    for (int r = 0; r < D; r++) {
        /* digit c collects lo_c + mid_{c-1} + hi_{c-2} */
        const uint32_t m1 = slip_dpp_shr1(mid[r], pm63);
        const uint32_t h1 = slip_dpp_shr1(hi[r], ph63);
        const uint32_t h2 = slip_dpp_shr1(h1, ph62);
        const uint64_t s = (uint64_t) lo[r] + m1 + h2 + (lane == 0 ? cin : 0u);
        const uint32_t dg = (uint32_t) s, e = (uint32_t)(s >> 32);          /* e <= 3 */
        const uint32_t e1 = slip_dpp_shr1(e, 0u);
        const uint64_t s2 = (uint64_t) dg + e1;
        const uint32_t d2 = (uint32_t) s2, g = (uint32_t)(s2 >> 32);        /* g => d2 <= 2 */
        const uint64_t G = slip_ballot(g != 0), Pm = slip_ballot(d2 == 0xFFFFFFFFu);
        uint32_t cout;
        const uint64_t A = wb_carry_arrivals(G, Pm, &cout);
        out.d[r] = d2 + (uint32_t)((A >> lane) & 1);
        pm63 = slip_readlane(mid[r], 63);
        ph62 = slip_readlane(hi[r], 62);
        ph63 = slip_readlane(hi[r], 63);
        cin = slip_readlane(e, 63) + cout;
    }
    return out;
}

/* (A * B) mod B^(64*D).  Only the first la digits of A are walked (la <= 64*D);
 * call with A = the operand with fewer digits. */
template <int D> SLIP_DEV WR<D> wr_mul(const WR<D> &A, int la, const WR<D> &B)
{
    uint64_t acc[D];
    uint32_t hi[D], Bs[D];
#pragma unroll
    for (int r = 0; r < D; r++) { acc[r] = 0; hi[r] = 0; Bs[r] = B.d[r]; }
#pragma unroll
    for (int ia = 0; ia < D; ia++) {
        int steps = la - 64 * ia;
        if (steps > 64) steps = 64;
        for (int il = 0; il < steps; il++) {
            const uint32_t a = slip_readlane(A.d[ia], il);
            /* low product: after 64*ia steps the chunks below ia of the shift register hold only zeros
             * (column c takes B[c - i], i >= 64*ia), so they neither multiply nor shift */
#pragma unroll
            for (int r = ia; r < D; r++) slip_mac96(acc[r], hi[r], a, Bs[r]);
            /* shift B one lane up across the chunks: lane l of chunk r now holds B[64r + l - (i+1)] */
#pragma unroll
            for (int r = D - 1; r >= ia; r--) {
                if (r > ia) Bs[r] = slip_dpp_shr1_in(Bs[r], slip_readlane(Bs[r - 1], 63));
                else Bs[r] = slip_dpp_shr1_zero(Bs[r]);
            }
        }
    }
    slip_valu_settle();
    uint32_t lo[D], mid[D];
#pragma unroll
    for (int r = 0; r < D; r++) { lo[r] = (uint32_t) acc[r]; mid[r] = (uint32_t)(acc[r] >> 32); }
    return wr_normalise<D>(lo, mid, hi);
}

/* a * M for a one-digit wave-uniform a (the product must fit 64*D digits): digit c = lo(a*M[c]) + hi(a*M[c-1]) + carry,
 * one multiply, one DPP shift and one ballot carry-lookahead per chunk */
template <int D> SLIP_DEV WR<D> wr_mul_digit(uint32_t a, const WR<D> &M)
{
    const int lane = slip_lane();
    WR<D> out;
    uint32_t cin = 0, hi63 = 0;
#pragma unroll
    for (int r = 0; r < D; r++) {
        const uint64_t p = (uint64_t) a * M.d[r];
        const uint32_t lo = (uint32_t) p, hi = (uint32_t)(p >> 32);
        const uint32_t hprev = slip_dpp_shr1(hi, hi63);
        const uint64_t s = (uint64_t) lo + hprev + (lane == 0 ? cin : 0u);
        const uint32_t dg = (uint32_t) s, g = (uint32_t)(s >> 32);
        const uint64_t G = slip_ballot(g != 0), Pm = slip_ballot(dg == 0xFFFFFFFFu);
        uint32_t cout;
        const uint64_t A = wb_carry_arrivals(G, Pm, &cout);
        out.d[r] = dg + (uint32_t)((A >> lane) & 1);
        hi63 = slip_readlane(hi, 63);
        cin = cout;
    }
    return out;
}

/* two independent one-digit products against the same M, chunk by chunk side by side: the two carry chains
 * (DPP shift -> add -> ballots -> scalar look-ahead) overlap instead of running back to back */
template <int D> SLIP_DEV void wr_mul_digit2(uint32_t a0, uint32_t a1, const WR<D> &M, WR<D> &out0, WR<D> &out1)
{
    const int lane = slip_lane();
    uint32_t cin0 = 0, cin1 = 0, h0 = 0, h1 = 0;
#pragma unroll
    for (int r = 0; r < D; r++) {
        const uint64_t p0 = (uint64_t) a0 * M.d[r], p1 = (uint64_t) a1 * M.d[r];
        const uint32_t lo0 = (uint32_t) p0, hi0 = (uint32_t)(p0 >> 32), lo1 = (uint32_t) p1, hi1 = (uint32_t)(p1 >> 32);
        const uint32_t q0 = slip_dpp_shr1(hi0, h0), q1 = slip_dpp_shr1(hi1, h1);
        const uint64_t s0 = (uint64_t) lo0 + q0 + (lane == 0 ? cin0 : 0u), s1 = (uint64_t) lo1 + q1 + (lane == 0 ? cin1 : 0u);
        const uint32_t d0 = (uint32_t) s0, g0 = (uint32_t)(s0 >> 32), d1 = (uint32_t) s1, g1 = (uint32_t)(s1 >> 32);
        const uint64_t G0 = slip_ballot(g0 != 0), P0 = slip_ballot(d0 == 0xFFFFFFFFu);
        const uint64_t G1 = slip_ballot(g1 != 0), P1 = slip_ballot(d1 == 0xFFFFFFFFu);
        uint32_t c0, c1;
        const uint64_t A0 = wb_carry_arrivals(G0, P0, &c0), A1 = wb_carry_arrivals(G1, P1, &c1);
        out0.d[r] = d0 + (uint32_t)((A0 >> lane) & 1);
        out1.d[r] = d1 + (uint32_t)((A1 >> lane) & 1);
        h0 = slip_readlane(hi0, 63); h1 = slip_readlane(hi1, 63);
        cin0 = c0; cin1 = c1;
    }
}

/* x +/- y modulo B^(64*D) */
template <int D> SLIP_DEV WR<D> wr_addsub(const WR<D> &x, const WR<D> &y, int sub)
{
    const int lane = slip_lane();
    WR<D> out;
    uint32_t cin = sub ? 1u : 0u;
#pragma unroll
    for (int r = 0; r < D; r++) {
        const uint32_t yv = sub ? ~y.d[r] : y.d[r];
        const uint64_t s = (uint64_t) x.d[r] + yv + (lane == 0 ? cin : 0u);
        const uint32_t dg = (uint32_t) s, g = (uint32_t)(s >> 32);
        const uint64_t G = slip_ballot(g != 0), Pm = slip_ballot(dg == 0xFFFFFFFFu);
        uint32_t cout;
        const uint64_t A = wb_carry_arrivals(G, Pm, &cout);
        out.d[r] = dg + (uint32_t)((A >> lane) & 1);
        cin = cout;
    }
    return out;
}

/* (x >> shift) modulo B^(64*D); shift < 32: pure DPP; larger shifts go through `scratch` (64*D+2 digits) */
template <int D> SLIP_DEV WR<D> wr_shr(const WR<D> &x, int shift, dig_t *scratch)
{
    if (shift == 0) return x;
    WR<D> out;
    if (shift < 32) {
#pragma unroll
        for (int q = 0; q < D; q++) {
            uint32_t fill = 0;
            if (q + 1 < D) fill = slip_readlane(x.d[q + 1], 0);
            const uint32_t nxt = slip_dpp_shl1(x.d[q], fill);
            out.d[q] = (x.d[q] >> shift) | (nxt << (32 - shift));
        }
        return out;
    }
    const int lane = slip_lane();
    slip_wave_sync();
    wr_store<D>(scratch, x, 64 * D);
    if (lane < 2) scratch[64 * D + lane] = 0;
    slip_wave_sync();
    const int sw = shift >> 5, sb = shift & 31;
#pragma unroll
    for (int q = 0; q < D; q++) {
        const int idx = 64 * q + lane + sw;
        const uint32_t a = idx < 64 * D ? scratch[idx] : 0u, b = idx + 1 < 64 * D ? scratch[idx + 1] : 0u;
        out.d[q] = sb ? ((a >> sb) | (b << (32 - sb))) : a;
    }
    slip_wave_sync();
    return out;
}

/* Extend the inverse V of the odd number Dodd from `have` (>= 0) to `want` digits (Newton):
 *   v <- v * (2 - d*v)  modulo B^(2m).  want <= 64*D. */
template <int D> SLIP_DEV WR<D> wr_inv_extend(WR<D> V, int have, int want, const WR<D> &Dodd)
{
    const int lane = slip_lane();
    if (have == 0) {
        const uint32_t d0 = slip_readlane(Dodd.d[0], 0);
        V = wr_zero<D>();
        if (lane == 0) V.d[0] = wb_inv32(d0);
        have = 1;
    }
    WR<D> two = wr_zero<D>();
    if (lane == 0) two.d[0] = 2u;
    while (have < want) {
        /* halving chain down from `want` (see wb_inv_extend): 1 -> 2 -> 3 -> 5 -> ... -> want/2 -> want */
        int m2 = want;
        while (m2 > 2 * have) m2 = (m2 + 1) >> 1;
        WR<D> e = wr_mul<D>(V, have, Dodd);              /* d*v, V's `have` digits walked */
        e = wr_mask<D>(wr_addsub<D>(two, e, 1), m2);     /* 2 - d*v  mod B^m2 */
        V = wr_mask<D>(wr_mul<D>(V, have, e), m2);
        have = m2;
    }
    return V;
}

/* Exact division without an inverse (Hensel, digit-serial): Q with Q * Dodd = T (mod B^W), Dodd odd, W <= 64*D.
 * The quotient digits come out one per step: column i of the running sum  sum_{i' < i} q_i' * Dodd * B^i'  is complete
 * when step i starts, so q_i = (t_i - column_i) * d0^{-1} mod B with d0^{-1} mod B a five-instruction scalar Newton; then
 * every lane adds q_i * d_{c-i} into its column (the same wave-wide shift register as wr_mul).  The carry from column to
 * column travels through the scalar unit (three v_readlane, a dozen s_ instructions per step), so nothing is normalised
 * and no 2-adic inverse of W digits is ever formed: 2.5 product-equivalents against the 3+ of Newton + product, with a
 * tenth of the fixed costs (no wr_normalise, no masks).  Used for divisors nobody else divides by (the rho[h] of a row's
 * history: mpz_divexact at slip_REF_triangular_solve.c:147,226,255); sources' rho[j-1] keep their cached inverses. */
template <int D> SLIP_DEV WR<D> wr_div_hensel(const WR<D> &T, int W, const WR<D> &Dodd)
{
    const uint32_t d0 = slip_readlane(Dodd.d[0], 0);
    uint32_t inv = d0;                                   /* d0 * d0 = 1 mod 8 */
    inv *= 2u - d0 * inv; inv *= 2u - d0 * inv; inv *= 2u - d0 * inv; inv *= 2u - d0 * inv;
    uint64_t acc[D];
    uint32_t hi[D], Ds[D];
    WR<D> Q;
#pragma unroll
    for (int r = 0; r < D; r++) { acc[r] = 0; hi[r] = 0; Ds[r] = Dodd.d[r]; Q.d[r] = 0; }
    uint32_t c0 = 0, c1 = 0;                             /* the carry into the column of the current step (wave-uniform) */
#pragma unroll
    for (int ia = 0; ia < D; ia++) {
        int steps = W - 64 * ia;
        if (steps > 64) steps = 64;
        for (int il = 0; il < steps; il++) {
            /* column i = what the earlier digits left there + the carry; q_i makes its low word t_i */
            const uint32_t s0 = slip_readlane((uint32_t) acc[ia], il), t = slip_readlane(T.d[ia], il);
            const uint32_t q = slip_uniform((t - (s0 + c0)) * inv);
            Q.d[ia] = slip_writelane(Q.d[ia], q, il);
            /* column c >= i takes q_i * d[c - i] (lane l of chunk r holds d[64r + l - i] at step i, as in wr_mul) */
#pragma unroll
            for (int r = ia; r < D; r++) slip_mac96(acc[r], hi[r], q, Ds[r]);
            /* lane i now holds column i without the carry that came in: low word + c0 = t_i (mod B), so the carry out of the
             * low word is 1 exactly when t_i < c0; the upper words go on to column i + 1 */
            const uint32_t m1 = slip_readlane((uint32_t)(acc[ia] >> 32), il), m2 = slip_readlane(hi[ia], il);
            slip_carry_step(m1, c1, t, c0, m2, c0, c1);
#pragma unroll
            for (int r = D - 1; r >= ia; r--) {
                if (r > ia) Ds[r] = slip_dpp_shr1_in(Ds[r], slip_readlane(Ds[r - 1], 63));
                else Ds[r] = slip_dpp_shr1_zero(Ds[r]);
            }
        }
    }
    slip_valu_settle();
    return Q;
}

#endif /* SLIP_WAVE_BIGINT_REG_H */
