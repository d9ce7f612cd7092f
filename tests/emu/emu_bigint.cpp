/* emu_bigint.cpp -- CPU (fiber-emulated wave64) entry points for unit tests of
 * slip_lu_amd/csrc/wave_bigint.h.  TEST ONLY. */
#define SLIP_EMULATE 1
#include "../../slip_lu_amd/csrc/wave_bigint.h"
#include <string.h>

extern "C" {

void emu_mul_lo(const uint32_t *a, int la, const uint32_t *b, int lb, int W, uint32_t *out)
{
    emu::launch(1, 64, [&]() { wb_mul_lo(out, a, la, b, lb, W); });
}

void emu_addsub(const uint32_t *x, int lx, const uint32_t *y, int ly, int W, int sub, uint32_t *out)
{
    emu::launch(1, 64, [&]() { wb_addsub(out, x, lx, y, ly, W, sub); });
}

void emu_shr(const uint32_t *x, int lx, int shift, int W, uint32_t *out)
{
    emu::launch(1, 64, [&]() { wb_copy_shr(out, x, lx, shift, W); });
}

void emu_shl(const uint32_t *x, int lx, int shift, int W, uint32_t *out)
{
    emu::launch(1, 64, [&]() { wb_copy_shl(out, x, lx, shift, W); });
}

int emu_len(const uint32_t *x, int W)
{
    int r = -1;
    emu::launch(1, 64, [&]() { int v = wb_len(x, W); if (slip_lane() == 0) r = v; });
    return r;
}

int emu_ctz(const uint32_t *x, int la)
{
    int r = -1;
    emu::launch(1, 64, [&]() { int v = wb_ctz(x, la); if (slip_lane() == 0) r = v; });
    return r;
}

int emu_cmp(const uint32_t *a, int la, const uint32_t *b, int lb)
{
    int r = -2;
    emu::launch(1, 64, [&]() { int v = wb_cmp(a, la, b, lb); if (slip_lane() == 17) r = v; });
    return r;
}

/* inverse of odd d modulo B^want, starting from `have` digits already in inv */
void emu_inv(const uint32_t *d, int ld, int have, int want, uint32_t *inv)
{
    uint32_t *e = new uint32_t[want + 1], *t = new uint32_t[want + 1];
    emu::launch(1, 64, [&]() { wb_inv_extend(inv, have, want, d, ld, e, t); });
    delete[] e; delete[] t;
}

}

/* ---- register-resident variants (wave_bigint_reg.h) ---- */
#include "../../slip_lu_amd/csrc/wave_bigint_reg.h"

template <int D> static void reg_mul(const uint32_t *a, int la, const uint32_t *b, int lb, int W, uint32_t *out)
{
    emu::launch(1, 64, [&]() {
        WR<D> A = wr_load<D>(a, la), B = wr_load<D>(b, lb);
        WR<D> P = la <= lb ? wr_mul<D>(A, la, B) : wr_mul<D>(B, lb, A);
        wr_store<D>(out, P, W);
    });
}
template <int D> static void reg_addsub(const uint32_t *a, int la, const uint32_t *b, int lb, int W, int sub, uint32_t *out)
{
    emu::launch(1, 64, [&]() { wr_store<D>(out, wr_addsub<D>(wr_load<D>(a, la), wr_load<D>(b, lb), sub), W); });
}
template <int D> static void reg_shr(const uint32_t *a, int la, int shift, int W, uint32_t *out)
{
    uint32_t *scr = new uint32_t[64 * D + 2];
    emu::launch(1, 64, [&]() { wr_store<D>(out, wr_shr<D>(wr_load<D>(a, la), shift, scr), W); });
    delete[] scr;
}
template <int D> static void reg_inv(const uint32_t *d, int ld, int have, int want, uint32_t *inv)
{
    emu::launch(1, 64, [&]() {
        WR<D> V = wr_load<D>(inv, have);
        V = wr_inv_extend<D>(V, have, want, wr_load<D>(d, ld));
        wr_store<D>(inv, V, want);
    });
}
template <int D> static void reg_mul_digit(uint32_t a, const uint32_t *b, int lb, int W, uint32_t *out)
{
    emu::launch(1, 64, [&]() { wr_store<D>(out, wr_mul_digit<D>(a, wr_load<D>(b, lb)), W); });
}
template <int D> static void reg_mul_digit2(uint32_t a0, uint32_t a1, const uint32_t *b, int lb, int W, uint32_t *out0, uint32_t *out1)
{
    emu::launch(1, 64, [&]() {
        WR<D> Y0, Y1;
        wr_mul_digit2<D>(a0, a1, wr_load<D>(b, lb), Y0, Y1);
        wr_store<D>(out0, Y0, W); wr_store<D>(out1, Y1, W);
    });
}
template <int D> static int reg_len(const uint32_t *a, int la)
{
    int r = -1;
    emu::launch(1, 64, [&]() { int v = wr_len<D>(wr_load<D>(a, la)); if (slip_lane() == 5) r = v; });
    return r;
}

#define DISPATCH(D, call) switch (D) { case 1: call<1>; break; case 2: call<2>; break; case 3: call<3>; break; default: call<4>; break; }

extern "C" {
void emu_reg_mul(int D, const uint32_t *a, int la, const uint32_t *b, int lb, int W, uint32_t *out)
{ switch (D) { case 1: reg_mul<1>(a, la, b, lb, W, out); break; case 2: reg_mul<2>(a, la, b, lb, W, out); break;
               case 3: reg_mul<3>(a, la, b, lb, W, out); break; default: reg_mul<4>(a, la, b, lb, W, out); } }
void emu_reg_addsub(int D, const uint32_t *a, int la, const uint32_t *b, int lb, int W, int sub, uint32_t *out)
{ switch (D) { case 1: reg_addsub<1>(a, la, b, lb, W, sub, out); break; case 2: reg_addsub<2>(a, la, b, lb, W, sub, out); break;
               case 3: reg_addsub<3>(a, la, b, lb, W, sub, out); break; default: reg_addsub<4>(a, la, b, lb, W, sub, out); } }
void emu_reg_shr(int D, const uint32_t *a, int la, int shift, int W, uint32_t *out)
{ switch (D) { case 1: reg_shr<1>(a, la, shift, W, out); break; case 2: reg_shr<2>(a, la, shift, W, out); break;
               case 3: reg_shr<3>(a, la, shift, W, out); break; default: reg_shr<4>(a, la, shift, W, out); } }
void emu_reg_inv(int D, const uint32_t *d, int ld, int have, int want, uint32_t *inv)
{ switch (D) { case 1: reg_inv<1>(d, ld, have, want, inv); break; case 2: reg_inv<2>(d, ld, have, want, inv); break;
               case 3: reg_inv<3>(d, ld, have, want, inv); break; default: reg_inv<4>(d, ld, have, want, inv); } }
void emu_reg_mul_digit(int D, uint32_t a, const uint32_t *b, int lb, int W, uint32_t *out)
{ switch (D) { case 1: reg_mul_digit<1>(a, b, lb, W, out); break; case 2: reg_mul_digit<2>(a, b, lb, W, out); break;
               case 3: reg_mul_digit<3>(a, b, lb, W, out); break; default: reg_mul_digit<4>(a, b, lb, W, out); } }
void emu_reg_mul_digit2(int D, uint32_t a0, uint32_t a1, const uint32_t *b, int lb, int W, uint32_t *out0, uint32_t *out1)
{ switch (D) { case 1: reg_mul_digit2<1>(a0, a1, b, lb, W, out0, out1); break; case 2: reg_mul_digit2<2>(a0, a1, b, lb, W, out0, out1); break;
               case 3: reg_mul_digit2<3>(a0, a1, b, lb, W, out0, out1); break; default: reg_mul_digit2<4>(a0, a1, b, lb, W, out0, out1); } }
int emu_reg_len(int D, const uint32_t *a, int la)
{ switch (D) { case 1: return reg_len<1>(a, la); case 2: return reg_len<2>(a, la); case 3: return reg_len<3>(a, la); default: return reg_len<4>(a, la); } }
}
