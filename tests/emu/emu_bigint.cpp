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
