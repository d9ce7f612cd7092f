/* slabio.h -- tiny named-array container used for golden fixtures.
 * TEST INFRASTRUCTURE ONLY (oracle/): the product never includes this.
 *
 * File layout:  8-byte magic "SLAB0001", then records
 *    char     name[24]   (zero padded)
 *    int32_t  dtype      (0 = int32, 1 = int64, 2 = uint64, 3 = float64)
 *    int32_t  reserved
 *    int64_t  count
 *    count * sizeof(dtype) bytes, zero padded to a multiple of 8
 * Read back by tests/slabfile.py.
 */
#ifndef SLABIO_H
#define SLABIO_H
#include <stdio.h>
#include <stdint.h>
#include <string.h>

enum { SLAB_I32 = 0, SLAB_I64 = 1, SLAB_U64 = 2, SLAB_F64 = 3 };

static inline FILE *slab_open(const char *path)
{
    FILE *f = fopen(path, "wb");
    if (f) fwrite("SLAB0001", 1, 8, f);
    return f;
}

static inline void slab_put(FILE *f, const char *name, int dtype, const void *data, int64_t count)
{
    char nm[24]; memset(nm, 0, sizeof nm); strncpy(nm, name, 23);
    int32_t hdr[2] = { dtype, 0 };
    size_t esz = (dtype == SLAB_I32) ? 4 : 8;
    fwrite(nm, 1, 24, f); fwrite(hdr, 4, 2, f); fwrite(&count, 8, 1, f);
    if (count > 0) fwrite(data, esz, (size_t)count, f);
    size_t bytes = esz * (size_t)count, pad = (8 - bytes % 8) % 8;
    static const char z[8] = {0};
    if (pad) fwrite(z, 1, pad, f);
}
#endif
