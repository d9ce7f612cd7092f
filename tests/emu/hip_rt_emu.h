/* hip_rt_emu.h -- the handful of HIP runtime calls the host side of
 * slip_hip.hip uses, mapped onto malloc/memcpy for the CPU emulation build
 * (TEST ONLY; see fiber_emu.h). */
#ifndef HIP_RT_EMU_H
#define HIP_RT_EMU_H
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int hipError_t;
#define hipSuccess 0
typedef void *hipStream_t;
typedef struct { double t; } *hipEvent_t;
enum { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipFree(void *p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { if (n) memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { if (n) memcpy(d, s, n); return 0; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return 0; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize(void) { return 0; }
static inline hipError_t hipGetLastError(void) { return 0; }
static inline const char *hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipGetDeviceCount(int *c) { *c = 1; return 0; }
static inline double emu_now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t) malloc(sizeof(**e)); return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu_now_ms(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return 0; }
#endif
