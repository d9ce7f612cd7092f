// latency_probe.hip -- what the primitives of the commit chain cost on this GPU, in shader-clock ticks (clock64) and ns
// (wall_clock64, 100 MHz): a workgroup barrier with 8 waves, dependent global loads (plain / sc1 = agent-scope atomic),
// a write-through store + drain, LDS round trips.  hipcc --offload-arch=gfx950 -O3 tools/latency_probe.hip -o latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline uint32_t ld_sc1(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_sc1(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// chain[i] holds the index of the next element (a random cycle over a buffer far larger than L2)
__global__ void probe(uint32_t *chain, uint32_t n, uint32_t *scratch, unsigned long long *out, int reps)
{
    __shared__ uint32_t lds[1024];
    const int tid = threadIdx.x;
    if (blockIdx.x != 0) {            // background blocks: idle spinning like waiting column workers
        unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 200000ull) __builtin_amdgcn_s_sleep(64);
        return;
    }
    unsigned long long c0, c1, w0, w1;
    // (0) barrier
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    for (int r = 0; r < reps; r++) __syncthreads();
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
    // (1) dependent plain loads (thread 0)
    uint32_t idx = tid * 977u % n;
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    if (tid == 0) for (int r = 0; r < reps; r++) idx = chain[idx];
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[2] = c1 - c0; out[3] = w1 - w0; scratch[0] = idx; }
    // (2) dependent sc1 loads (thread 0)
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    if (tid == 0) for (int r = 0; r < reps; r++) idx = ld_sc1(chain + idx);
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[4] = c1 - c0; out[5] = w1 - w0; scratch[1] = idx; }
    // (3) sc1 store + drain (thread 0)
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    if (tid == 0) for (int r = 0; r < reps; r++) { st_sc1(scratch + 64 + 32 * (r & 15), (uint32_t) r); __builtin_amdgcn_s_waitcnt(0); }
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[6] = c1 - c0; out[7] = w1 - w0; }
    // (4) plain store + drain
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    if (tid == 0) for (int r = 0; r < reps; r++) { scratch[1024 + 32 * (r & 15)] = (uint32_t) r; __builtin_amdgcn_s_waitcnt(0); }
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[8] = c1 - c0; out[9] = w1 - w0; }
    // (5) LDS round trip: write, wave barrier, read neighbour (wave 0)
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    uint32_t v = tid;
    if (tid < 64) for (int r = 0; r < reps; r++) { lds[tid] = v; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); v = lds[(tid + 1) & 63] + 1; }
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[10] = c1 - c0; out[11] = w1 - w0; scratch[2] = v; }
    // (6) 64 independent sc1 loads by one wave (one per lane), then a dependent round: what "one round of loads" costs
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    if (tid < 64) for (int r = 0; r < reps; r++) idx = ld_sc1(chain + idx);
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[12] = c1 - c0; out[13] = w1 - w0; }
    if (tid < 64) scratch[8 + tid] = idx;
    // (7) plain ALU chain: 1000 dependent integer adds/muls
    __syncthreads();
    c0 = clock64(); w0 = wall_clock64();
    uint32_t a = idx | 1u;
    for (int r = 0; r < reps * 16; r++) a = a * 2654435761u + 12345u;
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { out[14] = c1 - c0; out[15] = w1 - w0; scratch[3] = a; }
}

int main(int argc, char **argv)
{
    const int nblocks = argc > 1 ? atoi(argv[1]) : 1;
    const int reps = 64;
    const uint32_t n = 64u << 20;                 // 256 MB of indices
    std::vector<uint32_t> h(n);
    // a single cycle with a large stride pattern: i -> (i * odd + c) mod n visits lines pseudo-randomly
    for (uint32_t i = 0; i < n; i++) h[i] = (uint32_t)(((uint64_t) i * 40503u * 64u + 7919u * 64u + i / 1024u) % n);
    uint32_t *d, *scratch; unsigned long long *out;
    CHECK(hipMalloc(&d, (size_t) n * 4)); CHECK(hipMalloc(&scratch, 1 << 20)); CHECK(hipMalloc(&out, 16 * 8));
    CHECK(hipMemcpy(d, h.data(), (size_t) n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(scratch, 0, 1 << 20));
    for (int it = 0; it < 3; it++) {
        hipLaunchKernelGGL(probe, dim3(nblocks), dim3(512), 0, 0, d, n, scratch, out, reps);
        CHECK(hipDeviceSynchronize());
    }
    unsigned long long o[16];
    CHECK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
    const char *names[8] = { "workgroup barrier (8 waves)", "dependent plain load", "dependent sc1 load", "sc1 store + drain", "plain store + drain",
                             "LDS write/wave-barrier/read", "one round of 64 sc1 loads (wave)", "16 dependent mul+add" };
    printf("blocks %d: per operation, shader ticks / ns (wall clock 100 MHz)\n", nblocks);
    for (int q = 0; q < 8; q++) printf("  %-36s %8.0f ticks  %8.0f ns\n", names[q], (double) o[2 * q] / reps, (double) o[2 * q + 1] * 10.0 / reps);
    return 0;
}
