/* wave_shim.h -- the few wave64 / workgroup primitives the REF-LU kernels use.
 *
 * Device build (hipcc, gfx950): thin inline wrappers over the CDNA4 wave64
 * intrinsics (__shfl*, __ballot, __syncthreads, LDS/global atomics).
 *
 * Host test build (-DSLIP_EMULATE, tests/emu): the same kernel source is run
 * lane-by-lane on cooperative fibers so that the limb arithmetic, the carry
 * resolution and the column loop can be unit-tested, and run under
 * ASan/UBSan, in a container without a GPU.  The emulator is test
 * infrastructure only; the product library is always the hipcc build.
 */
#ifndef SLIP_WAVE_SHIM_H
#define SLIP_WAVE_SHIM_H

#include <stdint.h>

#define SLIP_WAVE 64

#ifdef SLIP_EMULATE
/* ------------------------------------------------------------------ */
#include "fiber_emu.h"          /* tests/emu */
#define SLIP_DEV static inline
#define SLIP_DEVN static __attribute__((noinline))
#define SLIP_SHARED static
#define SLIP_KERNEL

static inline int slip_tid(void)      { return emu::tid(); }
static inline int slip_nthreads(void) { return emu::nthreads(); }
static inline int slip_block(void)    { return emu::block(); }
static inline int slip_uniform_i32(int v) { return v; }
static inline int slip_nblocks(void)  { return emu::nblocks(); }
/* macros, so that the divergence check sees the CALL SITE's line */
#define slip_block_sync()      emu::block_sync(__LINE__)
#define slip_block_sync_named(id) emu::block_sync(-(id))     /* one barrier written at several places of the source */
#define slip_wave_sync()       ((void) emu::ballot(0, __LINE__))
#define slip_wave_sync_lds()   ((void) emu::ballot(0, __LINE__))
#define slip_ballot(pred)      emu::ballot((pred), __LINE__)
#define slip_shfl_u32(v, src)  ((uint32_t) emu::shfl((uint64_t)(v), (src), __LINE__))
#define slip_shfl_u64(v, src)  emu::shfl((uint64_t)(v), (src), __LINE__)
/* lane l receives lane l-1's value (lane 0: fill) / lane l+1's (lane 63: fill) / rotation; uniform-lane read */
static inline uint32_t slip_emu_shr1(const uint64_t *o, uint32_t fill) { int l = emu::tid() & 63; return l ? (uint32_t) o[l - 1] : fill; }
static inline uint32_t slip_emu_shl1(const uint64_t *o, uint32_t fill) { int l = emu::tid() & 63; return l < 63 ? (uint32_t) o[l + 1] : fill; }
#define slip_dpp_shr1(v, fill) slip_emu_shr1(emu::collective((uint64_t)(v), __LINE__), (fill))
#define slip_dpp_shl1(v, fill) slip_emu_shl1(emu::collective((uint64_t)(v), __LINE__), (fill))
#define slip_readlane(v, lane) ((uint32_t) emu::shfl((uint64_t)(v), (lane), __LINE__))
#define slip_dpp_shr1_in(v, in) slip_dpp_shr1((v), (in))
/* which XCD (die) this workgroup runs on; 0 in the emulation */
static inline uint32_t slip_xcc_id(void) { return 0u; }
/* a value the caller knows to be wave-uniform, kept in the scalar unit (no-op in the emulation) */
#define slip_uniform(v) ((uint32_t)(v))
/* Hensel carry step on wave-uniform words: (n1:n0) = m1 + c1 + (t < c0) + (m2 << 32) */
static inline void slip_carry_step(uint32_t m1, uint32_t c1, uint32_t t, uint32_t c0, uint32_t m2, uint32_t &n0, uint32_t &n1)
{
    const uint64_t cn = (uint64_t) m1 + c1 + (t < c0 ? 1u : 0u) + ((uint64_t) m2 << 32);
    n0 = (uint32_t) cn; n1 = (uint32_t)(cn >> 32);
}
/* the wave-uniform value s into lane `l` of v */
#define slip_writelane(v, s, l) (((emu::tid() & 63) == (l)) ? (uint32_t)(s) : (uint32_t)(v))
#define slip_dpp_shr1_zero(v) slip_dpp_shr1((v), 0u)
static inline void slip_valu_settle(void) {}
/* value of lane 0 in every lane (all lanes active) */
#define slip_bcast0_u32(v) ((uint32_t) emu::shfl((uint64_t)(v), 0, __LINE__))
/* reductions over the 64 lanes, result in every lane (all lanes active) */
static inline uint32_t slip_emu_red(const uint64_t *o, int op) { uint32_t r = (uint32_t) o[0]; for (int i = 1; i < 64; i++) { const uint32_t v = (uint32_t) o[i]; r = op == 0 ? r + v : (op == 1 ? (v > r ? v : r) : (v < r ? v : r)); } return r; }
#define slip_wave_sum_u32(v) slip_emu_red(emu::collective((uint64_t)(uint32_t)(v), __LINE__), 0)
#define slip_wave_max_u32(v) slip_emu_red(emu::collective((uint64_t)(uint32_t)(v), __LINE__), 1)
#define slip_wave_min_u32(v) slip_emu_red(emu::collective((uint64_t)(uint32_t)(v), __LINE__), 2)
/* (hi:acc) += a * b, a 96-bit per-lane accumulator */
static inline void slip_mac96(uint64_t &acc, uint32_t &hi, uint32_t a, uint32_t b)
{
    const uint64_t p = (uint64_t) a * b, sum = acc + p;
    hi += (uint32_t)(sum < p);
    acc = sum;
}
static inline uint32_t slip_atomic_or_u32(uint32_t *p, uint32_t v) { emu::sync_addr(p); uint32_t o = *p; *p = o | v; return o; }
static inline int32_t  slip_atomic_max_i32(int32_t *p, int32_t v) { emu::sync_addr(p); int32_t o = *p; if (v > o) *p = v; return o; }
static inline int32_t  slip_atomic_min_i32(int32_t *p, int32_t v) { emu::sync_addr(p); int32_t o = *p; if (v < o) *p = v; return o; }
static inline int32_t  slip_atomic_add_i32(int32_t *p, int32_t v) { emu::sync_addr(p); int32_t o = *p; *p = o + v; return o; }
static inline uint32_t slip_atomic_cas_u32(uint32_t *p, uint32_t expect, uint32_t v) { emu::sync_addr(p); uint32_t o = *p; if (o == expect) *p = v; return o; }
static inline unsigned long long slip_atomic_add_u64(unsigned long long *p, unsigned long long v) { emu::sync_addr(p); unsigned long long o = *p; *p = o + v; return o; }
static inline void slip_fence_block(void) {}
static inline void slip_fence_device(void) {}
/* agent-scope hand-off primitives (what they order only matters in the emulator's weak-store mode, fiber_emu.h: a drain /
 * release empties the calling wave's store buffer; the hand-off store and the read-modify-write words do so first) */
static inline void slip_vm_drain(void) { emu::drain(); }
static inline void slip_agent_release(void) { emu::drain(); }
static inline void slip_agent_acquire(void) {}
static inline int32_t slip_agent_load_i32(const int32_t *p) { return (int32_t) emu::load(p, 4); }
static inline void slip_agent_store_i32(int32_t *p, int32_t v) { emu::drain(); *(volatile int32_t *) p = v; }
static inline int32_t slip_agent_add_i32(int32_t *p, int32_t v) { emu::sync_addr(p); int32_t o = *p; *p = o + v; return o; }
static inline void slip_sleep(void) { emu::spin_yield(); }       /* a spin-wait iteration: let the other workgroups run */
static inline void slip_sleep_short(void) { emu::spin_yield(); }
static inline unsigned long long slip_clock(void) { return 0; }
static inline unsigned long long slip_realtime(void) { return 0; }
/* data other workgroups write / read during a launch (sc1 on the device): through the emulator's store buffers */
static inline uint32_t slip_ld_u32(const uint32_t *p) { return (uint32_t) emu::load(p, 4); }
static inline int32_t  slip_ld_i32(const int32_t *p)  { return (int32_t) emu::load(p, 4); }
static inline uint64_t slip_ld_u64(const uint64_t *p) { return emu::load(p, 8); }
static inline int64_t  slip_ld_i64(const int64_t *p)  { return (int64_t) emu::load(p, 8); }
static inline void slip_st_u32(uint32_t *p, uint32_t v) { emu::store(p, v, 4); }
static inline void slip_st_i32(int32_t *p, int32_t v)   { emu::store(p, (uint32_t) v, 4); }
static inline void slip_st_u64(uint64_t *p, uint64_t v) { emu::store(p, v, 8); }
static inline void slip_st_i64(int64_t *p, int64_t v)   { emu::store(p, (uint64_t) v, 8); }
/* this workgroup's own global data (plain global accesses on the device) */
static inline uint32_t slip_gld_u32(const uint32_t *p) { return *p; }
static inline void slip_gst_u32(uint32_t *p, uint32_t v) { *p = v; }
static inline int32_t slip_agent_max_i32(int32_t *p, int32_t v) { emu::sync_addr(p); int32_t o = *p; if (v > o) *p = v; return o; }
static inline int32_t slip_agent_cas_i32(int32_t *p, int32_t expect, int32_t v) { emu::sync_addr(p); int32_t o = *p; if (o == expect) *p = v; return o; }
static inline int64_t slip_agent_min_i64(int64_t *p, int64_t v) { emu::sync_addr(p); int64_t o = *p; if (v < o) *p = v; return o; }
static inline unsigned long long slip_agent_add_u64(unsigned long long *p, unsigned long long v) { emu::sync_addr(p); unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned long long slip_agent_max_u64(unsigned long long *p, unsigned long long v) { emu::sync_addr(p); unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline int slip_clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
static inline int slip_ctz32(uint32_t v) { return v ? __builtin_ctz(v) : 32; }
static inline int slip_clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }
static inline int slip_ctz64(uint64_t v) { return v ? __builtin_ctzll(v) : 64; }
static inline int slip_popc32(uint32_t v) { return __builtin_popcount(v); }
static inline int slip_popc64(uint64_t v) { return __builtin_popcountll(v); }

#else
/* ------------------------------------------------------------------ */
#include <hip/hip_runtime.h>
#define SLIP_DEV __device__ __forceinline__
/* out-of-line device function: ONE copy of a heavy wave-level routine instead of one per call site (the column loop
 * has to fit the 64 KB instruction cache) */
#define SLIP_DEVN __device__ __attribute__((noinline))
#define SLIP_SHARED __shared__
#define SLIP_KERNEL __global__

SLIP_DEV int slip_tid(void)      { return (int) threadIdx.x; }
SLIP_DEV int slip_nthreads(void) { return (int) blockDim.x; }
SLIP_DEV int slip_block(void)    { return (int) blockIdx.x; }
SLIP_DEV int slip_uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }   /* v is the same in every lane */
SLIP_DEV int slip_nblocks(void)  { return (int) gridDim.x; }
SLIP_DEV void slip_block_sync(void) { __syncthreads(); }
/* one barrier that different waves reach at different places of the source (the emulator checks that all threads of a
 * workgroup meet at the same barrier: these are matched by name instead of by line) */
SLIP_DEV void slip_block_sync_named(int) { __syncthreads(); }
/* orders this wave's LDS/global writes before its lanes' later cross-lane reads */
SLIP_DEV void slip_wave_sync(void)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
/* the same for LDS only: a wave's LDS operations execute in order, so only the compiler has to be told (no wait for
 * outstanding global stores, which the workgroup-scope fences above imply) */
SLIP_DEV void slip_wave_sync_lds(void)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
SLIP_DEV uint64_t slip_ballot(int pred) { return (uint64_t) __ballot(pred); }
SLIP_DEV uint32_t slip_shfl_u32(uint32_t v, int src) { return (uint32_t) __shfl((int) v, src, SLIP_WAVE); }
SLIP_DEV uint64_t slip_shfl_u64(uint64_t v, int src)
{
    uint32_t lo = (uint32_t) __shfl((int)(uint32_t) v, src, SLIP_WAVE);
    uint32_t hi = (uint32_t) __shfl((int)(uint32_t)(v >> 32), src, SLIP_WAVE);
    return ((uint64_t) hi << 32) | lo;
}
/* DPP wave shifts (GFX9 wave_shr:1 / wave_shl:1): one VALU op, no LDS crossbar */
SLIP_DEV uint32_t slip_dpp_shr1(uint32_t v, uint32_t fill)
{
    return (uint32_t) __builtin_amdgcn_update_dpp((int) fill, (int) v, 0x138, 0xF, 0xF, false);
}
SLIP_DEV uint32_t slip_dpp_shl1(uint32_t v, uint32_t fill)
{
    return (uint32_t) __builtin_amdgcn_update_dpp((int) fill, (int) v, 0x130, 0xF, 0xF, false);
}
SLIP_DEV uint32_t slip_readlane(uint32_t v, int lane) { return (uint32_t) __builtin_amdgcn_readlane((int) v, lane); }
/* which XCD (die) this workgroup runs on (XCC_ID register) */
SLIP_DEV uint32_t slip_xcc_id(void) { return (uint32_t) __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu; }
/* a value the caller knows to be wave-uniform, kept in the scalar unit */
SLIP_DEV uint32_t slip_uniform(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
/* Hensel carry step on wave-uniform words, in the scalar unit: (n1:n0) = m1 + c1 + (t < c0) + (m2 << 32).  Written out:
 * the compiler selects the vector unit's add-with-carry for this (and then moves the whole quotient chain there) */
SLIP_DEV void slip_carry_step(uint32_t m1, uint32_t c1, uint32_t t, uint32_t c0, uint32_t m2, uint32_t &n0, uint32_t &n1)
{
    /* (readfirstlane: an "s" operand the register allocator had put into a VGPR is emitted as such) */
    m1 = slip_uniform(m1); c1 = slip_uniform(c1); t = slip_uniform(t); c0 = slip_uniform(c0); m2 = slip_uniform(m2);
    asm("s_cmp_lt_u32 %4, %5\n\ts_addc_u32 %0, %2, %3\n\ts_addc_u32 %1, %6, 0"
        : "=&s"(n0), "=s"(n1) : "s"(m1), "s"(c1), "s"(t), "s"(c0), "s"(m2) : "scc");
}
/* the wave-uniform value s into lane `l` of v (v_writelane_b32) */
extern "C" __device__ int slip_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");     /* (clang has no builtin for it) */
SLIP_DEV uint32_t slip_writelane(uint32_t v, uint32_t s, int l) { return (uint32_t) slip_llvm_writelane((int) s, l, (int) v); }
/* wave shift by one lane with a wave-uniform value entering lane 0: zero-filling DPP move (no `old` operand to
 * set up) + v_writelane */
SLIP_DEV uint32_t slip_dpp_shr1_in(uint32_t v, uint32_t in)
{
    uint32_t sh = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x138, 0xF, 0xF, true);
    asm("v_writelane_b32 %0, %1, 0" : "+v"(sh) : "s"(in));
    return sh;
}
/* value of lane 0 in every lane (all lanes active): one v_readfirstlane instead of a trip through the LDS crossbar */
SLIP_DEV uint32_t slip_bcast0_u32(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
/* reductions over the 64 lanes with DPP row shifts and row broadcasts (VALU only; ds_bpermute shuffles cost an LDS round
 * trip per step); the total lands in lane 63 and is handed to every lane through an SGPR.  All lanes active. */
#define SLIP_DPP_STEP(op, ctrl, rowmask, ident) { const uint32_t o_ = (uint32_t) __builtin_amdgcn_update_dpp((int)(ident), (int) v, ctrl, rowmask, 0xF, false); v = op(v, o_); }
SLIP_DEV uint32_t slip_red_add_(uint32_t a, uint32_t b) { return a + b; }
SLIP_DEV uint32_t slip_red_max_(uint32_t a, uint32_t b) { return a > b ? a : b; }
SLIP_DEV uint32_t slip_red_min_(uint32_t a, uint32_t b) { return a < b ? a : b; }
SLIP_DEV uint32_t slip_wave_sum_u32(uint32_t v)
{
    SLIP_DPP_STEP(slip_red_add_, 0x111, 0xF, 0u) SLIP_DPP_STEP(slip_red_add_, 0x112, 0xF, 0u) SLIP_DPP_STEP(slip_red_add_, 0x114, 0xF, 0u)
    SLIP_DPP_STEP(slip_red_add_, 0x118, 0xF, 0u) SLIP_DPP_STEP(slip_red_add_, 0x142, 0xA, 0u) SLIP_DPP_STEP(slip_red_add_, 0x143, 0xC, 0u)
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}
SLIP_DEV uint32_t slip_wave_max_u32(uint32_t v)
{
    SLIP_DPP_STEP(slip_red_max_, 0x111, 0xF, 0u) SLIP_DPP_STEP(slip_red_max_, 0x112, 0xF, 0u) SLIP_DPP_STEP(slip_red_max_, 0x114, 0xF, 0u)
    SLIP_DPP_STEP(slip_red_max_, 0x118, 0xF, 0u) SLIP_DPP_STEP(slip_red_max_, 0x142, 0xA, 0u) SLIP_DPP_STEP(slip_red_max_, 0x143, 0xC, 0u)
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}
SLIP_DEV uint32_t slip_wave_min_u32(uint32_t v)
{
    SLIP_DPP_STEP(slip_red_min_, 0x111, 0xF, 0xFFFFFFFFu) SLIP_DPP_STEP(slip_red_min_, 0x112, 0xF, 0xFFFFFFFFu) SLIP_DPP_STEP(slip_red_min_, 0x114, 0xF, 0xFFFFFFFFu)
    SLIP_DPP_STEP(slip_red_min_, 0x118, 0xF, 0xFFFFFFFFu) SLIP_DPP_STEP(slip_red_min_, 0x142, 0xA, 0xFFFFFFFFu) SLIP_DPP_STEP(slip_red_min_, 0x143, 0xC, 0xFFFFFFFFu)
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}
/* results of the inline-asm VALU ops above may be read by a DPP op next: give the pipeline its two wait states */
SLIP_DEV void slip_valu_settle(void) { asm volatile("s_nop 1"); }
SLIP_DEV uint32_t slip_dpp_shr1_zero(uint32_t v) { return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x138, 0xF, 0xF, true); }
/* (hi:acc) += a * b with a wave-uniform (SGPR): v_mad_u64_u32 accumulates in place and hands its carry to one
 * v_addc -- the compiler's own expansion of the same C is mad + 64-bit add + 64-bit compare + addc */
SLIP_DEV void slip_mac96(uint64_t &acc, uint32_t &hi, uint32_t a, uint32_t b)
{
    uint64_t cy;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(acc), "+v"(hi), "=&s"(cy) : "s"(a), "v"(b));
}
SLIP_DEV uint32_t slip_atomic_or_u32(uint32_t *p, uint32_t v) { return atomicOr(p, v); }
SLIP_DEV int32_t  slip_atomic_max_i32(int32_t *p, int32_t v) { return atomicMax(p, v); }
SLIP_DEV int32_t  slip_atomic_min_i32(int32_t *p, int32_t v) { return atomicMin(p, v); }
SLIP_DEV int32_t  slip_atomic_add_i32(int32_t *p, int32_t v) { return atomicAdd(p, v); }
SLIP_DEV uint32_t slip_atomic_cas_u32(uint32_t *p, uint32_t expect, uint32_t v) { return atomicCAS(p, expect, v); }      /* returns the value found */
SLIP_DEV unsigned long long slip_atomic_add_u64(unsigned long long *p, unsigned long long v) { return atomicAdd(p, v); }
SLIP_DEV void slip_fence_block(void) { __threadfence_block(); }
SLIP_DEV void slip_fence_device(void) { __threadfence(); }
/* agent-scope hand-off between workgroups on different CUs / XCDs (cdna_hip_programming.md, Guideline 16):
 * producer: stores -> every storing wave drains (vmcnt(0)) -> workgroup barrier -> ONE lane release
 *           -> drain -> relaxed agent-scope flag store / counter add;
 * consumer: ONE lane polls relaxed -> agent acquire -> drain -> workgroup barrier -> plain loads. */
SLIP_DEV void slip_vm_drain(void) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
SLIP_DEV void slip_agent_release(void)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
SLIP_DEV void slip_agent_acquire(void)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
/* polled words are read with a returning agent-scope RMW (add 0): it is performed at the memory side, so it
 * cannot be served from a stale copy of the line in this XCD's L2 (observed: rare polls that never saw an update) */
SLIP_DEV int32_t slip_agent_load_i32(const int32_t *p) { return __hip_atomic_fetch_add((int32_t *) p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV void slip_agent_store_i32(int32_t *p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV int32_t slip_agent_add_i32(int32_t *p, int32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV void slip_sleep(void) { __builtin_amdgcn_s_sleep(32); }
SLIP_DEV void slip_sleep_short(void) { __builtin_amdgcn_s_sleep(2); }
/* Data other workgroups write / read during a launch: relaxed agent-scope accesses through GLOBAL pointers
 * (global_load/store ... sc1: the load bypasses this CU's L1, the store is written through; never flat_,
 * cdna_hip_programming.md Guideline 16).  The writer drains (vmcnt(0)) before it raises the flag that
 * publishes the data; the reader needs no acquire fence because EVERY load of such data is one of these. */
typedef __attribute__((address_space(1))) uint32_t slip_gu32;
typedef __attribute__((address_space(1))) int32_t  slip_gi32;
typedef __attribute__((address_space(1))) uint64_t slip_gu64;
typedef __attribute__((address_space(1))) int64_t  slip_gi64;
SLIP_DEV uint32_t slip_ld_u32(const uint32_t *p) { return __hip_atomic_load((slip_gu32 *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV int32_t  slip_ld_i32(const int32_t *p)  { return __hip_atomic_load((slip_gi32 *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV uint64_t slip_ld_u64(const uint64_t *p) { return __hip_atomic_load((slip_gu64 *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV int64_t  slip_ld_i64(const int64_t *p)  { return __hip_atomic_load((slip_gi64 *) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV void slip_st_u32(uint32_t *p, uint32_t v) { __hip_atomic_store((slip_gu32 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV void slip_st_i32(int32_t *p, int32_t v)   { __hip_atomic_store((slip_gi32 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV void slip_st_u64(uint64_t *p, uint64_t v) { __hip_atomic_store((slip_gu64 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV void slip_st_i64(int64_t *p, int64_t v)   { __hip_atomic_store((slip_gi64 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
/* this workgroup's own global data: plain accesses, but through global (not flat) instructions */
SLIP_DEV uint32_t slip_gld_u32(const uint32_t *p) { return *(const slip_gu32 *) p; }
SLIP_DEV void slip_gst_u32(uint32_t *p, uint32_t v) { *(slip_gu32 *) p = v; }
SLIP_DEV int32_t slip_agent_max_i32(int32_t *p, int32_t v) { return __hip_atomic_fetch_max((slip_gi32 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV int32_t slip_agent_cas_i32(int32_t *p, int32_t expect, int32_t v)      /* returns the value found */
{
    __hip_atomic_compare_exchange_strong((slip_gi32 *) p, &expect, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return expect;
}
SLIP_DEV int64_t slip_agent_min_i64(int64_t *p, int64_t v) { return __hip_atomic_fetch_min((slip_gi64 *) p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV unsigned long long slip_agent_add_u64(unsigned long long *p, unsigned long long v) { return __hip_atomic_fetch_add((slip_gu64 *) p, (uint64_t) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV unsigned long long slip_agent_max_u64(unsigned long long *p, unsigned long long v) { return __hip_atomic_fetch_max((slip_gu64 *) p, (uint64_t) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
SLIP_DEV unsigned long long slip_clock(void) { return (unsigned long long) clock64(); }
SLIP_DEV unsigned long long slip_realtime(void) { return (unsigned long long) __builtin_amdgcn_s_memrealtime(); }   /* 100 MHz, chip-wide */
SLIP_DEV int slip_clz32(uint32_t v) { return __clz((int) v); }
SLIP_DEV int slip_ctz32(uint32_t v) { return v ? __ffs((int) v) - 1 : 32; }
SLIP_DEV int slip_clz64(uint64_t v) { return __clzll((long long) v); }
SLIP_DEV int slip_ctz64(uint64_t v) { return v ? __ffsll((unsigned long long) v) - 1 : 64; }
SLIP_DEV int slip_popc32(uint32_t v) { return __popc(v); }
SLIP_DEV int slip_popc64(uint64_t v) { return __popcll(v); }
#endif

SLIP_DEV int slip_lane(void)   { return slip_tid() & (SLIP_WAVE - 1); }
/* wave-uniform by construction: handing it to the compiler as a scalar turns the wave-strided loops and their
 * branches into SALU code instead of exec-masked VALU code */
SLIP_DEV int slip_wave(void)   { return slip_uniform_i32(slip_tid() >> 6); }
SLIP_DEV int slip_nwaves(void) { return slip_nthreads() >> 6; }

/* value of lane (lane-d), own value for lanes < d */
SLIP_DEV uint32_t slip_shfl_up_u32(uint32_t v, int d)
{
    int l = slip_lane(), s = l - d;
    uint32_t r = slip_shfl_u32(v, s < 0 ? l : s);
    return r;
}

#endif /* SLIP_WAVE_SHIM_H */
