/* fiber_emu.h -- lane-by-lane CPU emulation of a grid of wave64 workgroups (TEST ONLY).
 *
 * Every GPU thread becomes a ucontext fiber.  All workgroups of a launch are alive at the same
 * time and their fibers are switched cooperatively at the cross-lane operations (shfl / ballot /
 * block barrier) and at the spin-wait points of the inter-workgroup protocols (emu::spin_yield),
 * so a persistent multi-workgroup kernel (column workers that wait for each other's commits) runs
 * to completion on one OS thread.  The order in which workgroups get the processor is a seeded
 * pseudo-random interleaving (emu::set_seed), which lets the tests shake the hand-off logic.
 * All 64 lanes of a wave must reach the same call site (checked): exactly the wave-uniform
 * control flow the real kernel needs.  Memory is sequentially consistent by default: cache-visibility
 * bugs (missing sc1 / release) are NOT found by this emulator, logic and ordering bugs are.
 *
 * WEAK-STORE MODE (emu::set_weak(1)): the stores other workgroups are meant to see (the slip_st_* /
 * sc1 accessors of wave_shim.h) do not reach memory when they are issued.  Each WAVE keeps them in a buffer; an entry
 * lands at a pseudo-random later moment, in any order except that stores of one wave to one address keep theirs (that is
 * all the hardware promises for write-through stores in flight).  The issuing wave sees its own pending stores (same-address
 * ordering within a wave), nobody else does -- not even the other waves of its workgroup.  A wave's buffer is emptied by what
 * empties it on the device: slip_vm_drain (s_waitcnt vmcnt(0)), slip_agent_release, a workgroup barrier (__syncthreads
 * waits for the wave's memory operations first), and the agent-scope store/atomics used as hand-off words.  Plain stores
 * (bulk data written before a release fence) are NOT modelled: they stay immediately visible, which errs on the quiet
 * side.  What the mode finds: a hand-off word that can overtake the data it announces (a missing drain), and two waves
 * writing one word with the order left to chance -- the two protocol races of rounds 2 and 3.
 * Nothing here is part of the product.
 */
#ifndef FIBER_EMU_H
#define FIBER_EMU_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>
#include <functional>
#include <vector>

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
#define EMU_ASAN 1
#endif

namespace emu {

struct Pending { void *addr; uint64_t val; int size; };
struct WaveCtx {
    uint64_t vals[64], out[64];
    int site[64];
    int arrived = 0;
    unsigned gen = 0;
    std::vector<Pending> pending;        /* weak-store mode: this wave's stores that have not landed yet, in program order */
};

struct Block {
    std::vector<ucontext_t> ctx;
    std::vector<char> done;
    std::vector<WaveCtx> waves;
    std::vector<int> last_site;          /* per thread: line of the cross-lane op it last entered (-line: barrier) */
    int bar_arrived = 0; unsigned bar_gen = 0; int bar_site = 0;
    int remaining = 0;
};

struct State {
    int nthreads = 0, nblocks = 0, block = 0, cur = 0;
    std::vector<Block> blocks;
    std::vector<char *> stacks;          /* nblocks * nthreads stacks, reused across launches */
    size_t stack_bytes = 0;
    ucontext_t sched;
    const void *sched_stack = nullptr; size_t sched_stack_size = 0;
    unsigned long progress = 0;          /* bumped whenever a collective / barrier completes or a thread ends */
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    int weak = 0;                        /* weak-store mode */
    unsigned long pending_total = 0;
    std::function<void()> body;
};

inline State &S() { static State s; return s; }

inline void set_seed(unsigned long long seed) { S().rng = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull; }
inline unsigned rnd() { State &s = S(); s.rng ^= s.rng << 13; s.rng ^= s.rng >> 7; s.rng ^= s.rng << 17; return (unsigned)(s.rng >> 32); }

inline void set_weak(int on) { S().weak = on; }
inline void poke(void *addr, uint64_t v, int size)
{
    if (size == 4) *(volatile uint32_t *) addr = (uint32_t) v; else *(volatile uint64_t *) addr = v;
}
/* a store other workgroups are meant to see */
inline void store(void *addr, uint64_t v, int size)
{
    State &s = S();
    if (!s.weak) { poke(addr, v, size); return; }
    s.blocks[s.block].waves[s.cur >> 6].pending.push_back(Pending{addr, v, size});
    s.pending_total++;
}
/* a load of such data: memory, or this wave's own newest pending store to the address */
inline uint64_t load(const void *addr, int size)
{
    State &s = S();
    if (s.weak) {
        const std::vector<Pending> &p = s.blocks[s.block].waves[s.cur >> 6].pending;
        for (size_t i = p.size(); i-- > 0;) {
            if (p[i].addr == addr && p[i].size == size) return p[i].val;
            /* a 4-byte read of half of a pending 8-byte store (and the reverse) is not forwarded: the code never does that
             * with data in flight, and memory order of mixed sizes is the hardware's business */
        }
    }
    return size == 4 ? (uint64_t) *(volatile const uint32_t *) addr : *(volatile const uint64_t *) addr;
}
/* everything this wave has issued reaches memory, in program order (s_waitcnt vmcnt(0) / a release / a barrier) */
inline void drain()
{
    State &s = S();
    if (!s.weak) return;
    std::vector<Pending> &p = s.blocks[s.block].waves[s.cur >> 6].pending;
    for (const Pending &e : p) poke(e.addr, e.val, e.size);
    s.pending_total -= p.size();
    p.clear();
}
/* a read-modify-write of a word: this wave's own pending stores to it land first (same-address order within a wave) */
inline void sync_addr(const void *addr)
{
    State &s = S();
    if (!s.weak) return;
    std::vector<Pending> &p = s.blocks[s.block].waves[s.cur >> 6].pending;
    for (size_t i = 0; i < p.size();) {
        if (p[i].addr == addr) { poke(p[i].addr, p[i].val, p[i].size); p.erase(p.begin() + (long) i); s.pending_total--; }
        else i++;
    }
}
/* a few pending stores of the grid land, in any order that keeps one wave's stores to one address in theirs */
inline void land_some()
{
    State &s = S();
    if (!s.weak || !s.pending_total) return;
    const int tries = 1 + (int)(rnd() % 6u);
    for (int t = 0; t < tries && s.pending_total; t++) {
        Block &b = s.blocks[rnd() % (unsigned) s.nblocks];
        if (b.waves.empty()) continue;
        std::vector<Pending> &p = b.waves[rnd() % (unsigned) b.waves.size()].pending;
        if (p.empty()) continue;
        size_t pick = rnd() % (unsigned) p.size();
        for (size_t i = 0; i < pick; i++) if (p[i].addr == p[pick].addr) { pick = i; break; }     /* the oldest store to that word first */
        poke(p[pick].addr, p[pick].val, p[pick].size);
        p.erase(p.begin() + (long) pick);
        s.pending_total--;
    }
}
inline void land_all()
{
    State &s = S();
    for (Block &b : s.blocks) for (WaveCtx &w : b.waves) { for (const Pending &e : w.pending) poke(e.addr, e.val, e.size); w.pending.clear(); }
    s.pending_total = 0;
}

inline int tid() { return S().cur; }
inline int nthreads() { return S().nthreads; }
inline int block() { return S().block; }
inline int nblocks() { return S().nblocks; }

inline void yield_()
{
    State &s = S();
    Block &b = s.blocks[s.block];
    const int me = s.cur, myb = s.block;
#ifdef EMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, s.sched_stack, s.sched_stack_size);
#endif
    swapcontext(&b.ctx[me], &s.sched);
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
    (void) me; (void) myb;
}

/* a spin-wait iteration of an inter-workgroup protocol: let every other fiber run */
inline void spin_yield() { yield_(); }

[[noreturn]] inline void die(const char *msg, int a, int b)
{
    fprintf(stderr, "fiber_emu: %s (%d vs %d) at block %d thread %d\n", msg, a, b, S().block, S().cur);
    abort();
}

/* generic wave collective: every lane deposits v, gets the whole vector */
inline const uint64_t *collective(uint64_t v, int site)
{
    State &s = S();
    Block &b = s.blocks[s.block];
    int lane = s.cur & 63;
    WaveCtx &w = b.waves[s.cur >> 6];
    unsigned g = w.gen;
    b.last_site[s.cur] = site;
    w.vals[lane] = v; w.site[lane] = site;
    if (++w.arrived == 64) {
        s.progress++;
        for (int i = 0; i < 64; i++) {
            if (w.site[i] != site) die("wave divergence at cross-lane op, lines", w.site[i], site);
            w.out[i] = w.vals[i];
        }
        w.arrived = 0; w.gen++;
    } else {
        while (w.gen == g) yield_();
    }
    return w.out;
}

inline uint64_t ballot(int pred, int site)
{
    const uint64_t *o = collective(pred ? 1 : 0, site);
    uint64_t m = 0;
    for (int i = 0; i < 64; i++) m |= (o[i] & 1) << i;
    return m;
}

inline uint64_t shfl(uint64_t v, int src, int site)
{
    const uint64_t *o = collective(v, site);
    return o[src & 63];
}

inline void block_sync(int site)
{
    State &s = S();
    Block &b = s.blocks[s.block];
    unsigned g = b.bar_gen;
    drain();                             /* __syncthreads waits for the wave's outstanding memory operations first */
    b.last_site[s.cur] = -site;
    if (b.bar_arrived == 0) b.bar_site = site;
    else if (b.bar_site != site) die("threads at different barriers, lines", b.bar_site, site);
    if (++b.bar_arrived == s.nthreads) { b.bar_arrived = 0; b.bar_gen++; s.progress++; }
    else while (b.bar_gen == g) yield_();
}

inline void trampoline()
{
    State &s = S();
#ifdef EMU_ASAN
    __sanitizer_finish_switch_fiber(nullptr, &s.sched_stack, &s.sched_stack_size);
#endif
    s.body();
    s.blocks[s.block].done[s.cur] = 1;
#ifdef EMU_ASAN
    void *fake = nullptr;
    __sanitizer_start_switch_fiber(&fake, s.sched_stack, s.sched_stack_size);   /* this fiber never comes back */
#endif
    swapcontext(&s.blocks[s.block].ctx[s.cur], &s.sched);
}

inline void report_deadlock()
{
    State &s = S();
    fprintf(stderr, "fiber_emu: DEADLOCK -- no collective completed for many scheduler passes; threads wait at "
                    "(line, negative = block barrier, 0 = spin):\n");
    for (int b = 0; b < s.nblocks; b++) {
        Block &B = s.blocks[b];
        for (int t = 0; t < s.nthreads; t++)
            if (!B.done[t] && (t % 64 == 0 || B.last_site[t] != B.last_site[t - 1]))
                fprintf(stderr, "  block %d thread %d.. : %d\n", b, t, B.last_site[t]);
    }
    abort();
}

/* run `body` as a grid of nblocks x nthreads (nthreads a multiple of 64); all blocks are alive together.
 * sequential != 0: block after block (a grid of independent workgroups; cheaper, deterministic). */
inline void launch(int nblocks, int nthreads_, std::function<void()> body, size_t stack_bytes = 256 * 1024, int sequential = 0)
{
    State &s = S();
    if (nthreads_ % 64) die("block size must be a multiple of 64", nthreads_, 64);
    s.nthreads = nthreads_; s.nblocks = nblocks; s.body = body;
    if (s.stack_bytes != stack_bytes) { for (char *p : s.stacks) free(p); s.stacks.clear(); s.stack_bytes = stack_bytes; }
    const size_t live_blocks = sequential ? 1 : (size_t) nblocks;
    const size_t need = live_blocks * (size_t) nthreads_;
    while (s.stacks.size() < need) s.stacks.push_back((char *) malloc(stack_bytes));
    s.blocks.assign(nblocks, Block());
    auto arm = [&](int b) {
        Block &B = s.blocks[b];
        B.ctx.resize(nthreads_); B.done.assign(nthreads_, 0); B.last_site.assign(nthreads_, 0);
        B.waves.assign(nthreads_ / 64, WaveCtx());
        B.remaining = nthreads_;
        for (int t = 0; t < nthreads_; t++) {
            getcontext(&B.ctx[t]);
            B.ctx[t].uc_stack.ss_sp = s.stacks[(sequential ? 0 : (size_t) b * nthreads_) + t];
            B.ctx[t].uc_stack.ss_size = stack_bytes;
            B.ctx[t].uc_link = &s.sched;
            makecontext(&B.ctx[t], (void (*)()) trampoline, 0);
        }
    };
    auto run_pass = [&](int b) {                /* one pass over the fibers of block b */
        Block &B = s.blocks[b];
        for (int t = 0; t < nthreads_ && B.remaining > 0; t++) {
            if (B.done[t]) continue;
            s.block = b; s.cur = t;
#ifdef EMU_ASAN
            void *fake = nullptr;
            __sanitizer_start_switch_fiber(&fake, B.ctx[t].uc_stack.ss_sp, B.ctx[t].uc_stack.ss_size);
#endif
            swapcontext(&s.sched, &B.ctx[t]);
#ifdef EMU_ASAN
            __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
            if (B.done[t]) { B.remaining--; s.progress++; }
        }
    };
    if (sequential) {
        for (int b = 0; b < nblocks; b++) {
            arm(b);
            unsigned long seen = s.progress; int idle = 0;
            while (s.blocks[b].remaining > 0) {
                if (s.progress != seen) { seen = s.progress; idle = 0; }
                else if (++idle > 64) report_deadlock();
                run_pass(b); land_some();
            }
        }
        land_all();
        return;
    }
    for (int b = 0; b < nblocks; b++) arm(b);
    int live = nblocks;
    unsigned long seen = s.progress; long idle = 0;
    while (live > 0) {
        if (s.progress != seen) { seen = s.progress; idle = 0; }
        else if (++idle > 200000) report_deadlock();
        /* pick a live block pseudo-randomly and give it a random number of passes */
        int b = (int)(rnd() % (unsigned) nblocks);
        while (s.blocks[b].remaining == 0) b = (b + 1) % nblocks;
        const int passes = 1 + (int)(rnd() % 4u);
        for (int p = 0; p < passes && s.blocks[b].remaining > 0; p++) { run_pass(b); land_some(); }
        if (s.blocks[b].remaining == 0) live--;
    }
    land_all();
}

} /* namespace emu */
#endif
