/* fiber_emu.h -- lane-by-lane CPU emulation of a wave64 workgroup (TEST ONLY).
 *
 * Every GPU thread becomes a ucontext fiber; one workgroup runs at a time and
 * its fibers are switched cooperatively at the cross-lane operations
 * (shfl / ballot / block barrier).  All 64 lanes of a wave must reach the
 * same call site (checked), which is exactly the wave-uniform control flow
 * the real kernel needs.  Nothing here is part of the product.
 */
#ifndef FIBER_EMU_H
#define FIBER_EMU_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <ucontext.h>
#include <functional>
#include <vector>

namespace emu {

struct WaveCtx {
    uint64_t vals[64], out[64];
    int site[64];
    int arrived = 0;
    unsigned gen = 0;
};

struct State {
    int nthreads = 0, nblocks = 0, block = 0, cur = 0;
    std::vector<ucontext_t> ctx;
    std::vector<char *> stacks;
    std::vector<char> done;
    std::vector<WaveCtx> waves;
    ucontext_t sched;
    int bar_arrived = 0; unsigned bar_gen = 0; int bar_site = 0;
    std::vector<int> last_site;          /* per thread: line of the cross-lane op it last entered (-line: barrier) */
    unsigned long progress = 0;          /* bumped whenever a collective / barrier completes or a thread ends */
    std::function<void()> body;
};

inline State &S() { static State s; return s; }

inline int tid() { return S().cur; }
inline int nthreads() { return S().nthreads; }
inline int block() { return S().block; }
inline int nblocks() { return S().nblocks; }

inline void yield_() { State &s = S(); swapcontext(&s.ctx[s.cur], &s.sched); }

[[noreturn]] inline void die(const char *msg, int a, int b)
{
    fprintf(stderr, "fiber_emu: %s (%d vs %d) at thread %d\n", msg, a, b, S().cur);
    abort();
}

/* generic wave collective: every lane deposits v, gets the whole vector */
inline const uint64_t *collective(uint64_t v, int site)
{
    State &s = S();
    int lane = s.cur & 63;
    WaveCtx &w = s.waves[s.cur >> 6];
    unsigned g = w.gen;
    s.last_site[s.cur] = site;
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
    unsigned g = s.bar_gen;
    s.last_site[s.cur] = -site;
    if (s.bar_arrived == 0) s.bar_site = site;
    else if (s.bar_site != site) die("threads at different barriers, lines", s.bar_site, site);
    if (++s.bar_arrived == s.nthreads) { s.bar_arrived = 0; s.bar_gen++; s.progress++; }
    else while (s.bar_gen == g) yield_();
}

inline void trampoline()
{
    State &s = S();
    s.body();
    s.done[s.cur] = 1;
    swapcontext(&s.ctx[s.cur], &s.sched);
}

/* run `body` as a grid of nblocks x nthreads (nthreads a multiple of 64) */
inline void launch(int nblocks, int nthreads_, std::function<void()> body, size_t stack_bytes = 256 * 1024)
{
    State &s = S();
    if (nthreads_ % 64) die("block size must be a multiple of 64", nthreads_, 64);
    s.nthreads = nthreads_; s.nblocks = nblocks; s.body = body;
    s.ctx.resize(nthreads_); s.done.assign(nthreads_, 0); s.last_site.assign(nthreads_, 0);
    s.waves.assign(nthreads_ / 64, WaveCtx());
    if ((int) s.stacks.size() < nthreads_) {
        size_t old = s.stacks.size();
        s.stacks.resize(nthreads_);
        for (size_t i = old; i < (size_t) nthreads_; i++) s.stacks[i] = (char *) malloc(stack_bytes);
    }
    for (int b = 0; b < nblocks; b++) {
        s.block = b; s.bar_arrived = 0;
        for (auto &w : s.waves) { w.arrived = 0; }
        for (int t = 0; t < nthreads_; t++) {
            getcontext(&s.ctx[t]);
            s.ctx[t].uc_stack.ss_sp = s.stacks[t];
            s.ctx[t].uc_stack.ss_size = stack_bytes;
            s.ctx[t].uc_link = &s.sched;
            makecontext(&s.ctx[t], (void (*)()) trampoline, 0);
            s.done[t] = 0;
        }
        int remaining = nthreads_;
        unsigned long seen = s.progress; int idle = 0;
        while (remaining > 0) {
            int progressed = 0;
            /* deadlock detector: whole passes over the fibers without any collective completing */
            if (s.progress != seen) { seen = s.progress; idle = 0; }
            else if (++idle > 4) {
                fprintf(stderr, "fiber_emu: DEADLOCK in block %d -- threads wait at different cross-lane ops "
                                "(line, negative = block barrier):\n", b);
                for (int t = 0; t < nthreads_; t++)
                    if (!s.done[t] && (t % 64 == 0 || s.last_site[t] != s.last_site[t - 1]))
                        fprintf(stderr, "  thread %d.. : %d\n", t, s.last_site[t]);
                abort();
            }
            for (int t = 0; t < nthreads_; t++) {
                if (s.done[t]) continue;
                s.cur = t;
                swapcontext(&s.sched, &s.ctx[t]);
                progressed = 1;
                if (s.done[t]) { remaining--; s.progress++; }
            }
            if (!progressed) break;
        }
    }
}

} /* namespace emu */
#endif
