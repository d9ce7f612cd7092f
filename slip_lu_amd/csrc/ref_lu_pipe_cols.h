/* ref_lu_pipe_cols.h -- the column worker: gated sweep, pattern, pivot commit (stage 1), L/U write (stage 2),
 * and the REF triangular solves on resident factors.  Included by ref_lu_pipe.h (see the design notes there). */
#ifndef SLIP_REF_LU_PIPE_COLS_H
#define SLIP_REF_LU_PIPE_COLS_H

#ifdef SLIP_PROFILING
#define SLIP_TR(i) do { if (tid == 0) { const unsigned long long n_ = slip_clock(); trs_[i] = (int32_t)(n_ - trp_); trp_ = n_; } } while (0)
#else
#define SLIP_TR(i) do { } while (0)
#endif
#define SLIPDEV_ABORTED 100                 /* internal to the kernel: this worker's column can never commit */
#if defined(SLIP_EMULATE) && defined(SLIP_EMU_TRACE)
#define SLIP_WHY(...) do { if (slip_tid() == 0) fprintf(stderr, __VA_ARGS__); } while (0)
#else
#define SLIP_WHY(...) do { } while (0)
#endif
/* an inconsistency names itself in the state the host prints (site 100 + n, the column, two values) */
#define SLIP_SITE(n, a, b) do { if (slip_tid() == 0 && !st->dbg_who) { st->dbg_who = 100 + (n); st->dbg_k = k; st->dbg_a = (int32_t)(a); st->dbg_b = (int32_t)(b); } } while (0)

/* the worker's side of the committer protocol (ref_lu_pipe_commit.h) */
SLIP_DEV void slip_export_package(const SlipParams &P, const int k, uint32_t *lds, const int F0, const int Fl);
SLIP_DEV void slip_export_full(const SlipParams &P, const int k, uint32_t *lds, const int F0, const int Fl);
#ifndef SLIP_K1_NEAR
#define SLIP_K1_NEAR 4                      /* a full package goes out when the frontier is within this many columns of the column */
#endif
SLIP_DEV void slip_retract_package(const SlipParams &P, const int k, volatile int32_t *sv);

SLIP_DEV void slip_raise_stop(SlipState *st, int k, int status) { slip_agent_min_i64(&st->stop, ((int64_t) k << 8) | (int64_t) status); }

/* Wait until the commit frontier reaches `need` (need <= k).  Called by all threads; returns the frontier, or -1 when
 * column k can never commit (an earlier column stopped the factorisation, or a wait timed out).  once: one look only
 * (the frontier as it is, possibly below `need`). */
SLIP_DEV int slip_wait_frontier(SlipState *st, uint32_t *lds, int need, int k, int once = 0, const SlipParams *Pf = (const SlipParams *) 0)
{
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    slip_block_sync();
    if (slip_tid() == 0) {
        int res;
        unsigned long long spins = 0;
        for (;;) {
            /* ONE round of loads per look: the frontier word, the stop word and the line of hints go out together (each with a wait
             * of its own they made a look 10+ us long, and that is what stands between a commit and the next column seeing it) */
            int pr;
            const bool hints = !once && Pf && Pf->farm;
            SlipHints H;
            if (hints) H = slip_farm_hints_load(st);
            const int64_t stop = slip_ld_i64(&st->stop);
            const int F = slip_ld_frontier(st, &pr);
            if (F >= need) { res = F; sv[SV_TMP3] = pr; break; }       /* pr = row_perm[F-1], for free */
            if (once) { res = F; break; }                               /* a look, not a wait: the caller has something to do meanwhile */
            /* nothing to do but wait: is another worker's update queue open to helpers?  (-2 - slot: the caller helps, then waits again) */
            if (hints) { const int h = slip_farm_peek_loaded(*Pf, H, k - F <= SLIP_FARM_URGENT_DIST); if (h) { res = -1 - h; break; } }
            if ((stop >> 8) < (int64_t) k || (int)(stop & 0xFF) == SLIPDEV_INTERNAL) { res = -1; break; }
            /* the further from its turn, the longer between polls: the frontier word is one line for the whole chip */
            /* (the committer moves the frontier eight columns at a time: a worker within two batches of its turn polls at the short
             * interval -- a column that commits itself is on the commit chain from the moment the frontier reaches it) */
            const int dist = k - F;
            if (dist <= SLIP_POLL_NEAR) slip_sleep_short();
            else { const int reps = dist < SLIP_POLL_MAXREPS ? dist : SLIP_POLL_MAXREPS; for (int q = 0; q < reps; q++) slip_sleep(); }
            if (++spins > SLIP_SPIN_LIMIT) { st->dbg_who = 1; st->dbg_k = k; st->dbg_a = need; st->dbg_b = F; slip_raise_stop(st, 0, SLIPDEV_INTERNAL); res = -1; break; }
        }
        sv[SV_TMP2] = res;
    }
    slip_block_sync();
    return sv[SV_TMP2];
}

/* The ready frontier F2: every column below it has published its L entries and limbs (stage 2).  Stage 2 ends out of
 * order, so F2 is advanced over the per-column flags Lready[] by whoever finds it behind: the worker that has just
 * finished a column, and any worker that waits for it (so a missed advance cannot strand the pipeline).  All accesses
 * are returning agent-scope atomics: an advance that follows a flag store in program order also follows it in memory. */
SLIP_DEV int slip_advance_ready(const SlipParams &P, SlipState *st)
{
    int f2 = slip_agent_add_i32(&st->F2, 0);
    int pr_; const int F = slip_ld_frontier(st, &pr_);
    while (f2 < F && slip_agent_add_i32(P.Lready.at(f2), 0) != 0) {
        const int seen = slip_agent_cas_i32(&st->F2, f2, f2 + 1);
        f2 = seen == f2 ? f2 + 1 : seen;
    }
    return f2;
}

/* Wait until column need-1 has published its L entries and limbs (need <= the commit frontier this worker knows): its own
 * flag, not the contiguous ready frontier -- one heavy column that is still writing must not hold up the columns that do
 * not read it.  Called by all threads; returns the ready frontier (possibly still below `need`: a cache of "everything
 * below is ready"), or -1 when the launch is being given up. */
SLIP_DEV int slip_wait_ready(const SlipParams &P, SlipState *st, uint32_t *lds, int need)
{
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    slip_block_sync();
    if (slip_tid() == 0) {
        int res;
        unsigned long long spins = 0;
        for (;;) {
            /* what this wait is for is ONE flag: it is looked at first, next to the stop word (one round of loads per look; the
             * walk along the ready frontier -- a chain of returning atomics -- only once the flag is up: a reader of a late
             * source stands on the commit chain while it waits here) */
            const int rdy = slip_agent_load_i32(P.Lready.at(need - 1));
            const int64_t stop = slip_ld_i64(&st->stop);
            if (rdy != 0) { res = slip_advance_ready(P, st); break; }
            if ((int)(stop & 0xFF) == SLIPDEV_INTERNAL) { res = -1; break; }
            slip_sleep_short();
            if (++spins > SLIP_SPIN_LIMIT) { st->dbg_who = 2; st->dbg_k = sv[SV_K]; st->dbg_a = need; st->dbg_b = rdy; slip_raise_stop(st, 0, SLIPDEV_INTERNAL); res = -1; break; }
        }
        sv[SV_TMP2] = res;
    }
    slip_block_sync();
    return sv[SV_TMP2];
}

/* append this lane's newly discovered row to the worker's row list (slots per wave: one LDS atomic per wave) */
SLIP_DEV void slip_rlist_push(const SlipParams &P, volatile int32_t *sv, uint32_t *lds, int has, int row)
{
    const int lane = slip_lane();
    const uint64_t nm = slip_ballot(has);
    int base = 0;
    if (lane == 0 && nm) base = slip_atomic_add_i32((int32_t *) &sv[SV_NROWS], slip_popc64(nm));
    base = (int) slip_bcast0_u32((uint32_t) base);
    if (has) {
        const int at = base + slip_popc64(nm & ((1ull << lane) - 1ull));
        P.rlist[at] = row;
        if (at < SLIP_TAB_CAP) lds[SLIP_LDS_TAB + at] = (uint32_t) row;      /* short patterns are listed in LDS as well */
    }
}

/* ------------------------------------------------------------------ */
/* The pre-pass of the commit chain.  When a column's sweep has run dry and the frontier is still below k, everything the
 * pivot choice needs EXCEPT rho[k-1] is already known: which rows are pivotal, the rows' states, rho[h] of the rows that
 * were updated.  The final values are x * rho[k-1] / rho[h]: rho[k-1]'s bit length shifts every bound by the same amount,
 * so WHICH rows can be the pivot (bound against bound) does not depend on it.  The pre-pass classifies the rows, adds up
 * what the capacity checks need and lists the candidates; when the frontier arrives, one wave only has to multiply the
 * listed candidates, search among them and publish (slip_do_column, "the short commit chain").  Any source applied
 * afterwards invalidates it (sv[SV_PP] = 0: the sweep does that).
 * Row table entry f_inf = class | (c + SLIP_PP_BIAS) << 2 with  bits(value) <= c + bits(rho[k-1]):
 *   class 0: pivotal or zero; 2: never updated, one limb (S: lane product or straight into the L slab, slot handed out
 *   here); 3: anything else (B: a wave item at commit time).
 * sv[SV_PP + ..]: 0 valid, 1 candidates, 2 S rows, 3 pivotal rows, 4 their limbs, 5 limb bound of the B rows (without
 * rho[k-1]'s share), 6 B rows, 7 largest c of a B row (biased), 8 largest ctz of their rho[h], 9 largest c (biased),
 * 10 longest pivotal row (bits), 11 best bound, 12 candidates that are not class S, 13 table index + 1 of the diagonal row
 * when it is a candidate.  Called by all threads; barriers inside. */
#define SLIP_PP_BIAS   (1 << 20)
#define SLIP_PP_CAND   64               /* one candidate per lane of the committing wave */
/* Two refinements on top of the bounds (round 3):
 *  - class-S rows (one limb, never updated) all carry the SAME factor rho[k-1]: among them only the rows whose |a| equals
 *    the smallest (largest) |a| can be the pivot -- an exact comparison, so a column usually lists one candidate;
 *  - the FULL package (ref_lu_pipe_commit.h, chain engine): when every non-pivotal row is a one-limb value the whole
 *    column state can travel to the committer.  Rows updated at an older column h are brought to level Fl-1 for that
 *    copy (x * rho[Fl-1] / rho[h], exact: the history update of slip_REF_triangular_solve.c:139-149 to an intermediate
 *    level), so that the committer needs no pivot older than the package.  f_k0/f_k1: the value, f_meta: sign | h+1,
 *    f_npi: the row's place among the non-pivotal (bit 31 clear) or among the pivotal rows (bit 31 set).
 *    sv[SV_PPF] = number of non-pivotal rows, or -1 when some row does not qualify. */
SLIP_DEV void slip_prepass(const SlipParams &P, const int k, const int tag, uint32_t *lds, const int Fl, dig_t *b0)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint32_t *f_row = lds + SLIP_LDS_TAB, *f_pos = f_row + SLIP_TAB_CAP, *f_inf = f_row + 2 * SLIP_TAB_CAP, *f_aux = f_row + 3 * SLIP_TAB_CAP;
    uint32_t *f_k0 = lds + SLIP_LDS_KEYS, *f_k1 = f_k0 + SLIP_PAT_CAP;
    uint32_t *f_meta = lds + SLIP_LDS_DIROFF, *f_npi = lds + SLIP_LDS_ROWS;
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint32_t *cl = lds + SLIP_LDS_WORK + SLIP_CAND_CAP;
    const int scheme = P.pivot_scheme;
    const int kind = (scheme == 4 || scheme == 5) ? 1 : 0;            /* 0 smallest, 1 largest (first-nonzero does not come here) */
    const int diagpref = scheme == 1 || scheme == 3 || scheme == 4;
    const int col = P.q[k];
    const int nrows = sv[SV_NROWS];
    if (tid < SLIP_PP_WORDS) sv[SV_PP + tid] = tid == 11 && kind == 0 ? 0x7FFFFFFF : 0;
    if (tid == SLIP_PP_WORDS) { sv[SV_PPF] = 0; sv[SV_TMP] = 0; sv[SV_TMP2] = 0; }
    /* the full package is a possibility only with the engine running, a committed predecessor and room for the fill */
    /* (a worker that has just seen a multi-limb pivot does not try for the next 64 columns: the values only grow) */
    const bool want_full = P.engine && Fl >= 1 && nrows <= SLIP_TAB_CAP - SLIP_ENG_ROWS && !(sv[SV_NOENG] > 0 && k - sv[SV_NOENG] < 64);
    SlipPiv Mf = slip_piv_none();
    if (want_full) Mf = slip_ld_piv(P.piv.at(Fl - 1));
    const bool mf_small = want_full && slip_abs(Mf.len) <= 2;
    slip_block_sync();
    if (want_full && !mf_small && tid == 0) sv[SV_NOENG] = k;
    uint32_t ulimbs = 0, nUc = 0, sB = 0, nB = 0, maxcB = 0, maxzh = 0, maxc = 0, maxubp = 0;
    uint32_t best = kind == 0 ? 0x7FFFFFFFu : 0u;
    uint64_t smin = ~0ull;                      /* smallest |a| (largest: its complement) over this thread's class-S rows */
    int fullbad = mf_small ? 0 : 1;
    for (int t0 = 0; t0 < nrows; t0 += T) {
        const int t = t0 + tid;
        int cls = 0, c = 0, isS = 0, r = 0, isU = 0, isNP = 0; uint32_t asgn = 0;
        if (t < nrows) {
            r = (int) f_row[t];
            const int pos = slip_ld_i32(P.pinv.at(r));
            const SlipRow xr = P.xrow[r];
            f_pos[t] = (uint32_t) pos;                      /* as read at a frontier >= Fl: the swap log brings it to column k */
            if (pos < Fl) { isU = 1; ulimbs += (uint32_t) slip_limbs(xr.len); nUc++; if ((uint32_t) xr.bits > maxubp) maxubp = (uint32_t) xr.bits; }
            else if (xr.len == 0) { cls = 0; isNP = 1; f_k0[t] = 0u; f_k1[t] = 0u; f_meta[t] = 0u; }      /* a zero keeps no history (slip_REF_triangular_solve.c:175-196 overwrites it) */
            else if (xr.h < 0 && slip_abs(xr.len) <= 2) {
                cls = 2; isS = 1; isNP = 1; c = xr.bits;
                const uint64_t xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                f_k0[t] = (uint32_t) xv; f_k1[t] = (uint32_t)(xv >> 32);
                f_meta[t] = xr.len < 0 ? 0x80000000u : 0u;
                asgn = ((uint32_t) slip_abs(xr.len) << 12) | (xr.len < 0 ? 1u << 14 : 0u);
                const uint64_t key = kind == 0 ? xv : ~xv;
                if (key < smin) smin = key;
            } else {
                cls = 3; isNP = 1;
                int bh = 0, zh = 0;
                SlipPiv H = slip_piv_none();
                if (xr.h >= 0) { H = slip_ld_piv(P.piv.at(xr.h)); bh = H.bits; zh = H.ctz; }
                c = xr.bits - bh + (xr.h >= 0 ? 1 : 0);
                sB += (uint32_t)(((c > 0 ? c : 0) + 63) >> 6) + 1u; nB++;
                if ((uint32_t)(c + SLIP_PP_BIAS) > maxcB) maxcB = (uint32_t)(c + SLIP_PP_BIAS);
                if ((uint32_t) zh > maxzh) maxzh = (uint32_t) zh;
                /* the copy for a full package: a one-limb value, brought to level Fl-1 when it was updated before that */
                if (mf_small) {
                    int okf = slip_abs(xr.len) <= 2 && xr.h >= 0 && xr.h <= Fl - 1;
                    if (okf) {
                        const uint64_t xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                        slip_u128 y = (slip_u128) xv; int ys = slip_sgn(xr.len), hh = xr.h;
                        if (xr.h < Fl - 1) {
                            if (slip_abs(H.len) <= 2 && xr.bits + Mf.bits <= 126) {
                                y = slip_divexact128((slip_u128) xv * Mf.lo, H.lo, H.ctz, H.inv64);
                                ys *= slip_sgn(Mf.len) * slip_sgn(H.len); hh = Fl - 1;
                            } else okf = 0;
                        }
                        if (okf && (uint64_t)(y >> 64) != 0) okf = 0;
                        if (okf) { f_k0[t] = (uint32_t)(uint64_t) y; f_k1[t] = (uint32_t)((uint64_t) y >> 32); f_meta[t] = (ys < 0 ? 0x80000000u : 0u) | (uint32_t)(hh + 1); }
                    }
                    if (!okf) fullbad = 1;
                }
            }
        }
        /* S rows get their slots in the L slab now (slot index kept in the row's own x area, behind the value) */
        const uint64_t am = slip_ballot(isS);
        int abase = 0;
        if (lane == 0 && am) abase = slip_atomic_add_i32((int32_t *) &sv[SV_PP + 2], slip_popc64(am));
        abase = (int) slip_bcast0_u32((uint32_t) abase);
        if (isS) {
            const int si = abase + slip_popc64(am & ((1ull << lane) - 1ull));
            f_aux[t] = (uint32_t) si | asgn;
            (P.xd + (int64_t) r * P.xcap)[2] = (uint32_t) si;
        }
        /* places among the non-pivotal and among the pivotal rows (full packages; what the worker rebuilds its lists from) */
        if (want_full) {
            const uint64_t nm = slip_ballot(isNP), um = slip_ballot(isU);
            int nb_ = 0, ub_ = 0;
            if (lane == 0) {
                if (nm) nb_ = slip_atomic_add_i32((int32_t *) &sv[SV_TMP], slip_popc64(nm));
                if (um) ub_ = slip_atomic_add_i32((int32_t *) &sv[SV_TMP2], slip_popc64(um));
            }
            nb_ = (int) slip_bcast0_u32((uint32_t) nb_); ub_ = (int) slip_bcast0_u32((uint32_t) ub_);
            const uint64_t below = (1ull << lane) - 1ull;
            if (isNP) f_npi[t] = (uint32_t)(nb_ + slip_popc64(nm & below));
            else if (isU) f_npi[t] = 0x80000000u | (uint32_t)(ub_ + slip_popc64(um & below));
        }
        if (t < nrows) {
            f_inf[t] = (uint32_t) cls | ((uint32_t)(c + SLIP_PP_BIAS) << 2);
            if (cls) {
                const uint32_t ubc = (uint32_t)(c + SLIP_PP_BIAS), lbc = ubc - (cls == 2 ? 1u : 2u);
                if (kind == 0) { if (ubc < best) best = ubc; } else { if (lbc > best) best = lbc; }
                if (ubc > maxc) maxc = ubc;
            }
        }
    }
    {
        const uint32_t w_u = slip_wave_sum_u32(ulimbs), w_n = slip_wave_sum_u32(nUc), w_s = slip_wave_sum_u32(sB), w_nb = slip_wave_sum_u32(nB);
        const uint32_t w_cb = slip_wave_max_u32(maxcB), w_zh = slip_wave_max_u32(maxzh), w_c = slip_wave_max_u32(maxc), w_up = slip_wave_max_u32(maxubp);
        const uint32_t w_b = kind == 0 ? slip_wave_min_u32(best) : slip_wave_max_u32(best);
        const uint32_t w_fb = slip_wave_max_u32((uint32_t) fullbad);
        if (lane == 0) {
            if (w_n) { slip_atomic_add_i32((int32_t *) &sv[SV_PP + 3], (int) w_n); slip_atomic_add_i32((int32_t *) &sv[SV_PP + 4], (int) w_u); }
            if (w_nb) { slip_atomic_add_i32((int32_t *) &sv[SV_PP + 5], (int) w_s); slip_atomic_add_i32((int32_t *) &sv[SV_PP + 6], (int) w_nb); }
            slip_atomic_max_i32((int32_t *) &sv[SV_PP + 7], (int) w_cb); slip_atomic_max_i32((int32_t *) &sv[SV_PP + 8], (int) w_zh);
            slip_atomic_max_i32((int32_t *) &sv[SV_PP + 9], (int) w_c); slip_atomic_max_i32((int32_t *) &sv[SV_PP + 10], (int) w_up);
            if (kind == 0) slip_atomic_min_i32((int32_t *) &sv[SV_PP + 11], (int) w_b); else slip_atomic_max_i32((int32_t *) &sv[SV_PP + 11], (int) w_b);
            if (w_fb) slip_atomic_max_i32((int32_t *) &sv[SV_PPF], 1);
        }
    }
    const uint64_t sbest = slip_block_min_u64(smin, scan_tmp);       /* barriers inside: the sums above are complete behind it */
    const uint32_t bestb = (uint32_t) sv[SV_PP + 11];
    const int any = sv[SV_PP + 9] != 0;                            /* a nonzero non-pivotal row exists */
    for (int t0 = 0; t0 < nrows && any; t0 += T) {
        const int t = t0 + tid;
        int cand = 0;
        if (t < nrows) {
            const uint32_t inf = f_inf[t];
            const int cls = (int)(inf & 3u);
            if (cls) {
                const uint32_t ubc = inf >> 2, lbc = ubc - (cls == 2 ? 1u : 2u);
                cand = kind == 0 ? lbc <= bestb : ubc >= bestb;
                if (cand && cls == 2) {                     /* the exact test among the rows that share rho[k-1] as their only factor */
                    const uint64_t xv = (uint64_t) f_k0[t] | ((uint64_t) f_k1[t] << 32);
                    cand = (kind == 0 ? xv : ~xv) == sbest;
                }
                if (diagpref && (int) f_row[t] == col) { cand = 1; sv[SV_PP + 13] = t + 1; }
                if (cand && cls != 2) sv[SV_PP + 12] = 1;
            }
        }
        const uint64_t mC = slip_ballot(cand);
        int bC = 0;
        if (lane == 0 && mC) bC = slip_atomic_add_i32((int32_t *) &sv[SV_PP + 1], slip_popc64(mC));
        bC = (int) slip_bcast0_u32((uint32_t) bC);
        if (cand) { const int at = bC + slip_popc64(mC & ((1ull << lane) - 1ull)); if (at < SLIP_PP_CAND) cl[at] = (uint32_t) t; }
    }
    slip_block_sync();
    /* Class-B candidates (updated rows; their bounds have two bits of slack) keep a column from travelling as a package.  The
     * worker has time now: compare each of them EXACTLY with the best class-S candidate at the level of the latest pivot it
     * knows, rho[Fl-1] -- x rho[Fl-1] / rho[h] against a_best rho[Fl-1]; the common factor rho[k-1] / rho[Fl-1] that is
     * still to come does not change the order.  A B candidate strictly on the far side can never be the pivot and leaves
     * the list; if all of them do, the column is a candidates-only package after all. */
    if (sv[SV_PP + 12] && any && sbest != ~0ull && Fl >= 1 && sv[SV_PP + 1] >= 2 && sv[SV_PP + 1] <= SLIP_PP_CAND && P.committer) {
        const int ncand0 = sv[SV_PP + 1], wave = slip_wave(), nw = slip_nwaves();
        const uint64_t av = kind == 0 ? sbest : ~sbest;
        const uint32_t a0 = (uint32_t) av, a1 = (uint32_t)(av >> 32);
        const int nd = a1 ? 2 : 1;
        if (tid == 0) sv[SV_TMP3] = 0;                         /* B candidates that stay */
        slip_block_sync();
        for (int c = wave; c < ncand0; c += nw) {
            const int t = (int) cl[c];
            const int cls = (int)(f_inf[t] & 3u), r = (int) f_row[t];
            if (cls != 3) continue;
            int keep = 1;
            const int h = P.xrow[r].h;
            if (r != col && h >= 0 && h < Fl - 1) {
                const int cmp = slip_cand_compare_out(&P, r, Fl - 1, h, a0, a1, nd, b0);
                keep = kind == 0 ? !(cmp == 1) : !(cmp == -1);
            } else if (r != col && h == Fl - 1) {
                /* already at level Fl-1: its digits against a_best rho[Fl-1] -- left to the commit (rare) */
                keep = 1;
            }
            if (lane == 0) { if (keep) sv[SV_TMP3] = 1; else cl[c] = 0xFFFFFFFFu; }
        }
        slip_block_sync();
        if (!sv[SV_TMP3]) {
            if (tid == 0) {
                int m = 0;
                for (int c = 0; c < ncand0; c++) if (cl[c] != 0xFFFFFFFFu) cl[m++] = cl[c];
                sv[SV_PP + 1] = m; sv[SV_PP + 12] = 0;
            }
        } else if (tid == 0) {
            /* some B candidate stays: the list is as the bounds made it (the marks are undone from the table) */
            int m = 0;
            for (int t = 0; t < nrows && m < SLIP_PP_CAND; t++) {
                const uint32_t inf = f_inf[t];
                const int cls = (int)(inf & 3u);
                if (!cls) continue;
                const uint32_t ubc = inf >> 2, lbc = ubc - (cls == 2 ? 1u : 2u);
                int cand = kind == 0 ? lbc <= bestb : ubc >= bestb;
                if (cand && cls == 2) { const uint64_t xv = (uint64_t) f_k0[t] | ((uint64_t) f_k1[t] << 32); cand = (kind == 0 ? xv : ~xv) == sbest; }
                if (diagpref && (int) f_row[t] == col) cand = 1;
                if (cand) cl[m++] = (uint32_t) t;
            }
            sv[SV_PP + 1] = m;
        }
        slip_block_sync();
    }
    if (tid == 0) {
        sv[SV_PP] = any && sv[SV_PP + 1] >= 1 && sv[SV_PP + 1] <= SLIP_PP_CAND;
        sv[SV_PPF] = (want_full && !sv[SV_PPF]) ? sv[SV_TMP] : -1;
        sv[SV_PPFL] = Fl;
    }
    slip_block_sync();
}

/* ------------------------------------------------------------------ */
/* The ascending sweep over the pivotal positions < k of the pattern (slip_REF_triangular_solve.c:124-241).
 * GATED (a column of the factorisation running ahead of the commit frontier): a source at position jn is
 * applied once jn is below the frontier this worker knows (sv[SV_F]) and L(:,jn) is published; when no such
 * source is left and the frontier is still below k the worker waits for it to move and then looks at the rows
 * that became pivotal meanwhile (row_perm[c] for the new positions c).  Not GATED (k = n, the right-hand side
 * scattered instead of A(:,col)): the REF forward substitution (slip_forward_sub.c:61-158 is the same
 * recurrence over all positions) on complete factors.
 * Called by all threads; returns 1 if the column must be given up; the caller syncs and checks sv[SV_ERR]. */
template <bool FAST, bool GATED>
SLIP_DEV int slip_sweep(const SlipParams &P, SlipState *st, const int k, const int tag, uint32_t *lds, uint32_t *bm, dig_t *b0, dig_t *b1, dig_t *b2,
                        unsigned long long &c_read, unsigned long long &c_upd, unsigned long long &c_src, unsigned long long &c_str,
                        unsigned long long &c_mac, unsigned long long *t_wait, unsigned long long *t_last, int *cur_io = (int *) 0)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint32_t *work = lds + SLIP_LDS_WORK;
    /* cur_io: where a sweep that parked on a full package (return 2) is taken up again when the package comes back */
    int cur = cur_io ? *cur_io : -1, step = 0;
    if (cur_io) { slip_block_sync(); if (tid == 0) { sv[SV_CNT0] = 0; sv[SV_CNT0 + 1] = 0; sv[SV_CNT0 + 2] = 0; } }      /* the rotating queue counters start over with step 0 */
    (void) t_wait; (void) t_last;
    /* the pre-pass of the commit chain runs when the sweep is dry and the frontier has not moved (slip_prepass) */
    const bool pp_want = GATED && k >= 1 && !P.no_early && P.pivot_scheme != 2 && slip_nwaves() >= 2;
    int pp_fresh = 0;                            /* the pre-pass describes the rows as they are now */
    int pj = -1, pjn = -1;                       /* the source whose queued (wave) updates are pending */
    int dj = -1, dys = 1, dh = -1;               /* finalised one-limb source value not yet written back */
    slip_u128 dy = 0;
    for (;; step++) {
        slip_block_sync();
        /* write back the previous source's finalised value: every thread has read the old one by now */
        if (dj >= 0) { if (tid == 0) slip_store_small(P, dj, dy, dys, dh, tag); dj = -1; }
        /* queued multi-limb updates of the previous source: one wavefront each */
        const int nq = sv[SV_CNT0 + (step + 2) % 3];
        if (tid == 0) sv[SV_CNT0 + (step + 1) % 3] = 0;
        const int Fl = GATED ? sv[SV_F] : k;     /* everything below Fl is final */
        const int jn = slip_bitmap_next(bm, cur + 1, Fl);
        if (nq > 0) {
            slip_block_sync();
            slip_drain(P, lds, 1, pj, pjn, -1, slip_ld_i64(&P.Lp[pjn]), nq, work + ((step + 1) & 1) * 2 * SLIP_WORK_CAP, b0, b1, b2);
            if (sv[SV_ERR]) break;
        }
        if (jn < 0) {
            if (!GATED || Fl >= k) break;        /* the sweep is complete */
            /* wait for the frontier to move; the rows of the pattern that became pivotal are row_perm[c], c in [Fl, Fn) */
#ifdef SLIP_PROFILING
            const unsigned long long tw0_ = slip_clock();
#endif
            int Fn;
            /* still dry: the exported package holds for every pivot below the frontier this worker knows */
            if (GATED && pp_fresh && tid == 0 && sv[SV_PKGX]) slip_st_u32(P.pkg.at() + (int64_t)(k % P.nworkers) * SLIP_PKG_WORDS + SLIP_PKG_STAMP, (uint32_t) Fl);
            Fn = -2;
            if (pp_want && sv[SV_NROWS] <= SLIP_TAB_CAP) {
                if (!pp_fresh) {
                    Fn = slip_wait_frontier(st, lds, Fl + 1, k, 1);
                    if (Fn >= 0 && Fn <= Fl) {       /* nothing to do but wait: classify the rows and list the pivot candidates meanwhile */
                        slip_prepass(P, k, tag, lds, Fl, b0);
                        pp_fresh = 1;
#ifdef SLIP_EMU_TRACE
                        if (tid == 0) fprintf(stderr, "worker: col %d prepass valid %d ncand %d nonS %d nrows %d full %d committer %d\n", k, (int) sv[SV_PP], (int) sv[SV_PP + 1], (int) sv[SV_PP + 12], (int) sv[SV_NROWS], (int) sv[SV_PPF], P.committer);
#endif
                        Fn = -2;
                    }
                }
                if (pp_fresh && Fn == -2 && P.committer && !sv[SV_PKGX]) {
                    /* packages go out only once the committer workgroup has been seen running (a launch whose block 0 is not
                     * resident yet must not wait for it: those columns are committed by their workers) */
                    {
                        const int up_ = sv[SV_CUP];
                        slip_block_sync();                   /* every thread has read the flag before thread 0 may set it */
                        if (!up_) { if (tid == 0) sv[SV_CUP] = slip_ld_i32(&st->committer_up); slip_block_sync(); }
                    }
                    const bool can_pkg = sv[SV_CUP] && sv[SV_PKGVER] < 120 && k < (1 << 24) - 1;
                    /* every non-pivotal row a one-limb value: the FULL package -- the committer's chain engine applies whatever
                     * sources arrive after this frontier itself and hands the finished rows back (ref_lu_pipe_commit.h); this
                     * worker parks until then.  It goes out when the column's turn is NEAR: the engine is one workgroup, every source
                     * applied by the workers while they are far from their turn is work done in parallel. */
                    const bool k1_ok = P.engine && sv[SV_PPF] >= 1 && sv[SV_PPF] <= SLIP_PKG_FULLMAX && !sv[SV_NOK1] && sv[SV_PPFL] > sv[SV_K1STAMP];
                    if (can_pkg && k1_ok) {
                        if (k - Fl <= SLIP_K1_NEAR) {
                            if (tid == 0) { sv[SV_PKGF] = sv[SV_PPFL]; if (sv[SV_PKGVER]) slip_agent_add_u64(&st->c_retract, 1ull << 32); }
                            slip_export_full(P, k, lds, sv[SV_PPFL], Fl);
#ifdef SLIP_PROFILING
                            if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 1] = (int32_t) slip_realtime();  /* time line 1: package exported */
#endif
                            if (cur_io) { *cur_io = cur; return 2; }
                        }
                    } else if (can_pkg && sv[SV_PP] && !sv[SV_PP + 12] && sv[SV_NROWS] <= SLIP_PKG_NROWMAX && sv[SV_PP + 1] <= SLIP_PKG_CANDS) {
                        /* a column whose candidates are all one-limb values is handed to the committer */
                        if (tid == 0) { sv[SV_PKGF] = sv[SV_PPFL]; if (sv[SV_PKGVER]) slip_agent_add_u64(&st->c_retract, 1ull << 32); }
                        slip_export_package(P, k, lds, sv[SV_PPFL], Fl);
#ifdef SLIP_PROFILING
                        if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 1] = (int32_t) slip_realtime();  /* time line 1: package exported */
#endif
                    }
                }
            }
            /* the wait proper; a worker that waits helps with the long update queues of others (slip_farm_help) */
            while (Fn <= -2) {
                Fn = slip_wait_frontier(st, lds, Fl + 1, k, 0, &P);
                if (Fn <= -2) slip_farm_help(P, st, lds, -Fn - 2, b0, b1, b2);
            }
#ifdef SLIP_PROFILING
            *t_last = slip_clock(); t_wait[0] += *t_last - tw0_; t_wait[2] = slip_realtime();
#endif
            if (Fn < 0) return 1;
            const int Fseen = Fn, prow = sv[SV_TMP3];       /* the frontier word carried row_perm[Fseen-1] */
            if (Fn > k) Fn = k;
            const int nr = sv[SV_NROWS];
            if (nr <= SLIP_TAB_CAP) {
                /* the rows of the pattern are listed in LDS: every thread looks at its share for each new pivot row */
                const uint32_t *lrow = lds + SLIP_LDS_TAB;
                if (Fn - Fl == 1 && Fseen == Fn) {
                    for (int t = tid; t < nr; t += T) if ((int) lrow[t] == prow) slip_atomic_or_u32(&bm[Fl >> 5], 1u << (Fl & 31));
                } else {
                    /* several columns at once (the committer moves the frontier in batches; a worker far from its turn polls
                     * rarely): the new pivot rows come in ONE round of loads through the part of the work area the sweep
                     * does not use, not one dependent load per column */
                    uint32_t *pv = work + 4 * SLIP_WORK_CAP;
                    for (int c0 = Fl; c0 < Fn; c0 += 2 * SLIP_WORK_CAP) {
                        const int nc = Fn - c0 < 2 * SLIP_WORK_CAP ? Fn - c0 : 2 * SLIP_WORK_CAP;
                        slip_block_sync();
                        for (int e = tid; e < nc; e += T) pv[e] = (uint32_t) slip_ld_i32(P.row_perm.at(c0 + e));
                        slip_block_sync();
                        for (int t = tid; t < nr; t += T) {
                            const uint32_t r = lrow[t];
                            for (int e = 0; e < nc; e++) if (pv[e] == r) slip_atomic_or_u32(&bm[(c0 + e) >> 5], 1u << ((c0 + e) & 31));
                        }
                    }
                }
            } else
                for (int c = Fl + tid; c < Fn; c += T) {
                    const int r = slip_ld_i32(P.row_perm.at(c));
                    if (P.xrow[r].tag == tag) slip_atomic_or_u32(&bm[c >> 5], 1u << (c & 31));
                }
            if (tid == 0) sv[SV_F] = Fn;
            continue;
        }
        cur = jn;
        /* this source changes the rows: what the pre-pass found no longer holds (also after this sweep was taken up again behind a
         * full package that came back: pp_fresh is per call, the flags are the column's) */
        if (GATED) { pp_fresh = 0; if (tid == 0 && (sv[SV_PP] || sv[SV_PKGX])) { sv[SV_PP] = 0; if (sv[SV_PKGX]) slip_retract_package(P, k, sv); } }
        if (GATED && jn >= sv[SV_F2] && P.engine && pp_want && cur_io && !sv[SV_NOK1] && !(sv[SV_NOENG] > 0 && k - sv[SV_NOENG] < 64) && k - Fl <= SLIP_K1_NEAR && jn >= 1 && jn > sv[SV_K1STAMP]      /* (a package of this column, if any, has just been retracted) */
            && sv[SV_NROWS] <= SLIP_TAB_CAP) {
            /* this column's turn is near and the next source's L column is not published yet (its worker is still in stage 2):
             * rather than wait for it, hand the column to the chain engine with everything from position jn on still to be
             * applied -- the engine has the recent L columns in its LDS */
            slip_block_sync();
            if (tid == 0) sv[SV_TMP2] = slip_agent_add_i32(P.Lready.at(jn), 0) != 0;
            slip_block_sync();
            if (!sv[SV_TMP2]) {
                {
                    const int up_ = sv[SV_CUP];
                    slip_block_sync();
                    if (!up_) { if (tid == 0) sv[SV_CUP] = slip_ld_i32(&st->committer_up); slip_block_sync(); }
                }
                if (sv[SV_CUP] && sv[SV_PKGVER] < 120 && k < (1 << 24) - 1) {
                    slip_prepass(P, k, tag, lds, jn, b0);               /* rows at positions >= jn travel with their values */
                    if (sv[SV_PPF] >= 1 && sv[SV_PPF] <= SLIP_PKG_FULLMAX) {
                        if (tid == 0) { sv[SV_PKGF] = jn; sv[SV_PP] = 0; if (sv[SV_PKGVER]) slip_agent_add_u64(&st->c_retract, 1ull << 32); }
                        slip_export_full(P, k, lds, jn, jn);
                        *cur_io = jn - 1;                            /* taken up again AT this source should the package come back */
                        return 2;
                    }
                    if (tid == 0) { sv[SV_NOK1] = 1; sv[SV_PP] = 0; }   /* a value the engine does not take: no further attempts for this column */
                    slip_block_sync();
                }
            }
        }
        if (GATED && jn >= sv[SV_F2]) {
            /* the source is committed but its L column may still be on its way (stage 2 of column jn) */
#ifdef SLIP_PROFILING
            const unsigned long long tw0_ = slip_clock();
#endif
            const int f2 = slip_wait_ready(P, st, lds, jn + 1);
#ifdef SLIP_PROFILING
            t_wait[1] += slip_clock() - tw0_;
#endif
            if (f2 < 0) return 1;
            if (tid == 0) sv[SV_F2] = f2;
            slip_block_sync();
        }
        const int j = slip_ld_i32(P.row_perm.at(jn));
        SlipRow xj = P.xrow[j];
        uint64_t xjv = xj.len != 0 ? slip_limb0(P.xd + (int64_t) j * P.xcap) : 0;
        const SlipPiv R = slip_ld_piv(P.piv.at(jn));
        SlipPiv D = slip_piv_none();
        if (jn >= 1) D = slip_ld_piv(P.piv.at(jn - 1));
        /* bring x[j] to its final value: history update to level jn-1 (:139-149) */
        if (xj.len != 0 && xj.h < jn - 1) {
            slip_u128 y = 0; int ys = 1;
            if (slip_history_small(P, xj, xjv, D, xj.h, &y, &ys)) {
                /* every thread derives the same value in registers; thread 0 stores it one step later */
                dj = j; dy = y; dys = ys; dh = xj.h;
                const int yb = slip_bits128(y), yl = (yb + 31) >> 5;
                xj.len = ys < 0 ? -yl : yl; xj.bits = yb; xjv = (uint64_t) y;
            } else {
                slip_block_sync();
                if (wave == 0) {
                    if (slip_history_wave_out(&P, j, jn - 1, xj.h, b0, b1, b2)) { if (lane == 0) sv[SV_ERR] = 1; }
                }
                slip_block_sync();
                xj = P.xrow[j];
                xjv = slip_limb0(P.xd + (int64_t) j * P.xcap);
            }
        }
        const int src_nz = xj.len != 0;
        const int64_t m0 = slip_ld_i64(&P.Lp[jn]), m1 = slip_ld_i64(&P.Lp[jn + 1]);
        const int src_small = slip_abs(xj.len) <= 2 && slip_abs(R.len) <= 2 && slip_abs(D.len) <= 2;
        if (src_nz && tid == 0) {
            c_src++;
            c_read += 8ull * slip_limbs(R.len) + (jn >= 1 ? 8ull * slip_limbs(D.len) : 0ull);
        }
        uint32_t *wl = work + (step & 1) * 2 * SLIP_WORK_CAP;
        volatile int32_t *wcnt = &sv[SV_CNT0 + step % 3];
        /* stream L(:,jn): one entry per lane, SLIP_WORK_CAP entries per pass */
        for (int64_t mb = m0; mb < m1; mb += SLIP_WORK_CAP) {
            if (mb > m0) {
                /* long column: drain the queue of the previous pass before refilling it */
                slip_block_sync();
                if (dj >= 0) {                      /* the wave path reads x[j] from memory */
                    if (tid == 0) slip_store_small(P, dj, dy, dys, dh, tag);
                    dj = -1;
                    slip_block_sync();
                }
                const int nq2 = *wcnt;
                slip_drain(P, lds, 1, j, jn, -1, m0, nq2, wl, b0, b1, b2);
                if (tid == 0) *wcnt = 0;
                slip_block_sync();
            }
            const int64_t me = mb + SLIP_WORK_CAP < m1 ? mb + SLIP_WORK_CAP : m1;
            for (int64_t mm = mb; mm < me; mm += T) {
                const int64_t m = mm + tid;
                int queue = 0, qi = 0;                   /* this lane's update goes to the wave-item queue */
                int fresh = 0, fi = 0;                   /* this lane discovered a row */
                if (m < me) do {
                const int i = slip_ld_i32(&P.Li[m]);
                const SlipEnt le = slip_ld_ent(&P.Le[m]);
                /* structural discovery (what the reference's DFS does): a row not yet tagged with this column */
                SlipRow xi = P.xrow[i];
                if (xi.tag != tag) {
                    xi.len = 0; xi.h = -1; xi.bits = 0; xi.tag = tag; P.xrow[i] = xi;
                    fresh = 1; fi = i;
                }
                if (!src_nz) break;
                c_str++; c_read += 4 + 8ull * slip_limbs(le.len);
                /* L(:,jn) holds its own pivot row j and rows that were non-pivotal when it was built: those sit at
                 * positions above jn ever after (slip_REF_triangular_solve.c:160 `inew > jnew`) */
                if (i == j || le.len == 0) break;
                c_upd++;
                c_mac += (unsigned long long) slip_limbs(le.len) * slip_limbs(xj.len) + (unsigned long long) slip_limbs(xi.len) * slip_limbs(R.len);
                /* ---- one-limb operands: finish the update in this lane ---- */
                int done = 0;
                if (src_small && slip_abs(le.len) <= 2 && slip_abs(xi.len) <= 2) {
                    const int lx = xi.len != 0, has_d = jn >= 1;
                    const int hist = lx && has_d && xi.h < jn - 1, hdiv = hist && xi.h > -1;
                    SlipPiv H = slip_piv_none();
                    if (hdiv) H = slip_ld_piv(P.piv.at(xi.h));
                    const int bxp = !lx ? 0 : (!hist ? xi.bits : (hdiv ? xi.bits + D.bits - H.bits + 1 : xi.bits + D.bits));
                    const int b1b = lx ? bxp + R.bits : 0, b2b = le.bits + xj.bits;
                    const int bnum = (b1b > b2b ? b1b : b2b) + 1;
                    if (bnum <= 126 && (!hdiv || slip_abs(H.len) <= 2)) {
                        slip_u128 y = 0; int s1 = slip_sgn(xi.len) * slip_sgn(R.len);
                        if (lx) {
                            y = (slip_u128) slip_limb0(P.xd + (int64_t) i * P.xcap);
                            if (hist) { y *= D.lo; s1 *= slip_sgn(D.len); }
                            if (hdiv) { y = slip_divexact128(y, H.lo, H.ctz, H.inv64); s1 *= slip_sgn(H.len); }
                            y *= R.lo;
                        }
                        const slip_u128 p2 = (slip_u128) slip_limb0_s((const dig_t *)(P.Llimbs + le.off)) * xjv;
                        const int s2 = slip_sgn(le.len) * slip_sgn(xj.len);
                        slip_u128 mag; int sT;
                        if (!lx) { mag = p2; sT = -s2; }
                        else if (s1 == s2) { if (y >= p2) { mag = y - p2; sT = s1; } else { mag = p2 - y; sT = -s1; } }
                        else { mag = y + p2; sT = s1; }
                        if (has_d) { mag = slip_divexact128(mag, D.lo, D.ctz, D.inv64); sT *= slip_sgn(D.len); }
                        slip_store_small(P, i, mag, sT, jn, tag);
                        done = 1;
                    }
                }
                if (!done) { queue = 1; qi = i; }
                } while (0);
                /* new rows join the row list; a row that is pivotal below the frontier becomes a later source */
                slip_rlist_push(P, sv, lds, fresh, fi);
                if (fresh) {
                    const int pos = slip_ld_i32(P.pinv.at(fi));
                    if (pos < Fl) slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
                }
                /* queue slots per wave: one LDS atomic per wave instead of one per update on the same counter */
                const uint64_t qm = slip_ballot(queue);
                int qbase = 0;
                if (lane == 0 && qm) qbase = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(qm));
                qbase = (int) slip_bcast0_u32((uint32_t) qbase);
                if (queue) {
                    const int at = qbase + slip_popc64(qm & ((1ull << lane) - 1ull));
                    wl[2 * at] = (uint32_t)(m - m0); wl[2 * at + 1] = (uint32_t) qi;
                }
            }
        }
        pj = j; pjn = jn;
    }
    return 0;
}

/* bitmap -> pattern: positions in ascending order (what slip_sort_xi.c produces) in LDS (or P.pat when long);
 * *nU_out = how many of them are below k.  Called by all threads; ends before a barrier. */
SLIP_DEV void slip_pattern(const SlipParams &P, uint32_t *lds, const uint32_t *bm, int k, int *npat_out, int *nU_out)
{
    const int tid = slip_tid(), T = slip_nthreads();
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    const int nwords = P.bm_words;
    const int per = (nwords + T - 1) / T;
    int w0 = tid * per, w1 = w0 + per;
    if (w0 > nwords) w0 = nwords;
    if (w1 > nwords) w1 = nwords;
    uint64_t cntA = 0, cntU = 0;
    for (int w = w0; w < w1; w++) {
        const uint32_t word = bm[w];
        uint32_t below;
        if ((w + 1) * 32 <= k) below = word;
        else if (w * 32 >= k) below = 0;
        else below = word & ((1u << (k - w * 32)) - 1u);
        cntA += (uint64_t) slip_popc32(word); cntU += (uint64_t) slip_popc32(below);
    }
    uint32_t exA, exU, totA, totU_;              /* at most n < 2^31 set bits */
    slip_block_scan2_small((uint32_t) cntA, (uint32_t) cntU, scan_tmp, &exA, &exU, &totA, &totU_);
    {
        /* short patterns never leave the CU (the readers pick the same place by npat) */
        int o = (int) exA;
        int32_t *patl = (int32_t *)(lds + SLIP_LDS_PAT);
        const bool in_lds = totA <= SLIP_PAT_CAP;
        for (int w = w0; w < w1; w++) {
            uint32_t word = bm[w];
            while (word) {
                int b = slip_ctz32(word); word &= word - 1;
                if (in_lds) patl[o++] = w * 32 + b; else P.pat[o++] = w * 32 + b;
            }
        }
    }
    *npat_out = (int) totA; *nU_out = (int) totU_;
}

/* copy `len` digits (padded to whole limbs) from a private x row or the L slab to the L slab, write-through;
 * returns (wave-uniform) the number of trailing zero bits of the value; one wavefront */
SLIP_DEV int slip_publish_digits(dig_t *dst, const dig_t *src, int src_shared, int len)
{
    const int lane = slip_lane();
    const int lw = (len + 1) & ~1;
    int ctz = -1;
    for (int base = 0; base < lw; base += SLIP_WAVE) {
        const int c = base + lane;
        uint32_t v = 0;
        if (c < len) v = src_shared ? slip_ld_u32(src + c) : src[c];
        if (c < lw) slip_st_u32(dst + c, v);              /* also when it is in place already: plain-stored there, written through here */
        if (ctz < 0) {
            const uint64_t nz = slip_ballot(v != 0);
            if (nz) { const int t = slip_ctz64(nz); ctz = 32 * (base + t) + slip_ctz32(slip_shfl_u32(v, t)); }
        }
    }
    return ctz < 0 ? 0 : ctz;
}

/* Wait for the committer's verdict on this worker's exported package (its mailbox; not the frontier line).  Called by all
 * threads; a waiting worker helps with open update queues.  Returns 1 committed, 0 sent back, -1 the column can never commit. */
SLIP_DEV int slip_wait_verdict(const SlipParams &P, SlipState *st, uint32_t *lds, const int k, dig_t *b0, dig_t *b1, dig_t *b2)
{
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    const uint32_t *pk = P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS;      /* this worker's mailbox */
    for (;;) {
        slip_block_sync();
        if (slip_tid() == 0) {
            int res; unsigned long long spins = 0;
            for (;;) {
                /* (one round of loads per look: the hints, the stop word, the verdict) */
                SlipHints H;
                if (P.farm) H = slip_farm_hints_load(st);
                const int64_t stop = slip_ld_i64(&st->stop);
                const int v = (int) slip_ld_u32(pk + SLIP_PKG_OUT), mine_ = (sv[SV_PKGVER] << 24) | (k + 1);      /* a verdict names the version it is about */
                if (v == mine_) { res = 1; break; }
                if (v == -mine_) { res = 0; break; }
                if ((stop >> 8) < (int64_t) k || (int)(stop & 0xFF) == SLIPDEV_INTERNAL) { res = -1; break; }
                if (P.farm) { const int h = slip_farm_peek_loaded(P, H, 1); if (h) { res = -1 - h; break; } }
                slip_sleep_short();
                if (++spins > SLIP_SPIN_LIMIT) { st->dbg_who = 3; st->dbg_k = k; st->dbg_a = v; st->dbg_b = mine_; slip_raise_stop(st, 0, SLIPDEV_INTERNAL); res = -1; break; }
            }
            sv[SV_TMP2] = res;
            if (res == 0) sv[SV_PKGX] = 0;               /* sent back: this worker carries on with the column itself */
        }
        slip_block_sync();
        const int res = sv[SV_TMP2];
        if (res <= -2) { slip_farm_help(P, st, lds, -res - 2, b0, b1, b2); continue; }
        return res;
    }
}

/* A FULL package has been committed by the chain engine: the mailbox holds every row that was non-pivotal when the package
 * left, plus the rows later sources filled in, with their FINAL values (level k-1; a row that became pivotal meanwhile: its
 * value as the U entry) and their positions at column k.  The worker's row lists become: its own pivotal rows (final
 * since the export), then the rows handed back; the values go into its private x rows, so that the rest of the column
 * (pattern, table, L / U stores) runs as after any other early commit.  Called by all threads. */
SLIP_DEV void slip_takeover_full(const SlipParams &P, const int k, const int tag, uint32_t *lds, uint32_t *bm)
{
    const int tid = slip_tid(), T = slip_nthreads();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint32_t *f_row = lds + SLIP_LDS_TAB, *f_pos = f_row + SLIP_TAB_CAP;
    const uint32_t *f_npi = lds + SLIP_LDS_ROWS;
    uint32_t *t_row = lds + SLIP_LDS_KEYS, *t_pos = t_row + SLIP_PAT_CAP;
    const uint32_t *mb = P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS;
    const int nold = sv[SV_NROWS], nU = sv[SV_PP + 3];
    const int nfin = (int) slip_ld_u32(mb + SLIP_PKG_OUT + 5);
    slip_block_sync();
    for (int t = tid; t < nold; t += T) {
        const uint32_t pi = f_npi[t];
        if (pi >> 31) { t_row[pi & 0x7FFFFFFFu] = f_row[t]; t_pos[pi & 0x7FFFFFFFu] = f_pos[t]; }
    }
    const uint32_t *hb = mb + SLIP_MBOX_HDR;
    for (int t = tid; t < nfin; t += T) {
        const uint32_t r = slip_ld_u32(hb + t), vlo = slip_ld_u32(hb + SLIP_ENG_ROWS + t), vhi = slip_ld_u32(hb + 2 * SLIP_ENG_ROWS + t);
        const uint32_t me = slip_ld_u32(hb + 3 * SLIP_ENG_ROWS + t);
        t_row[nU + t] = r; t_pos[nU + t] = me & 0xFFFFFFu;
        const uint64_t mag = (uint64_t) vlo | ((uint64_t) vhi << 32);
        slip_store_small(P, (int) r, (slip_u128) mag, (me >> 31) ? -1 : 1, k - 1, tag);
    }
    slip_block_sync();
    const int nnew = nU + nfin;
    for (int t = tid; t < nnew; t += T) {
        const uint32_t r = t_row[t], pos = t_pos[t];
        f_row[t] = r; f_pos[t] = pos; P.rlist[t] = (int) r; P.rpos[t] = (int) pos;
        slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
    }
    if (tid == 0) sv[SV_NROWS] = nnew;
#if defined(SLIP_EMULATE) && defined(SLIP_EMU_TRACE)
    if (tid == 0) {
        fprintf(stderr, "takeover col %d: stamp %d nold %d nU %d nfin %d:", k, (int) sv[SV_PKGF], nold, nU, nfin);
        for (int t = 0; t < nnew; t++) fprintf(stderr, " %s%u@%u", t < nU ? "U" : "", t_row[t], t_pos[t]);
        fprintf(stderr, "\n");
    }
#endif
    slip_block_sync();
}

/* ------------------------------------------------------------------ */
/* one column; returns a SLIPDEV_* status (0 = committed), SLIPDEV_ABORTED when the column must be dropped */
/* ------------------------------------------------------------------ */
/* FAST: bitmap and wave scratch both in LDS (addresses provably LDS, ds_* instructions);
 * otherwise the generic build picks either place at run time (flat addressing). */
template <bool FAST>
SLIP_DEV int slip_do_column(const SlipParams &P, SlipState *st, const int k, const int tag, uint32_t *lds,
                            unsigned long long *acc_unused = (unsigned long long *) 0)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave(), nw = slip_nwaves();
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    const int col = P.q[k];
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    volatile int64_t *sv64 = (volatile int64_t *)(lds + SLIP_LDS_VARS);
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint32_t *work = lds + SLIP_LDS_WORK;
    uint32_t *bm = BM_LDS ? lds + SLIP_LDS_BITMAP : P.gbitmap;
    const int wcap = P.wcap;
    dig_t *b0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                        : P.gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    unsigned long long c_read = 0, c_upd = 0, c_src = 0, c_str = 0, c_mac = 0;
    SLIP_STAMP_INIT();

#ifdef SLIP_PROFILING
    if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k] = (int32_t) slip_realtime();              /* time line 0: the column starts */
#endif
    /* ---- phase 0: clear the pattern bitmap, take a snapshot of the commit frontier ---- */
    for (int w = tid; w < P.bm_words; w += T) bm[w] = 0;
    if (tid == 0) {
        sv[SV_ERR] = 0; sv[SV_CNT0] = 0; sv[SV_CNT0 + 1] = 0; sv[SV_CNT0 + 2] = 0; sv[SV_MAXDIG] = 0; sv[SV_NROWS] = 0; sv[SV_ACNT] = 0; sv[SV_EST] = -1; sv[SV_PP] = 0; sv[SV_PKGX] = 0; sv[SV_PKGVER] = 0; sv[SV_ABORT] = 0; sv[SV_PPF] = -1; sv[SV_PKGK] = 0; sv[SV_NOK1] = 0; sv[SV_K1STAMP] = 0;
        sv64[SV_LALLOC / 2] = 0; sv64[SV_LEXACT / 2] = 0;
        /* the ready frontier first: it never passes the commit frontier, also not between the two loads */
        sv[SV_F2] = slip_ld_i32(&st->F2);
        int pr_; int F = slip_ld_frontier(st, &pr_);
        sv[SV_F] = F < k ? F : k;
    }
    slip_block_sync();
    const int F0 = sv[SV_F];

    /* ---- phase 1: scatter A(:,col) into x (slip_REF_triangular_solve.c:105-119) ---- */
    for (int64_t p0 = P.Ap[col]; p0 < P.Ap[col + 1]; p0 += T) {
        const int64_t p = p0 + tid;
        const int have = p < P.Ap[col + 1];
        int row = 0;
        if (have) {
            row = P.Ai[p];
            const int pos = slip_ld_i32(P.pinv.at(row));
            if (pos < F0) slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));     /* pivotal below the frontier: a source */
            const int32_t al = P.Alen[p];
            const int la = slip_abs(al);
            const dig_t *src = (const dig_t *)(P.Alimbs + P.Aoff[p]);
            dig_t *X = P.xd + (int64_t) row * P.xcap;
            SlipRow r; r.len = al; r.h = -1; r.tag = tag; r.bits = 0;
            if (la > P.xcap) sv[SV_ERR] = 1;
            else {
                const int lw = (la + 1) & ~1;
                for (int c = 0; c < lw; c++) X[c] = c < la ? src[c] : 0u;
                r.bits = la ? 32 * la - slip_clz32(src[la - 1]) : 0;
            }
            P.xrow[row] = r;
            c_read += 4 + 8 * (unsigned long long)((la + 1) >> 1);
        }
        slip_rlist_push(P, sv, lds, have, row);
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;
    SLIP_STAMP(0);

    /* ---- phase 2: ascending sweep over the pivotal part of the pattern, ahead of the frontier ---- */
    unsigned long long t_wait_[3] = {0, 0, 0}, t_last_ = 0;
#ifdef SLIP_PROFILING
    int32_t trs_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long trp_ = 0;
#endif
    int full_adopt = 0;
    {
        int sweep_at = -1;
        for (;;) {
            const int sr = slip_sweep<FAST, true>(P, st, k, tag, lds, bm, b0, b1, b2, c_read, c_upd, c_src, c_str, c_mac, t_wait_, &t_last_, &sweep_at);
            if (sr == 1) return SLIPDEV_ABORTED;
            if (sr != 2) break;
            /* parked on a full package: the chain engine commits the column from it, or sends it back */
            const int v = slip_wait_verdict(P, st, lds, k, b0, b1, b2);
#ifdef SLIP_PROFILING
            if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 3] = (int32_t) slip_realtime();      /* time line 3: verdict seen */
#endif
            if (v < 0) return SLIPDEV_ABORTED;
            if (v == 1) { full_adopt = 1; break; }
            if (tid == 0) {
                /* sent back: for good (a value the engine does not handle), or until this worker has applied the source
                 * the engine could not (its column was committed elsewhere): the next full package must be a later one */
                const uint32_t *mb_ = P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS;
                if (slip_ld_u32(mb_ + SLIP_PKG_OUT + 1) != 1u) sv[SV_NOK1] = 1;
                sv[SV_K1STAMP] = sv[SV_PKGF];
            }
            slip_block_sync();
        }
    }
    if (full_adopt) slip_takeover_full(P, k, tag, lds, bm);
    slip_block_sync();
    if (sv[SV_ERR]) { if (sv[SV_ERR] >= 6 && sv[SV_ERR] != SLIPDEV_ABORTED) SLIP_SITE(11, sv[SV_ERR], 0); return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X; }
    SLIP_STAMP(1);
#ifdef SLIP_PROFILING
    trp_ = t_last_ ? t_last_ : slip_clock(); SLIP_TR(0);        /* 0: sweep tail */
    /* slot 1: the sweep's own work; 16/17: waiting for the commit / ready frontier; 19: columns counted */
    if (tid == 0) { prof_[1] -= t_wait_[0] + t_wait_[1]; prof_[16] += t_wait_[0]; prof_[17] += t_wait_[1]; prof_[19] += 1;
                    if (t_last_) prof_[20] += t_prev_ - t_last_; }      /* 20: sweep work after the last frontier wait (on the commit chain) */
#endif
    /* from here on F == k: columns 0..k-1 are committed, pinv / row_perm are those of the reference at column k */

    /* ---- phase 3: snapshot of the rows' positions and states.  F == k, so pinv is the reference's at column k; once
     *      this column has published its pivot (which may happen BEFORE its bulk is written, see the early commit
     *      below) later columns swap pinv / row_perm again, so everything after this point works on the snapshot.
     *      One round of loads: per row pinv[r] and the row state; rho[k-1]'s record; the column cursors. ---- */
    const int nrows = sv[SV_NROWS];
    const bool small = nrows <= SLIP_TAB_CAP;
    uint32_t *f_row = lds + SLIP_LDS_TAB, *f_pos = f_row + SLIP_TAB_CAP, *f_inf = f_pos + SLIP_TAB_CAP, *f_aux = f_inf + SLIP_TAB_CAP;
    uint32_t *f_k0 = lds + SLIP_LDS_KEYS, *f_k1 = f_k0 + SLIP_PAT_CAP;                    /* search keys of exact candidates */
    const int scheme = P.pivot_scheme;
    const int kind = (scheme == 2) ? 2 : ((scheme == 4 || scheme == 5) ? 1 : 0);   /* 0 smallest, 1 largest, 2 first nonzero */
    const bool try_early = k >= 1 && nrows <= SLIP_FAST_CAP && !P.no_early;
    /* the first round of loads of the commit chain: everything is issued before anything is waited for -- this thread's
     * row (its position and state), then rho[k-1]'s record, and the column cursors by six lanes of the last wave */
    /* the short commit chain: the pre-pass (slip_prepass, run while this worker waited) still describes the rows, so wave 0
     * alone multiplies the listed candidates, searches and publishes while the other waves take the position snapshot */
    /* a column whose package the committer holds: wait for the verdict (the outcome word of the package: this worker's
     * own line, not the frontier) */
    int adopted = 0;
#ifdef SLIP_PROFILING
    if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 5] = (int32_t) slip_realtime();          /* time line 5: the sweep has seen F >= k */
#endif
#ifdef SLIP_PROFILING
    /* a heavy column: the wall-clock time of every phase stamp (tools/phase_probe.py prints them) */
    if (nrows > 400 && tid == 0) { hstamp_ = P.dbg + 24 * (int64_t) P.n + 32 * (int64_t)(k & 63); for (int q_ = 0; q_ < 32; q_++) hstamp_[q_] = 0; hstamp_[31] = k; hstamp_[30] = nrows; hstamp_[29] = (int32_t) slip_realtime(); }
#endif
    const int packaged = !full_adopt && P.committer && try_early && sv[SV_PKGX];
    slip_block_sync();                                   /* (thread 0 clears the flag below) */
    if (full_adopt) adopted = 1;
    if (packaged) {
        const int v = slip_wait_verdict(P, st, lds, k, b0, b1, b2);
#ifdef SLIP_PROFILING
        if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 3] = (int32_t) slip_realtime();      /* time line 3: verdict seen */
#endif
        if (v < 0) return SLIPDEV_ABORTED;
        adopted = v;
    }
    const bool fastc = !adopted && try_early && nw >= 2 && sv[SV_PP] != 0;
    uint32_t *ppcl = work + SLIP_CAND_CAP;                 /* the pre-pass's candidate list (table indices) */
    int r0_ = 0, pos0_ = 0; SlipRow xr0_; xr0_.len = 0; xr0_.h = 0; xr0_.bits = 0; xr0_.tag = 0;
    int c_t = -1, c_r = 0, c_pos = 0; SlipRow c_x = xr0_;  /* short chain, wave 0: this lane's candidate */
    if (fastc) {
        if (wave == 0 && lane < sv[SV_PP + 1]) {
            c_t = (int) ppcl[lane]; c_r = (int) f_row[c_t];
            c_pos = slip_ld_i32(P.pinv.at(c_r));
            c_x = P.xrow[c_r];
        }
    } else if (tid < nrows) {
        r0_ = small ? (int) f_row[tid] : P.rlist[tid];
        pos0_ = slip_ld_i32(P.pinv.at(r0_));
        if (try_early) xr0_ = P.xrow[r0_];
    }
    SlipPiv M = slip_piv_none();
    if (k >= 1) M = slip_ld_piv(P.piv.at(k - 1));
    if (adopted) slip_agent_acquire();                   /* what the committer wrote for this column is read below */
    if (!fastc || wave == 0) {
        const int q_ = fastc ? lane - (SLIP_WAVE - 8) : tid - (T - 8);
        if (q_ == 0) sv64[SV_LNZ / 2] = slip_ld_i64(&P.Lp[k]);
        else if (q_ == 1) sv64[SV_LNL / 2] = slip_ld_i64(&P.Lo[k]);
        else if (q_ == 2) sv64[SV_UNZ / 2] = slip_ld_i64(&P.Up[k]);
        else if (q_ == 3) sv64[SV_UNL / 2] = slip_ld_i64(&P.Uo[k]);
        else if (q_ == 4) sv[SV_TMP3] = slip_ld_i32(P.pinv.at(col));        /* position of the "diagonal" row: fixed until this column's swap */
        else if (q_ == 5) sv[SV_TMP] = slip_ld_i32(P.row_perm.at(k));       /* the row the pivot will change places with                    */
        else if (q_ == 6) { sv[SV_CNT0] = 0; sv[SV_CNT0 + 1] = 0; sv[SV_CNT0 + 2] = 0; sv[SV_LISTN] = 0; }
    }
    const int lm = slip_abs(M.len), brho = M.bits;
    const int slot = (lm + 3) >> 1;
    /* rho[k-1]'s digits for the candidate multiplies are staged in LDS */
    const bool BMs = SCR_LDS && lm <= wcap;
    dig_t *Ms = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + nw * 3 * wcap : (dig_t *) 0;
    int pc_col = 0;
    int early = 0, e_pivrow = -1, e_pivpos = -1;
#ifdef SLIP_PROFILING
    unsigned long long tr_t2_ = slip_clock(), tr_sweep_ = t_last_ ? tr_t2_ - t_last_ : 0;
#endif
    /* where the digits of a row are: its x row (private), or (rows multiplied straight into L: h == -2) the slab (shared) */
    auto row_direct = [&](int r) -> int { return P.xrow[r].h == -2; };
    auto row_digits = [&](int r) -> const dig_t * {
        const dig_t *X = P.xd + (int64_t) r * P.xcap;
        return P.xrow[r].h == -2 ? (const dig_t *)(P.Llimbs + *(const int64_t *) X) : X;
    };
    /* the tolerance test of the diagonal preference between the best candidate `pr` and the diagonal row `col`
     * (slip_get_pivot.c:89-118, 126-146): 1 = take the diagonal; *err: scratch too small */
    auto diag_rule = [&](int pr, int *err) -> int {
        if (scheme == 1 || P.tol_mode == 0) return 1;
        const int lp_ = slip_abs(P.xrow[pr].len), lc_ = slip_abs(P.xrow[col].len);
        /* tol_m has exactly 53 bits, so tol_m*|den| has 52 or 53 bits more than |den|: most columns
         * are decided by the bit lengths alone */
        const int te0 = P.tol_e;
        const int bnum_ = (scheme == 3 ? P.xrow[pr].bits : P.xrow[col].bits) + (te0 < 0 ? -te0 : 0);
        const int bden_ = (scheme == 3 ? P.xrow[col].bits : P.xrow[pr].bits) + (te0 > 0 ? te0 : 0);
        if (bnum_ < 52 + bden_) return 0;
        if (bnum_ > 53 + bden_) return 1;
        /* exact comparison.  Every wave runs it redundantly in its own scratch.  A row that lives in the L
         * slab is shared data: its digits are staged (sc1 loads) into scratch the comparison does not need
         * at that point -- the denominator into b1 (free until the last step, when the denominator has
         * been consumed), the numerator behind the product in b2. */
        const int rn = scheme == 3 ? pr : col, rd = scheme == 3 ? col : pr;   /* |small|/|diag| or |diag|/|large| >= tol */
        const int ln = scheme == 3 ? lp_ : lc_, ldn = scheme == 3 ? lc_ : lp_;
        const dig_t *num = P.xd + (int64_t) rn * P.xcap, *den = P.xd + (int64_t) rd * P.xcap;
        if (ldn + 2 > wcap) { *err = 1; return 0; }
        if (row_direct(rd)) { slip_stage_shared(b1, row_digits(rd), ldn); den = b1; }
        if (row_direct(rn)) {
            if (ldn + 4 + ln > wcap) { *err = 1; return 0; }
            slip_stage_shared(b2 + ldn + 4, row_digits(rn), ln); num = b2 + ldn + 4;
        }
        const int tk = slip_tol_compare_out(P.tol_m, P.tol_e, num, ln, den, ldn, b0, b1, b2, wcap);
        if (tk < 0) { *err = 1; return 0; }
        return tk;
    };

    /* ---- the exact search among the candidates and stage 1 of the commit.  ONE wave runs this (the publishing wave rewrites
     * the pivot row's state: nobody else may be looking at it); the outcome lands in sv[SV_EPR], sv[SV_EPP], sv[SV_EST]. */
    struct CommitArgs { int ncand, diag_cand, nA, nLc, narith, slotw, sync; uint32_t nUc_all; uint64_t U_l, Lb_total;
                        int64_t Lnz_, Lnl_, Unz_, Unl_; const uint32_t *cl; };
    dig_t *stage = lds + SLIP_LDS_PAT;       /* a candidate's product is also left in LDS (slots over the areas phase 3c fills later) */
    auto search_publish = [&](const CommitArgs &A) {
            /* the exact search among the candidates (all exact now), slip_get_pivot.c:58-155: (bit length, leading bits)
             * keys; the candidates that tie on the best key are compared exactly, then by position.  Every wave does the
             * whole (short) search by itself: lanes = candidates, no workgroup barrier. */
            auto key_of = [&](int t) -> uint64_t {
                if (kind == 2) return (uint64_t) f_pos[t];
                const int cls = (int)(f_inf[t] & 3u);
                uint64_t key = (uint64_t) f_k0[t] | ((uint64_t) f_k1[t] << 32);
                if (cls == 3 || (cls == 1 && key == ~0ull)) {          /* value produced by a wave item / final before: key from its digits */
                    const int r = (int) f_row[t];
                    const SlipRow xr = P.xrow[r];
                    const uint64_t top = slip_top64(row_digits(r), slip_abs(xr.len), xr.h == -2);
                    key = ((uint64_t) xr.bits << 40) | (top >> 24);
                    if (kind == 1) key = ~key;
                }
                return key;
            };
            int est = 0;
            uint64_t mk = ~0ull;
            for (int c0 = 0; c0 < A.ncand; c0 += SLIP_WAVE) {
                const int c = c0 + lane;
                const uint64_t key = c < A.ncand ? key_of((int) A.cl[c]) : ~0ull;
                /* 64-bit minimum over the wave: the high words first, the low words among the lanes that hold the minimum */
                const uint32_t mh = slip_wave_min_u32((uint32_t)(key >> 32));
                const uint32_t ml = slip_wave_min_u32((uint32_t)(key >> 32) == mh ? (uint32_t) key : 0xFFFFFFFFu);
                const uint64_t wm = ((uint64_t) mh << 32) | ml;
                if (wm < mk) mk = wm;
            }
            int bt = -1;
            const int kbits = kind == 2 ? 0 : (int)((kind == 0 ? mk : ~mk) >> 40);
            for (int c0 = 0; c0 < A.ncand; c0 += SLIP_WAVE) {
                const int c = c0 + lane;
                const int t = c < A.ncand ? (int) A.cl[c] : -1;
                uint64_t tie = slip_ballot(t >= 0 && key_of(t) == mk);
                while (tie) {
                    const int l = slip_ctz64(tie); tie &= tie - 1;
                    const int tt = (int) slip_readlane((uint32_t) t, l);
                    if (bt < 0) { bt = tt; continue; }
                    const int rb = (int) f_row[bt], rt = (int) f_row[tt];
                    int cmp = 0;
                    if (kbits > 40)      /* at most 40 bits: equal keys are equal values */
                        cmp = slip_cmp_mag(row_digits(rb), row_direct(rb), row_digits(rt), row_direct(rt), slip_abs(P.xrow[rt].len));
                    if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0) || (cmp == 0 && f_pos[tt] < f_pos[bt])) bt = tt;
                }
            }
            if (bt < 0) { est = SLIPDEV_INTERNAL; bt = 0; if (lane == 0 && !st->dbg_who) { st->dbg_who = 109; st->dbg_k = k; st->dbg_a = A.ncand; } }
            e_pivrow = (int) f_row[bt]; e_pivpos = (int) f_pos[bt];
            int stg = (int)((f_inf[bt] >> 26) & 31u) - 1;     /* LDS slot of the pivot's digits, or -1 */
            /* the diagonal preference (slip_get_pivot.c:68-76, 89-118, 126-146); col's value is exact: it was a candidate */
            if (!est && A.diag_cand && e_pivrow != col) {
                int derr = 0;
                const int take = diag_rule(e_pivrow, &derr);
                if (derr) est = SLIPDEV_GROW_X;
                else if (take) { e_pivrow = col; e_pivpos = pc_col; stg = -1; }
            }
            SLIP_STAMP(13);
            if (A.sync) slip_block_sync_named(1);             /* short chain: the other waves have taken their position snapshot */
            if (est) { if (lane == 0) sv[SV_EST] = est; }
            else {
            SLIP_TR(6);                                       /* 6: search + diag */
            /* stage 1, early: the pivot's digits written through to the L slab (its class-A slot, or the reserved slot behind
             * those), the pivot record, the permutation swap, the column pointers (limb offsets from the bounds), ONE
             * drain, the frontier */
            SlipRow pxr; int pdirect; int64_t poff;
            if (stg >= 0) {
                /* a class-A candidate multiplied a moment ago: everything about it is in LDS */
                uint64_t key = (uint64_t) f_k0[bt] | ((uint64_t) f_k1[bt] << 32);
                if (kind == 1) key = ~key;
                pxr.bits = (int)(key >> 40);
                const int len_ = (pxr.bits + 31) >> 5;
                pxr.len = (f_inf[bt] >> 31) ? -len_ : len_; pxr.h = -2; pxr.tag = tag;
                pdirect = 1; poff = A.Lnl_ + (int64_t)(f_aux[bt] & 0x3FFu) * slot;
            } else {
                pxr = P.xrow[e_pivrow];
                pdirect = pxr.h == -2;
                poff = pdirect ? *(const int64_t *)(P.xd + (int64_t) e_pivrow * P.xcap) : A.Lnl_ + (int64_t) A.nA * slot;
            }
            const uint64_t plimbs = (uint64_t) slip_limbs(pxr.len);
            {
                const int lp_ = slip_abs(pxr.len);
                dig_t *dst = (dig_t *)(P.Llimbs + poff);
                /* where the digits are: the LDS slot the multiplying wave left, the slab (a class-A row that was not staged:
                 * its stores were drained above), or the row's private x */
                const dig_t *src = stg >= 0 ? (const dig_t *)(stage + stg * A.slotw) : (pdirect ? (const dig_t *) dst : P.xd + (int64_t) e_pivrow * P.xcap);
                const int z = slip_publish_digits(dst, src, stg < 0 && pdirect, lp_);
                const uint64_t lo64 = (stg < 0 && pdirect) ? slip_ld_u64((const uint64_t *) dst) : *(const uint64_t *) src;
                if (lane == 0) {
                    SlipPiv pr; pr.off = poff; pr.len = pxr.len; pr.bits = pxr.bits; pr.ctz = z; pr.invlen = 0;
                    pr.lo = lo64; pr.inv64 = 0; pr.pad = 0;
                    if (lp_ <= 2) pr.inv64 = slip_inv64(pr.lo >> z);
                    slip_st_piv(P.piv.at(k), pr);
                    const int intermed = e_pivpos, intermed2 = sv[SV_TMP];
                    slip_st_i32(P.row_perm.at(k), e_pivrow); slip_st_i32(P.row_perm.at(intermed), intermed2);
                    slip_st_i32(P.pinv.at(e_pivrow), k); slip_st_i32(P.pinv.at(intermed2), intermed);
                    slip_st_i32(P.sw_row.at(k), intermed2); slip_st_i32(P.sw_pos.at(k), intermed);
                    slip_st_i64(&P.Up[k + 1], A.Unz_ + (int) A.nUc_all + 1); slip_st_i64(&P.Lp[k + 1], A.Lnz_ + A.nLc);
                    slip_st_i64(&P.Uo[k + 1], A.Unl_ + (int64_t)(A.U_l + plimbs)); slip_st_i64(&P.Lo[k + 1], A.Lnl_ + (int64_t) A.Lb_total);
                }
                SLIP_TR(7);                                       /* 7: publish stores issued */
                slip_vm_drain();                                  /* the digits (all lanes) and the records (lane 0) have left */
                if (lane == 0) {
                    slip_st_frontier(st, k + 1, e_pivrow);
#ifdef SLIP_PROFILING
                    P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 2] = (int32_t) slip_realtime();      /* time line 2: committed (by its worker) */
                    if (t_last_) prof_[18] += slip_clock() - t_last_;
                    {
                        int32_t *tr = P.dbg + 8 * (int64_t) k; const unsigned long long nowc = slip_clock();
                        tr[0] = t_last_ ? (int32_t)(nowc - t_last_) : -1; tr[1] = 1; tr[2] = A.narith; tr[3] = nrows;
                        tr[4] = (int32_t) tr_sweep_; tr[5] = (int32_t)(nowc - tr_t2_); tr[7] = P.worker;
                        tr[6] = (int32_t) slip_realtime(); P.dbg[8 * (int64_t) P.n + k] = (int32_t) t_wait_[2];
                        for (int q_ = 0; q_ < 8; q_++) P.dbg[9 * (int64_t) P.n + 8 * (int64_t) k + q_] = trs_[q_];
                        P.dbg[9 * (int64_t) P.n + 8 * (int64_t) k + 7] = (int32_t)(nowc - trp_) + trs_[7] * 0;     /* 7b: drain */
                    }
#endif
                    /* from now on the pivot row lives in the slab like a class-A row */
                    if (!pdirect) {
                        SlipRow nr = pxr; nr.h = -2; P.xrow[e_pivrow] = nr;
                        *(int64_t *)(P.xd + (int64_t) e_pivrow * P.xcap) = poff;
                    }
                    sv64[SV_LALLOC / 2] = (int64_t)((uint64_t) A.nA * (uint64_t) slot + (pdirect ? 0ull : plimbs));
                    sv[SV_EPR] = e_pivrow; sv[SV_EPP] = e_pivpos; sv[SV_EST] = 0;
                }
            }
            }       /* !est */
    };

    /* ---- the short commit chain (see slip_prepass) ---- */
    int ec = -1;                                 /* 0: committed early; > 0: a status; -1: the full pass below decides */
    if (adopted) {
        /* the committer has published this column's pivot: take over the outcome from this worker's mailbox */
        const uint32_t *pk = P.pkg.at() + (int64_t) P.nworkers * SLIP_PKG_WORDS + (int64_t) P.worker * SLIP_MBOX_WORDS;
        if (BMs) { const dig_t *Mg = slip_piv_digits(P, M); for (int c = tid; c < lm; c += T) Ms[c] = slip_ld_u32(Mg + c); }
        /* the position snapshot (pinv as the reference has it at column k): the value the pre-pass read at frontier stamp0, or
         * where the LAST swap in [stamp0, k) that displaced the row put it (positions of non-pivotal rows only ever grow, and a
         * value read late already shows the swaps before it).  The log goes through LDS in pieces. */
        if (!full_adopt) {                               /* (a full package comes back with the positions at column k) */
            const int stamp0 = sv[SV_PKGF];
            uint32_t *lg_row = lds + SLIP_LDS_KEYS, *lg_pos = lg_row + SLIP_PAT_CAP;
            int myr[4] = {-1, -1, -1, -1}, myp[4] = {0, 0, 0, 0};      /* SLIP_PKG_NROWMAX rows over at least 128 threads */
            for (int q = 0; q < 4; q++) { const int t = tid + q * T; if (t < nrows) { myr[q] = (int) f_row[t]; myp[q] = (int) f_pos[t]; } }
            for (int e0 = stamp0; e0 < k; e0 += SLIP_PAT_CAP) {
                const int ne = k - e0 < SLIP_PAT_CAP ? k - e0 : SLIP_PAT_CAP;
                slip_block_sync();
                for (int e = tid; e < ne; e += T) { lg_row[e] = (uint32_t) slip_ld_i32(P.sw_row.at(e0 + e)); lg_pos[e] = (uint32_t) slip_ld_i32(P.sw_pos.at(e0 + e)); }
                slip_block_sync();
                for (int q = 0; q < 4; q++)
                    if (myr[q] >= 0) for (int e = 0; e < ne; e++) if ((int) lg_row[e] == myr[q]) myp[q] = (int) lg_pos[e];
            }
            slip_block_sync();
            for (int q = 0; q < 4; q++) {
                const int t = tid + q * T;
                if (t < nrows) { slip_atomic_or_u32(&bm[myp[q] >> 5], 1u << (myp[q] & 31)); f_pos[t] = (uint32_t) myp[q]; }
            }
        }
        if (tid == 0) {
            const int pr = (int) slip_ld_u32(pk + SLIP_PKG_OUT + 1);
            SlipRow nr; nr.len = (int32_t) slip_ld_u32(pk + SLIP_PKG_OUT + 3); nr.h = -2; nr.bits = (int32_t) slip_ld_u32(pk + SLIP_PKG_OUT + 4); nr.tag = tag;
            P.xrow[pr] = nr;
            *(int64_t *)(P.xd + (int64_t) pr * P.xcap) = (int64_t) slip_ld_u64((const uint64_t *)(pk + SLIP_PKG_OUT + 6));
            sv64[SV_LALLOC / 2] = (int64_t) slip_ld_u64((const uint64_t *)(pk + SLIP_PKG_OUT + 8));
            sv[SV_EPR] = pr; sv[SV_EPP] = (int) slip_ld_u32(pk + SLIP_PKG_OUT + 2);
        }
        slip_block_sync();
        pc_col = sv[SV_TMP3];
        ec = 0; early = 1; e_pivrow = sv[SV_EPR]; e_pivpos = sv[SV_EPP];
        SLIP_STAMP(6);
    }
    if (fastc) {
        if (wave == 0) {
            const int ncand = sv[SV_PP + 1], nS = sv[SV_PP + 2], nB = sv[SV_PP + 6];
            const uint32_t nUc_all = (uint32_t) sv[SV_PP + 3];
            const uint64_t U_l = (uint64_t)(uint32_t) sv[SV_PP + 4];
            /* rho[k-1]'s digits into LDS (phase 4 uses the copy as well) */
            if (BMs) { const dig_t *Mg = slip_piv_digits(P, M); for (int c = lane; c < lm; c += SLIP_WAVE) Ms[c] = slip_ld_u32(Mg + c); }
            slip_wave_sync();                    /* the cursors and the digits are in LDS */
            SLIP_TR(1);                          /* 1: the chain's one round of loads */
            pc_col = sv[SV_TMP3];
            const int64_t Lnz_ = sv64[SV_LNZ / 2], Lnl_ = sv64[SV_LNL / 2], Unz_ = sv64[SV_UNZ / 2], Unl_ = sv64[SV_UNL / 2];
            /* the bounds with rho[k-1]'s share added; the same checks as the full pass makes */
            const bool A_ok = lm + 2 <= P.xcap && lm + 2 <= 256;
            const int nA = lm > 2 ? nS : 0;      /* one limb times a one-limb pivot stays in the lane, anything longer goes into the slab */
            const int maxc = sv[SV_PP + 9] - SLIP_PP_BIAS + brho;
            const int maxub_all = maxc > sv[SV_PP + 10] ? maxc : sv[SV_PP + 10];
            const uint64_t L_b = (uint64_t)(uint32_t) sv[SV_PP + 5] + (uint64_t) nB * (uint64_t)((brho + 63) >> 6) + (lm <= 2 ? 2ull * (uint64_t) nS : 0ull);
            const uint64_t preserve = (uint64_t)((maxub_all + 63) >> 6) + 1;
            const uint64_t Lb_total = (uint64_t) nA * (uint64_t) slot + preserve + L_b;
            const uint64_t Ub_total = U_l + preserve;
            const int nLc = nrows - (int) nUc_all;
            int ok = 1;
            if (lm > 2 && !A_ok && nS > 0) ok = 0;
            if (nB > 0) {
                const int Wn = ((sv[SV_PP + 7] - SLIP_PP_BIAS + brho + 31) >> 5) + ((sv[SV_PP + 8] + 31) >> 5) + 1;
                if (Wn > P.wcap || Wn > P.xcap || Wn > P.invcap || lm > P.wcap) ok = 0;
            }
            if (Lnz_ + nLc > P.Lcap_nz || Lnl_ + (int64_t) Lb_total > P.Lcap_nl) ok = 0;
            if (Unz_ + (int) nUc_all + 1 > P.Ucap_nz || Unl_ + (int64_t) Ub_total > P.Ucap_nl) ok = 0;
            if (P.limb_cap > 0 && (int)((maxub_all + 63) >> 6) > P.limb_cap) ok = 0;
#ifdef SLIP_PROFILING
            if (lane == 0) P.dbg[17 * (int64_t) P.n + k] |= 0x100 | (ok ? 0x200 : 0) | ((lm > 2 && !A_ok && nS > 0) ? 0x400 : 0)
                | ((Lnz_ + nLc > P.Lcap_nz || Lnl_ + (int64_t) Lb_total > P.Lcap_nl) ? 0x800 : 0)
                | ((Unz_ + (int) nUc_all + 1 > P.Ucap_nz || Unl_ + (int64_t) Ub_total > P.Ucap_nl) ? 0x1000 : 0)
                | ((P.limb_cap > 0 && (int)((maxub_all + 63) >> 6) > P.limb_cap) ? 0x2000 : 0);
#endif
            const int diag_cand = (scheme == 1 || scheme == 3 || scheme == 4) && pc_col >= k && P.xrow[col].tag == tag && P.xrow[col].len != 0;
            const int slotw = (lm + 5) & ~1;
            const int nstage = (3 * SLIP_PAT_CAP) / slotw < 30 ? (3 * SLIP_PAT_CAP) / slotw : 30;
            uint32_t *wlB = work, *wlA = work + 2 * SLIP_CAND_CAP;
            int ncA = 0, ncB = 0;
            if (ok) {
                /* every lane brings its candidate to level k-1, or lists it for the wave: A -> 5-word record, B -> history item */
                int wantA = 0, wantB = 0;
                if (c_t >= 0) {
                    const uint32_t inf = f_inf[c_t];
                    const int cls = (int)(inf & 3u), ub = (int)(inf >> 2) - SLIP_PP_BIAS + brho;
                    int done = 0;
                    if (slip_abs(c_x.len) <= 2 && lm <= 2) {
                        const uint64_t xv = cls == 2 ? ((uint64_t) f_k0[c_t] | ((uint64_t) f_k1[c_t] << 32)) : slip_limb0(P.xd + (int64_t) c_r * P.xcap);
                        slip_u128 y = 0; int ys = 1;
                        if (slip_history_small(P, c_x, xv, M, c_x.h, &y, &ys)) {
                            slip_store_small(P, c_r, y, ys, k - 1, tag);
                            const int yb = slip_bits128(y);
                            const uint64_t top = yb ? (uint64_t)((y << (128 - yb)) >> 64) : 0ull;
                            uint64_t key = ((uint64_t) yb << 40) | (top >> 24);
                            if (kind == 1) key = ~key;
                            f_k0[c_t] = (uint32_t) key; f_k1[c_t] = (uint32_t)(key >> 32);
                            f_inf[c_t] = 1u | ((uint32_t) yb << 2);
                            done = 1;
                        }
                    }
                    if (!done) {
                        if (cls == 2) wantA = 1; else wantB = 1;
                        f_inf[c_t] = (uint32_t) cls | ((uint32_t)(ub > 1 ? ub : 1) << 2);
                    }
                    f_pos[c_t] = (uint32_t) c_pos;
                }
                const uint64_t mA = slip_ballot(wantA), mB = slip_ballot(wantB), below = (1ull << lane) - 1ull;
                ncA = slip_popc64(mA); ncB = slip_popc64(mB);
                if (wantA) {
                    const int at = slip_popc64(mA & below);
                    const uint32_t ax = f_aux[c_t];
                    wlA[5 * at] = (uint32_t) c_r; wlA[5 * at + 1] = f_k0[c_t]; wlA[5 * at + 2] = f_k1[c_t];
                    wlA[5 * at + 3] = ((uint32_t) c_t << 3) | ((ax >> 14) & 1u ? 4u : 0u) | ((ax >> 12) & 3u);
                    wlA[5 * at + 4] = (ax & 0x3FFu) * (uint32_t) slot;
                } else if (wantB) wlB[slip_popc64(mB & below)] = (uint32_t) c_r;
                slip_wave_sync();
            }
            /* the candidates' arithmetic is shared by all waves (a class-B candidate is a multi-limb history update with a
             * division, ~10 us: eight of them on one wave were most of this column's time on the commit chain) */
            if (lane == 0) { sv[SV_CNT0 + 1] = ncA; sv[SV_CNT0 + 2] = ncB; sv[SV_LISTN] = ok; }
            slip_block_sync_named(1);            /* the lists are there; the other waves have taken their position snapshot */
            if (ok) {
                if (ncA > 0) {
                    const SlipCandOut co = { f_k0, f_k1, f_inf, stage, slotw, nstage, kind };
                    const int e = slip_mul_rows_any(P, M, BMs ? Ms : slip_piv_digits(P, M), BMs ? 0 : 1, wlA, 0, nw, ncA, Lnl_, (uint32_t *) 0, (uint32_t *) 0, tag, &co);
                    if (e && lane == 0) sv[SV_ERR] = 1;
                    slip_vm_drain();
                }
                for (int t = 0; t < ncB; t += nw) {
                    const int e = slip_run_item_out(&P, 2, 0, 0, k, 0, wlB[t], 0u, b0, b1, b2);
                    if (e && lane == 0) sv[SV_ERR] = e;
                }
                slip_wave_sync();
            }
            slip_block_sync_named(2);            /* every candidate is at level k-1 */
            SLIP_TR(5);                          /* 5: the candidates' arithmetic */
            if (ok && !sv[SV_ERR]) {
                const CommitArgs ca = { ncand, diag_cand, nA, nLc, ncA + ncB, slotw, 0, nUc_all, U_l, Lb_total, Lnz_, Lnl_, Unz_, Unl_, ppcl };
                search_publish(ca);
            } else {
                if (lane == 0) { sv[SV_EST] = ok ? SLIPDEV_INTERNAL : -1; if (ok && !st->dbg_who) { st->dbg_who = 110; st->dbg_k = k; st->dbg_a = sv[SV_ERR]; } }      /* the bounds said this could not happen / the full pass */
            }
            if (lane == 0 && ok && !sv[SV_EST]) slip_agent_add_u64(&st->c_short, 1ull);
#ifdef SLIP_PROFILING
            if (lane == 0 && ok) { prof_[23] += 1; prof_[15] += (unsigned long long)(ncA + ncB); }
#endif
        } else {
            /* the position snapshot (pinv as the reference has it at column k) by the other waves, before the swap */
            for (int t = tid - SLIP_WAVE; t < nrows; t += T - SLIP_WAVE) {
                const int r = (int) f_row[t];
                const int pos = slip_ld_i32(P.pinv.at(r));
                slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
                f_pos[t] = (uint32_t) pos;
            }
            slip_block_sync_named(1);
            /* ... then their share of the candidates */
            const int ncA = sv[SV_CNT0 + 1], ncB = sv[SV_CNT0 + 2];
            if (sv[SV_LISTN]) {
                const int slotw = (lm + 5) & ~1;
                const int nstage = (3 * SLIP_PAT_CAP) / slotw < 30 ? (3 * SLIP_PAT_CAP) / slotw : 30;
                const uint32_t *wlB = work, *wlA = work + 2 * SLIP_CAND_CAP;
                if (ncA > wave) {
                    const SlipCandOut co = { f_k0, f_k1, f_inf, stage, slotw, nstage, kind };
                    const int e = slip_mul_rows_any(P, M, BMs ? Ms : slip_piv_digits(P, M), BMs ? 0 : 1, wlA, wave, nw, ncA, sv64[SV_LNL / 2], (uint32_t *) 0, (uint32_t *) 0, tag, &co);
                    if (e && lane == 0) sv[SV_ERR] = 1;
                    slip_vm_drain();
                }
                for (int t = wave; t < ncB; t += nw) {
                    const int e = slip_run_item_out(&P, 2, 0, 0, k, 0, wlB[t], 0u, b0, b1, b2);
                    if (e && lane == 0) sv[SV_ERR] = e;
                }
                slip_wave_sync();
            }
            slip_block_sync_named(2);
        }
        slip_block_sync();                       /* the join: wave 0 has published (or given the column to the full pass) */
        ec = sv[SV_EST];
        pc_col = sv[SV_TMP3];
        if (ec > 0) return ec;
        if (ec == 0) { early = 1; e_pivrow = sv[SV_EPR]; e_pivpos = sv[SV_EPP]; }
        SLIP_STAMP(6);
    }
    if (ec < 0) {
    /* class of a row for the early commit: 0 not a candidate (pivotal or zero), 1 exact (value at level k-1 in its x
     * row), 2 pending A (one limb times the long pivot -> straight into the L slab), 3 pending B (wave item) */
    uint64_t ulimbs = 0, lbound = 0; int nUc = 0, bad = 0, maxub = 0, maxlb = 0; uint32_t best_b = kind == 1 ? 0u : 0xFFFFFFFFu;
    volatile int32_t *acnt = &sv[SV_ACNT];               /* zero since the column started: no barrier needed before the first slot is drawn */
    for (int t0 = 0; t0 < nrows; t0 += T) {
        const int t = t0 + tid;
        int cls = 0, ub = 0, isA = 0, r = 0, pos = 0; uint32_t asgn = 0;
        if (t < nrows) {
            if (t0 == 0 && !fastc) { r = r0_; pos = pos0_; }  /* loaded above, with everything else */
            else {
                r = small ? (int) f_row[t] : P.rlist[t];      /* short patterns: listed in LDS since their discovery */
                pos = slip_ld_i32(P.pinv.at(r));
            }
            slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
            if (small) f_pos[t] = (uint32_t) pos; else P.rpos[t] = pos;
        }
        if (try_early && t < nrows) {
            const SlipRow xr = (t0 == 0 && !fastc) ? xr0_ : P.xrow[r];
            if (pos < k) { ulimbs += (uint64_t) slip_limbs(xr.len); nUc++; if (xr.bits > maxub) maxub = xr.bits; if (xr.bits > maxlb) maxlb = xr.bits; }
            else if (xr.len == 0) cls = 0;
            else if (xr.h >= k - 1) {
                cls = 1; ub = xr.bits; lbound += (uint64_t) slip_limbs(xr.len);
                f_k1[t] = 0xFFFFFFFFu; f_k0[t] = 0xFFFFFFFFu;        /* key not known yet: formed from the digits if it becomes a candidate */
            } else {
                slip_u128 y = 0; int ys = 1;
                uint64_t xv = 0;
                if (slip_abs(xr.len) <= 2) xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                if (slip_abs(xr.len) <= 2 && slip_history_small(P, xr, xv, M, xr.h, &y, &ys)) {
                    slip_store_small(P, r, y, ys, k - 1, tag);        /* now at level k-1 */
                    cls = 1; ub = slip_bits128(y); lbound += (uint64_t)((((ub + 31) >> 5) + 1) >> 1);
                    /* the lane has the value: search key from registers */
                    const uint64_t top = ub ? (uint64_t)((y << (128 - ub)) >> 64) : 0ull;
                    uint64_t key = ((uint64_t) ub << 40) | (top >> 24);
                    if (kind == 1) key = ~key;
                    f_k0[t] = (uint32_t) key; f_k1[t] = (uint32_t)(key >> 32);
                } else if (xr.h < 0 && slip_abs(xr.len) <= 2 && lm + 2 <= P.xcap && lm + 2 <= 256) {
                    cls = 2; isA = 1; ub = xr.bits + brho;
                    f_k0[t] = (uint32_t) xv; f_k1[t] = (uint32_t)(xv >> 32);       /* the one-limb value, for the candidate record (the key comes later) */
                    asgn = ((uint32_t) slip_abs(xr.len) << 12) | (xr.len < 0 ? 1u << 14 : 0u);
                } else {
                    cls = 3;
                    int bh = 0, zh = 0;
                    if (xr.h >= 0) { const SlipPiv H = slip_ld_piv(P.piv.at(xr.h)); bh = H.bits; zh = H.ctz; }
                    ub = xr.bits + brho - bh + (xr.h >= 0 ? 1 : 0);
                    if (ub < 1) ub = 1;
                    /* the item's working width (slip_history_wave) must fit the scratch, the x stride and the inverse cache */
                    const int Wn = ((ub + 31) >> 5) + ((zh + 31) >> 5) + 1;
                    if (Wn > P.wcap || Wn > P.xcap || Wn > P.invcap || lm > P.wcap) bad = 1;
                    lbound += (uint64_t)((ub + 63) >> 6);
                }
            }
            if (ub > maxub) maxub = ub;
            /* what the value certainly reaches (the window cap is decided on it when it is already beyond the cap) */
            { const int lb_ = cls == 1 ? ub : (cls == 2 ? ub - 1 : (cls == 3 ? (ub > 2 ? ub - 2 : 1) : 0)); if (lb_ > maxlb) maxlb = lb_; }
        }
        if (try_early) {
            /* class A rows get their slots in the L slab now (slot index kept in the row's own x area, behind the value) */
            const uint64_t am = slip_ballot(isA);
            int abase = 0;
            if (lane == 0 && am) abase = slip_atomic_add_i32((int32_t *) acnt, slip_popc64(am));
            abase = (int) slip_bcast0_u32((uint32_t) abase);
            if (isA) {
                const int si = abase + slip_popc64(am & ((1ull << lane) - 1ull));
                f_aux[t] = (uint32_t) si | asgn;                  /* slot (10 bits), digits of the value (12-13), its sign (14) */
                (P.xd + (int64_t) r * P.xcap)[2] = (uint32_t) si;
            }
            if (t < nrows) {
                f_inf[t] = (uint32_t) cls | ((uint32_t) ub << 2);
                if (cls) {
                    const int lb = cls == 1 ? ub : (cls == 2 ? ub - 1 : (ub > 2 ? ub - 2 : 1));
                    if (kind == 0) { if ((uint32_t) ub < best_b) best_b = (uint32_t) ub; }
                    else if (kind == 1) { if ((uint32_t) lb > best_b) best_b = (uint32_t) lb; }
                    else { if ((uint32_t) pos < best_b) best_b = (uint32_t) pos; }     /* first nonzero: by position */
                }
            }
        }
    }
    SLIP_TR(1);                                              /* 1: loads + classification */
    if (try_early && BMs) { const dig_t *Mg = slip_piv_digits(P, M); for (int c = tid; c < lm; c += T) Ms[c] = slip_ld_u32(Mg + c); }
    slip_block_sync();
    SLIP_TR(2);                                              /* 2: rho staging + barrier */
    pc_col = sv[SV_TMP3];
#ifdef SLIP_PROFILING
    tr_t2_ = slip_clock();
    tr_sweep_ = t_last_ ? tr_t2_ - t_last_ : 0;              /* sweep tail + position snapshot */
#endif
    SLIP_STAMP(2);

    /* ---- early commit: choose and publish the pivot BEFORE the column's bulk arithmetic.
     * The final values are x * rho[k-1] / rho[h]; their bit lengths are known to within two bits from the operands'
     * bit lengths, so only the rows whose bounds reach the best bound can be the pivot: those are brought to level
     * k-1 now (one-limb rows were, in the lane, above), the exact search runs among them, the pivot is published
     * (stage 1) and the frontier moves; every other row is multiplied / divided afterwards, off the commit chain.
     * The column must be certain to complete: capacities, widths and the column-window cap are checked on the bounds
     * first, otherwise the column takes the complete path below (everything computed, then the search, then the commit). */
    /* returns 0: committed (outcome in sv[SV_EPR], sv[SV_EPP]); -1: take the complete path; > 0: a status.  Run by every
     * thread, or (single-wave chain) by wave 0 alone */
    auto early_commit = [&]() -> int {
        /* one fused reduction: sums (U limbs, L limb bound, pivotal rows), maxima (trouble flag, longest bound), best bound.
         * Inside a wave with DPP row shifts (limb counts of a column stay far below 2^32), across the waves through LDS. */
        {
            const uint32_t w_u = slip_wave_sum_u32((uint32_t) ulimbs), w_l = slip_wave_sum_u32((uint32_t) lbound);
            const uint32_t w_n = slip_wave_sum_u32((uint32_t) nUc), w_bad = slip_wave_max_u32((uint32_t) bad);
            const uint32_t w_mx = slip_wave_max_u32((uint32_t) maxub), w_lb = slip_wave_max_u32((uint32_t) maxlb);
            const uint32_t w_b = slip_wave_min_u32(kind == 1 ? 0xFFFFFFFFu - best_b : best_b);     /* smaller is better in every kind */
            if (lane == 0) {
                uint32_t *rp = (uint32_t *) scan_tmp + 8 * wave;
                rp[0] = w_u; rp[1] = w_l; rp[2] = w_n; rp[3] = w_bad; rp[4] = w_mx; rp[5] = w_b; rp[6] = w_lb;
            }
        }
        slip_block_sync();
        uint64_t U_l = 0, L_b = 0; uint32_t nUc_all = 0, bad_all = 0, maxub_all = 0, maxlb_all = 0, bb = 0xFFFFFFFFu;
        for (int w = 0; w < nw; w++) {
            const uint32_t *rp = (const uint32_t *) scan_tmp + 8 * w;
            U_l += rp[0]; L_b += rp[1]; nUc_all += rp[2];
            if (rp[3] > bad_all) bad_all = rp[3];
            if (rp[4] > maxub_all) maxub_all = rp[4];
            if (rp[5] < bb) bb = rp[5];
            if (rp[6] > maxlb_all) maxlb_all = rp[6];
        }
        SLIP_STAMP(21);                                     /* early: classification + reduction */
        SLIP_TR(3);                                         /* 3: wave reduce + barrier + combine */
        if (bb == 0xFFFFFFFFu) return SLIPDEV_SINGULAR;      /* no nonzero non-pivotal row at all (slip_get_smallest_pivot.c:93-96) */
        /* column-window mode: a value whose LOWER bound is beyond the cap ends the window here, without the arithmetic of
         * the complete path (the last column of a window used to cost a millisecond to say so) */
        if (P.limb_cap > 0 && (int)((maxlb_all + 63) >> 6) > P.limb_cap) return SLIPDEV_WINDOW_END;
        const int nA = *acnt;
        const uint32_t bestb = kind == 1 ? 0xFFFFFFFFu - bb : bb;
        const int nLc = nrows - (int) nUc_all;
        const int64_t Lnz_ = sv64[SV_LNZ / 2], Lnl_ = sv64[SV_LNL / 2], Unz_ = sv64[SV_UNZ / 2], Unl_ = sv64[SV_UNL / 2];
        /* a pivot living in its x row gets a slot of its own behind the class-A slots: bound it by the longest value */
        const uint64_t preserve = (uint64_t)((maxub_all + 63) >> 6) + 1;
        const uint64_t Lb_total = (uint64_t) nA * (uint64_t) slot + preserve + L_b;
        const uint64_t Ub_total = U_l + preserve;
        int ok = bad_all == 0;
        if (Lnz_ + nLc > P.Lcap_nz || Lnl_ + (int64_t) Lb_total > P.Lcap_nl) ok = 0;
        if (Unz_ + (int) nUc_all + 1 > P.Ucap_nz || Unl_ + (int64_t) Ub_total > P.Ucap_nl) ok = 0;
        if (P.limb_cap > 0 && (int)((maxub_all + 63) >> 6) > P.limb_cap) ok = 0;  /* the window may end here: decide on exact values */
#ifdef SLIP_PROFILING
        if (tid == 0) P.dbg[17 * (int64_t) P.n + k] |= 0x4000 | (ok ? 0x8000 : 0) | (bad_all ? 0x400 : 0)
            | ((Lnz_ + nLc > P.Lcap_nz || Lnl_ + (int64_t) Lb_total > P.Lcap_nl) ? 0x800 : 0)
            | ((Unz_ + (int) nUc_all + 1 > P.Ucap_nz || Unl_ + (int64_t) Ub_total > P.Ucap_nl) ? 0x1000 : 0)
            | ((P.limb_cap > 0 && (int)((maxub_all + 63) >> 6) > P.limb_cap) ? 0x2000 : 0);
#endif
        const int diag_cand = (scheme == 1 || scheme == 3 || scheme == 4) && pc_col >= k && P.xrow[col].tag == tag && P.xrow[col].len != 0;
        /* lists in the work area: class-B candidates (rows), all candidates (table indices), class-A candidates (5-word records) */
        uint32_t *wlB = work, *cl = work + SLIP_CAND_CAP, *wlA = work + 2 * SLIP_CAND_CAP;
        volatile int32_t *cntA = &sv[SV_CNT0 + 1], *cntB = &sv[SV_CNT0 + 2], *cntC = &sv[SV_LISTN];
        const int slotw = (lm + 5) & ~1;                      /* even: the low limb is read as one aligned 64-bit word */
        const int nstage = (3 * SLIP_PAT_CAP) / slotw < 30 ? (3 * SLIP_PAT_CAP) / slotw : 30;   /* the slot number travels in 5 bits of f_inf */
        if (ok) {
            /* marking: the candidates.  Pending ones are listed for the waves: class A -> 5-word records (one limb times
             * rho[k-1] straight into its slot of the L slab), class B -> history items. */
            for (int t0 = 0; t0 < nrows; t0 += T) {
                const int t = t0 + tid;
                int wantA = 0, wantB = 0, cand = 0, r = 0;
                if (t < nrows) {
                    const uint32_t inf = f_inf[t];
                    const int cls = (int)(inf & 3u), ub = (int)(inf >> 2);
                    r = (int) f_row[t];
                    if (cls) {
                        const int lb = cls == 1 ? ub : (cls == 2 ? ub - 1 : (ub > 2 ? ub - 2 : 1));
                        if (kind == 0) cand = (uint32_t) lb <= bestb;
                        else if (kind == 1) cand = (uint32_t) ub >= bestb;
                        else cand = f_pos[t] == bestb;
                        if (diag_cand && r == col) cand = 1;
                    }
                    wantA = cand && cls == 2; wantB = cand && cls == 3;
                }
                const uint64_t mA = slip_ballot(wantA), mB = slip_ballot(wantB), mC = slip_ballot(cand);
                int bA = 0, bB = 0, bC = 0;
                if (lane == 0) {
                    if (mA) bA = slip_atomic_add_i32((int32_t *) cntA, slip_popc64(mA));
                    if (mB) bB = slip_atomic_add_i32((int32_t *) cntB, slip_popc64(mB));
                    if (mC) bC = slip_atomic_add_i32((int32_t *) cntC, slip_popc64(mC));
                }
                bA = (int) slip_bcast0_u32((uint32_t) bA); bB = (int) slip_bcast0_u32((uint32_t) bB); bC = (int) slip_bcast0_u32((uint32_t) bC);
                const uint64_t below = (1ull << lane) - 1ull;
                if (cand) { const int at = bC + slip_popc64(mC & below); if (at < SLIP_CAND_CAP) cl[at] = (uint32_t) t; }
                if (wantA) {
                    const int at = bA + slip_popc64(mA & below);
                    if (at < SLIP_CAND_CAP) {
                        const uint32_t ax = f_aux[t];
                        wlA[5 * at] = (uint32_t) r; wlA[5 * at + 1] = f_k0[t]; wlA[5 * at + 2] = f_k1[t];
                        wlA[5 * at + 3] = ((uint32_t) t << 3) | ((ax >> 14) & 1u ? 4u : 0u) | ((ax >> 12) & 3u);
                        wlA[5 * at + 4] = (ax & 0x3FFu) * (uint32_t) slot;
                    }
                } else if (wantB) { const int at = bB + slip_popc64(mB & below); if (at < SLIP_CAND_CAP) wlB[at] = (uint32_t) r; }
            }
            slip_block_sync();
            if (*cntC > SLIP_CAND_CAP) ok = 0;                /* too many candidates for the lists: the complete path */
            SLIP_TR(4);                                       /* 4: marking + barrier */
        }
        if (ok) {
            const int ncA = *cntA, ncB = *cntB, ncand = *cntC;
            if (ncA > 0) {
                const SlipCandOut co = { f_k0, f_k1, f_inf, stage, slotw, nstage, kind };
                const int e = slip_mul_rows_any(P, M, BMs ? Ms : slip_piv_digits(P, M), BMs ? 0 : 1, wlA, wave, nw, ncA, Lnl_, (uint32_t *) 0, (uint32_t *) 0, tag, &co);
                if (e && lane == 0) sv[SV_ERR] = 1;
                /* a product that is read back from the slab (not staged, or the diagonal rule looks at it) must have landed */
                if (ncA > nstage || diag_cand) slip_vm_drain();
            }
            slip_block_sync();
            if (ncB > 0) slip_drain(P, lds, 2, 0, 0, k, 0, ncB, wlB, b0, b1, b2);
            if (sv[SV_ERR]) { SLIP_SITE(1, sv[SV_ERR], 0); return SLIPDEV_INTERNAL; }          /* the bounds said this could not happen */
            SLIP_STAMP(22);                                   /* early: candidate lists and arithmetic */
            SLIP_TR(5);                                       /* 5: candidate multiplies + barrier */
            if (wave == 0) {
                const CommitArgs ca = { ncand, diag_cand, nA, nLc, ncA + ncB, slotw, 0, nUc_all, U_l, Lb_total, Lnz_, Lnl_, Unz_, Unl_, cl };
                search_publish(ca);
            }
#ifdef SLIP_PROFILING
            if (tid == 0) { prof_[23] += 1; prof_[15] += (unsigned long long)(ncA + ncB); }   /* early commits; their candidates that needed arithmetic */
#endif
            slip_block_sync();
            return sv[SV_EST];
        }
        return -1;
    };
    if (try_early) {
        ec = early_commit();
        if (ec > 0) return ec;
        if (ec == 0) { early = 1; e_pivrow = sv[SV_EPR]; e_pivpos = sv[SV_EPP]; }
        SLIP_STAMP(6);
    }
    }
    /* ---- phase 3c: reading the bitmap in order = the sorted pattern (slip_sort_xi.c); every row goes to the place of
     *      its snapshot position (binary search over the sorted positions) ---- */
    int npat_, nU_;
    slip_pattern(P, lds, bm, k, &npat_, &nU_);
    const int npat = npat_, nU = nU_, nL = npat - nU;
    const int32_t *patl = (const int32_t *)(lds + SLIP_LDS_PAT);
    auto pat_at = [&](int t) -> int { return npat <= SLIP_PAT_CAP ? patl[t] : P.pat[t]; };
    int32_t *rowl = (int32_t *)(lds + SLIP_LDS_ROWS);
    uint32_t *diroff = lds + SLIP_LDS_DIROFF;
    auto row_at = [&](int t) -> int { return npat <= SLIP_PAT_CAP ? rowl[t] : P.srow[t]; };
    slip_block_sync();
    if (npat != nrows) { SLIP_SITE(2, npat, nrows | (full_adopt << 16) | (adopted << 17) | (early << 18)); SLIP_WHY("col %d: npat %d != nrows %d (early %d full %d)\n", k, npat, nrows, early, full_adopt); return SLIPDEV_INTERNAL; }      /* every discovered row has exactly one position */
    for (int t = tid; t < nrows; t += T) {
        const int r = small ? (int) f_row[t] : P.rlist[t];
        const int pos = small ? (int) f_pos[t] : P.rpos[t];
        int lo = 0, hi = npat - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (pat_at(mid) < pos) lo = mid + 1; else hi = mid; }
        if (npat <= SLIP_PAT_CAP) rowl[lo] = r; else P.srow[lo] = r;
    }
    slip_block_sync();
    SLIP_STAMP(14);

    /* ---- phase 4: history update of the non-pivotal rows to level k-1 (:248-257) ---- */
    /* one-limb rows finished by a lane enter the column table (and the key list of the pivot search) from that lane's
     * registers (diroff[] = 0x7FFFFFFF); for rows multiplied straight into the L slab diroff[] holds the slab offset;
     * all-ones: neither */
    const int prefill_ok = k >= 1 && npat <= SLIP_PAT_CAP;
    uint32_t *ctab = lds + SLIP_LDS_TAB, *ckeys = lds + SLIP_LDS_KEYS;
    if (k >= 1) {
        volatile int32_t *wcnt = &sv[SV_CNT0], *wcnt2 = &sv[SV_CNT0 + 1];
        if (tid == 0) { *wcnt = 0; *wcnt2 = 0; }
        uint32_t *wl2 = work + SLIP_WORK_CAP;              /* 5-word records, SLIP_WORK_CAP of them */
        /* rho[k-1] is the multiplier of every row: its digits are staged once (LDS when it fits; the early pass did it) */
        const dig_t *Mg = slip_piv_digits(P, M);
        const dig_t *Md = Mg; int md_shared = 1;
        if (BMs) { if (!try_early) for (int c = tid; c < lm; c += T) Ms[c] = slip_ld_u32(Mg + c); Md = Ms; md_shared = 0; }
        slip_block_sync();
        SLIP_STAMP(8);
        uint32_t *wl = work;
        for (int t0 = 0; t0 < nL; t0 += SLIP_WORK_CAP) {
            const int te = t0 + SLIP_WORK_CAP < nL ? t0 + SLIP_WORK_CAP : nL;
            const unsigned long long chunk_base = (unsigned long long) sv64[SV_LALLOC / 2];
            for (int tb = t0; tb < te; tb += T) {
                /* every lane classifies its row; list slots are then handed out per WAVE (one LDS atomic per wave and list
                 * instead of one per row on the same counter) */
                const int t = tb + tid;
                int cls = 0, r = 0;                          /* 1: one limb times the long pivot, 2: wave item (division) */
                SlipRow xr; xr.len = 0; xr.h = 0; xr.bits = 0; xr.tag = 0;
                uint64_t xv = 0;
                if (t < te) {
                r = row_at(nU + t);
                xr = P.xrow[r];
                if (npat <= SLIP_PAT_CAP) diroff[nU + t] = 0xFFFFFFFFu;
                }
                if (t < te && !(xr.len == 0 || xr.h >= k - 1 || xr.h == -2)) {     /* zero, already at level k-1, or already in the slab */
                int done = 0;
                if (slip_abs(xr.len) <= 2) {
                    xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                    slip_u128 y = 0; int ys = 1;
                    if (slip_history_small(P, xr, xv, M, xr.h, &y, &ys)) {
                        slip_store_small(P, r, y, ys, k - 1, tag);
                        if (npat <= SLIP_PAT_CAP) {          /* the lane has the value: table entry and pivot key from registers */
                            const int yb = slip_bits128(y), yl = (yb + 31) >> 5, pidx = nU + t;
                            ctab[0 * SLIP_TAB_CAP + pidx] = (uint32_t) r; ctab[1 * SLIP_TAB_CAP + pidx] = (uint32_t)(ys < 0 ? -yl : yl);
                            ctab[2 * SLIP_TAB_CAP + pidx] = (uint32_t) yb; ctab[3 * SLIP_TAB_CAP + pidx] = 0u;
                            const uint64_t top = yb ? (uint64_t)((y << (128 - yb)) >> 64) : 0ull;
                            ckeys[2 * pidx] = (uint32_t) top; ckeys[2 * pidx + 1] = (uint32_t)(top >> 32);
                            diroff[pidx] = 0x7FFFFFFFu;      /* entered, value in its x row */
                        }
                        done = 1;
                    } else if (xr.h < 0 && lm + 2 <= P.xcap && lm + 2 <= 256) {
                        /* one limb times a long pivot, no division: wave path with the pivot in registers */
                        /* every such row gets a slot of (lm+3)/2 limbs in the L slab (the product has at most lm+2 digits) */
                        cls = 1;
                        done = 1;
                    } else if (xr.h < 0 && lm + 2 <= P.xcap) {
                        /* beyond 256 digits: this lane walks the pivot's digits */
                        dig_t *X = P.xd + (int64_t) r * P.xcap;
                        const uint64_t a0 = xv & 0xFFFFFFFFu, a1 = xv >> 32;
                        uint64_t carry = 0;              /* < 2^64 */
                        for (int c = 0; c < lm; c++) {
                            const uint64_t d = md_shared ? slip_ld_u32(Md + c) : Md[c];
                            const uint64_t lo = a0 * d + (carry & 0xFFFFFFFFu);          /* < 2^64 */
                            X[c] = (uint32_t) lo;
                            carry = a1 * d + (carry >> 32) + (lo >> 32);                 /* < 2^64 */
                        }
                        int len = lm;
                        if (carry) { X[len++] = (uint32_t) carry; if (carry >> 32) X[len++] = (uint32_t)(carry >> 32); }
                        if (len & 1) X[len] = 0;
                        SlipRow nr; nr.len = (slip_sgn(xr.len) * slip_sgn(M.len)) < 0 ? -len : len; nr.h = k - 1; nr.tag = tag;
                        nr.bits = 32 * len - slip_clz32(X[len - 1]);
                        P.xrow[r] = nr;
                        done = 1;
                    }
                }
                if (!done) cls = 2;
                }
                const uint64_t m1 = slip_ballot(cls == 1), m2 = slip_ballot(cls == 2);
                int base1 = 0, base2 = 0;
                if (lane == 0) {
                    if (m1) base1 = slip_atomic_add_i32((int32_t *) wcnt2, slip_popc64(m1));
                    if (m2) base2 = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(m2));
                }
                base1 = (int) slip_bcast0_u32((uint32_t) base1); base2 = (int) slip_bcast0_u32((uint32_t) base2);
                const uint64_t below = (1ull << lane) - 1ull;
                if (cls == 1) {
                    const int at = base1 + slip_popc64(m1 & below);
                    wl2[5 * at] = (uint32_t) r; wl2[5 * at + 1] = (uint32_t) xv; wl2[5 * at + 2] = (uint32_t)(xv >> 32);
                    wl2[5 * at + 3] = ((uint32_t)(nU + t) << 3) | (xr.len < 0 ? 4u : 0u) | (uint32_t) slip_abs(xr.len);
                    /* an early commit has handed out the slots already (index kept behind the row's value) */
                    wl2[5 * at + 4] = early ? (P.xd + (int64_t) r * P.xcap)[2] * (uint32_t) slot
                                            : (uint32_t)(chunk_base + (unsigned long long) at * slot);
                    if (npat <= SLIP_PAT_CAP) diroff[nU + t] = wl2[5 * at + 4];
                } else if (cls == 2) {
                    wl[base2 + slip_popc64(m2 & below)] = (uint32_t) r;
                }
            }
            slip_block_sync();
            SLIP_STAMP(9);
            /* the slots handed out above must exist before anything is written into them */
            const unsigned long long lalloc_now = early ? chunk_base : chunk_base + (unsigned long long) *wcnt2 * slot;
            if (!early && sv64[SV_LNL / 2] + (int64_t) lalloc_now > P.Lcap_nl) return SLIPDEV_GROW_L;
            const int nq = *wcnt;
            const int n2 = *wcnt2;
            const int64_t sb = sv64[SV_LNL / 2];
            if (n2 > 0) {
                const int e = slip_mul_rows_any(P, M, Md, md_shared, wl2, wave, nw, n2, sb, prefill_ok ? ctab : (uint32_t *) 0, ckeys, tag);
                if (e && lane == 0) sv[SV_ERR] = 1;
            }
            slip_block_sync();
            SLIP_STAMP(11);                                   /* the products of the class-A rows */
            slip_drain(P, lds, 2, 0, 0, k, 0, nq, wl, b0, b1, b2);
            SLIP_STAMP(10);
            if (tid == 0) { *wcnt = 0; *wcnt2 = 0; sv64[SV_LALLOC / 2] = (int64_t) lalloc_now; }
            slip_block_sync();
        }
    }
    if (sv[SV_ERR]) { if (early || sv[SV_ERR] >= 6) SLIP_SITE(3, sv[SV_ERR], early); SLIP_WHY("col %d: error %d in the history phase (early %d)\n", k, (int) sv[SV_ERR], early); return (early || sv[SV_ERR] >= 6) ? SLIPDEV_INTERNAL : SLIPDEV_GROW_X; }    /* after an early commit nothing may fail */
    SLIP_STAMP(3);

    /* ---- phase 5: column-window cap, then the pivot search ---- */
    /* column table in LDS (four arrays of SLIP_TAB_CAP words: row, signed length, bit length, flag/offset per pattern
     * entry -- struct-of-arrays so that consecutive lanes hit consecutive banks): read once, used by
     * the cap test, the pivot search, the offsets and the copy below */
    uint32_t *tab = lds + SLIP_LDS_TAB;
    const bool use_tab = npat <= SLIP_TAB_CAP;
    const int64_t Lnl0 = sv64[SV_LNL / 2];              /* L slab cursor at the start of this column */
    auto ent_row  = [&](int t) -> int { return use_tab ? (int) tab[0 * SLIP_TAB_CAP + t] : row_at(t); };
    auto ent_len  = [&](int t) -> int32_t { return use_tab ? (int32_t) tab[1 * SLIP_TAB_CAP + t] : P.xrow[row_at(t)].len; };
    auto ent_bits = [&](int t) -> int { return use_tab ? (int) tab[2 * SLIP_TAB_CAP + t] : P.xrow[row_at(t)].bits; };
    auto ent_direct = [&](int t) -> int { return use_tab ? (int)(tab[3 * SLIP_TAB_CAP + t] >> 31) : row_direct(row_at(t)); };
    auto ent_digits = [&](int t) -> const dig_t * {
        if (use_tab) return (tab[3 * SLIP_TAB_CAP + t] >> 31) ? (const dig_t *)(P.Llimbs + Lnl0 + (int64_t)(tab[3 * SLIP_TAB_CAP + t] & 0x7FFFFFFFu))
                                                  : P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + t] * P.xcap;
        return row_digits(row_at(t));
    };
    {
        int mx = 0;
        for (int t = tid; t < npat; t += T) {
            if (prefill_ok && t >= nU && diroff[t] == 0x7FFFFFFFu) {     /* entered by the lane that produced the value */
                const int l = slip_abs((int32_t) tab[1 * SLIP_TAB_CAP + t]);
                if (l > mx) mx = l;
                continue;
            }
            if (prefill_ok && t >= nU && diroff[t] != 0xFFFFFFFFu) {     /* entered by the wave that multiplied the row into the slab */
                const int l = slip_abs((int32_t) tab[1 * SLIP_TAB_CAP + t]);
                if (l > mx) mx = l;
                continue;
            }
            const int r = row_at(t);
            const SlipRow xr = P.xrow[r];
            const int l = slip_abs(xr.len);
            if (l > mx) mx = l;
            if (use_tab) {
                tab[0 * SLIP_TAB_CAP + t] = (uint32_t) r; tab[1 * SLIP_TAB_CAP + t] = (uint32_t) xr.len; tab[2 * SLIP_TAB_CAP + t] = (uint32_t) xr.bits;
                tab[3 * SLIP_TAB_CAP + t] = xr.h == -2 ? (0x80000000u | (uint32_t)(*(const int64_t *)(P.xd + (int64_t) r * P.xcap) - Lnl0)) : 0u;
            }
        }
        mx = slip_wave_max_i32(mx);
        if (mx > 0 && lane == 0) slip_atomic_max_i32((int32_t *) &sv[SV_MAXDIG], mx);
        slip_block_sync();
    }
    const int maxdig = sv[SV_MAXDIG];
    if (!early && P.limb_cap > 0 && ((maxdig + 1) >> 1) > P.limb_cap) return SLIPDEV_WINDOW_END;
    SLIP_STAMP(12);                                       /* column table built */

    /* kind of search: 0 smallest, 1 largest, 2 first nonzero (slip_get_pivot.c:58-155).
     * Lanes order the candidates by (bit length, leading bits); the candidates that tie on
     * that key are compared exactly, ties resolved towards the earlier pattern position. */
    int best = -1;
    if (early) {
        /* the pivot is committed: only its place in the pattern is needed */
        if (tid == 0) sv[SV_TMP] = -1;
        slip_block_sync();
        for (int t = tid; t < nL; t += T) if (ent_row(nU + t) == e_pivrow) sv[SV_TMP] = t;
        slip_block_sync();
        best = sv[SV_TMP];
        if (best < 0) { SLIP_SITE(4, e_pivrow, nL); SLIP_WHY("col %d: pivot row %d not in the L part\n", k, e_pivrow); return SLIPDEV_INTERNAL; }
    } else if (kind != 2 && maxdig < (1 << 18)) {
        /* one pass: (bit length, leading 40 bits) packed into one key; the candidates that share the best key are
         * compared exactly, ties towards the earlier pattern position (slip_get_smallest_pivot.c:79) */
        auto key_of = [&](int t) -> uint64_t {
            const int32_t xl = ent_len(nU + t);
            if (xl == 0) return ~0ull;
            const bool pre = prefill_ok && diroff[nU + t] != 0xFFFFFFFFu;
            const uint64_t top = pre ? ((uint64_t) ckeys[2 * (nU + t)] | ((uint64_t) ckeys[2 * (nU + t) + 1] << 32))
                                     : slip_top64(ent_digits(nU + t), slip_abs(xl), ent_direct(nU + t));
            const uint64_t v = ((uint64_t) ent_bits(nU + t) << 40) | (top >> 24);
            return kind == 0 ? v : ~v;
        };
        uint64_t mykey = ~0ull, k1 = ~0ull;
        for (int t = tid; t < nL; t += T) { const uint64_t c = key_of(t); if (t == tid) mykey = c; if (c < k1) k1 = c; }
        const uint64_t mk = slip_block_min_u64(k1, scan_tmp);
        if (mk == ~0ull) return SLIPDEV_SINGULAR;
        if (tid == 0) sv[SV_LISTN] = 0;
        slip_block_sync();
        for (int t = tid; t < nL; t += T) {
            const uint64_t c = t == tid ? mykey : key_of(t);
            if (c != mk) continue;
            const int at = slip_atomic_add_i32((int32_t *) &sv[SV_LISTN], 1);
            if (at < 2 * SLIP_WORK_CAP) work[at] = (uint32_t) t;
        }
        slip_block_sync();
        const int nc = sv[SV_LISTN];
        /* every wave performs the same reduction (wave-uniform, reads only) */
        const int listed = nc <= 2 * SLIP_WORK_CAP;
        const int kbits = (int)((kind == 0 ? mk : ~mk) >> 40);   /* at most 40 bits: equal keys are equal values */
        for (int c = 0; c < (listed ? nc : nL); c++) {
            const int t = listed ? (int) work[c] : c;
            if (!listed && key_of(t) != mk) continue;
            if (best < 0) { best = t; continue; }
            int cmp = 0;
            if (kbits > 40)      /* equal keys have equal bit lengths, hence equal digit counts */
                cmp = slip_cmp_mag(ent_digits(nU + best), ent_direct(nU + best), ent_digits(nU + t), ent_direct(nU + t), slip_abs(ent_len(nU + t)));
            if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0) || (cmp == 0 && t < best)) best = t;
        }
        slip_block_sync();
    } else {
        uint64_t k1 = ~0ull;                                     /* (key, t) packed: smaller is better */
        for (int t = tid; t < nL; t += T) {
            if (ent_len(nU + t) == 0) continue;
            const int bits = ent_bits(nU + t);
            const uint64_t key = kind == 2 ? 0 : (kind == 0 ? (uint64_t) bits : (uint64_t)(0x7FFFFFFF - bits));
            const uint64_t c = (key << 32) | (uint32_t) t;
            if (c < k1) k1 = c;
        }
        k1 = slip_block_min_u64(k1, scan_tmp);
        if (k1 == ~0ull) return SLIPDEV_SINGULAR;
        if (kind == 2) best = (int)(k1 & 0xFFFFFFFFu);
        else {
            const uint32_t bkey = (uint32_t)(k1 >> 32);
            const int bbits = kind == 0 ? (int) bkey : 0x7FFFFFFF - (int) bkey;
            /* exact comparison among the candidates of the winning bit-length class, in pattern order */
            for (int t = 0; t < nL; t++) {
                const int32_t xl = ent_len(nU + t);
                if (xl == 0 || ent_bits(nU + t) != bbits) continue;
                if (best < 0) { best = t; continue; }
                const int cmp = slip_cmp_mag(ent_digits(nU + best), ent_direct(nU + best), ent_digits(nU + t), ent_direct(nU + t), slip_abs(xl));
                if ((kind == 0 && cmp > 0) || (kind == 1 && cmp < 0)) best = t;
            }
            slip_block_sync();
        }
    }
    SLIP_STAMP(13);                                       /* smallest / largest candidate known */
    int pivrow = ent_row(nU + best);
    /* the diagonal preference (slip_get_pivot.c:68-76, 89-118, 126-146); an early commit has applied it already */
    if (!early && (scheme == 1 || scheme == 3 || scheme == 4)) {
        const int pc = pc_col;
        const int diag_ok = pc >= k && ((bm[pc >> 5] >> (pc & 31)) & 1u) && P.xrow[col].tag == tag && P.xrow[col].len != 0;
        if (diag_ok && pivrow != col) {
            int err = 0;
            const int take = diag_rule(pivrow, &err);
            if (err) return SLIPDEV_GROW_X;
            if (take) pivrow = col;
        }
    }
    const int pivpos = early ? e_pivpos : (pivrow == col ? pc_col : pat_at(nU + best));   /* pre-swap position (the pattern holds positions), >= k */
    SLIP_STAMP(4);

    /* ---- phase 6: append U(:,k) and L(:,k) (SLIP_LU_factorize.c:226-263) ---- */
    /* U(:,k): pattern rows below k in order, then the pivot.  L(:,k): rows at or above k in order. */
    const int nUe = nU + 1, nE = nUe + nL;
    const int64_t Lnz = sv64[SV_LNZ / 2], Lnl = sv64[SV_LNL / 2], Unz = sv64[SV_UNZ / 2], Unl = sv64[SV_UNL / 2];
    uint64_t baseU = 0, baseL = 0;
    /* pattern index of output entry e: U part, then the pivot (L position `pividx`), then the L part */
    int pividx = nU + best;
    if (pivrow != ent_row(nU + best)) {               /* the diagonal was preferred: find it in the L part */
        if (tid == 0) sv[SV_TMP] = -1;
        slip_block_sync();
        for (int t = tid; t < nL; t += T) if (ent_row(nU + t) == pivrow) sv[SV_TMP] = nU + t;
        slip_block_sync();
        pividx = sv[SV_TMP];
    }
    /* rows multiplied straight into the L slab (phase 4) already own the first `lalloc` limbs behind Lnl;
     * the other L rows are copied behind them */
    const uint64_t lalloc = (uint64_t) sv64[SV_LALLOC / 2];
    uint64_t dirL = 0;                                   /* exact limbs of the direct rows (channel packed with U) */
    for (int e0 = 0; e0 < nE; e0 += T) {
        const int e = e0 + tid;
        int r = -1, pt = 0, direct = 0; uint64_t lu = 0, ll = 0; int32_t xl = 0; int xb = 0; int64_t doff = 0;
        if (e < nE) {
            pt = e < nU ? e : (e == nU ? pividx : e - 1);
            r = ent_row(pt); xl = ent_len(pt); xb = ent_bits(pt);
            if (e < nUe) lu = (uint64_t) slip_limbs(xl);
            else {
                if (use_tab) { direct = (int)(tab[3 * SLIP_TAB_CAP + pt] >> 31); doff = Lnl + (int64_t)(tab[3 * SLIP_TAB_CAP + pt] & 0x7FFFFFFFu); }
                else if (P.xrow[r].h == -2) { direct = 1; doff = *(const int64_t *)(P.xd + (int64_t) r * P.xcap); }
                if (!direct) ll = (uint64_t) slip_limbs(xl);
                else lu = (uint64_t) slip_limbs(xl) << 32;             /* summed in the high half of the U channel */
            }
        }
        uint64_t eu, el, tu, tl;
        slip_block_scan2(lu, ll, scan_tmp, &eu, &el, &tu, &tl);
        if (e < nE) {
            /* capacity is verified before anything is committed; these records are provisional (nobody reads beyond the
             * published column pointers) */
            if (e < nUe) {
                const int64_t at = Unz + e;
                if (at < P.Ucap_nz) { P.Ui[at] = r; SlipEnt en; en.off = Unl + (int64_t)(baseU + (eu & 0xFFFFFFFFull)); en.len = xl; en.bits = xb; P.Ue[at] = en; }
                if (use_tab) work[e] = (uint32_t)(baseU + (eu & 0xFFFFFFFFull));     /* copy destination inside the U slab */
            } else {
                const int64_t at = Lnz + (e - nUe);
                const int64_t off = direct ? doff : Lnl + (int64_t)(lalloc + baseL + el);
#ifdef SLIP_BULK_SC1
                if (at < P.Lcap_nz) { slip_st_i32(&P.Li[at], r); SlipEnt en; en.len = xl; en.bits = xb; en.off = off; slip_st_ent(&P.Le[at], en); }
#else
                if (at < P.Lcap_nz) { P.Li[at] = r; SlipEnt en; en.len = xl; en.bits = xb; en.off = off; P.Le[at] = en; }   /* plain: published by the release before Lready[k] */
#endif
                if (use_tab && !direct) tab[3 * SLIP_TAB_CAP + pt] = (uint32_t)(off - Lnl);     /* copy destination, flag bit clear */
            }
        }
        baseU += tu & 0xFFFFFFFFull; dirL += tu >> 32; baseL += tl;
    }
    const uint64_t totU = baseU, totL = lalloc + baseL;         /* limbs of slab consumed by this column */
    const uint64_t totLexact = baseL + dirL;
    if (Unz + nUe > P.Ucap_nz || Unl + (int64_t) totU > P.Ucap_nl) { if (early) SLIP_SITE(5, nUe, totU); return early ? SLIPDEV_INTERNAL : SLIPDEV_GROW_U; }
    if (Lnz + nL > P.Lcap_nz || Lnl + (int64_t) totL > P.Lcap_nl) { if (early) SLIP_SITE(6, nL, totL); return early ? SLIPDEV_INTERNAL : SLIPDEV_GROW_L; }
    if (early && (Lnl + (int64_t) totL > slip_ld_i64(&P.Lo[k + 1]) || Unl + (int64_t) totU != slip_ld_i64(&P.Uo[k + 1]))) { SLIP_SITE(7 + full_adopt, (Lnl + (int64_t) totL) - slip_ld_i64(&P.Lo[k + 1]), (Unl + (int64_t) totU) - slip_ld_i64(&P.Uo[k + 1])); SLIP_WHY("col %d: bounds: L %lld + %lld vs %lld, U %lld + %lld vs %lld (full %d)\n", k, (long long) Lnl, (long long) totL, (long long) slip_ld_i64(&P.Lo[k + 1]), (long long) Unl, (long long) totU, (long long) slip_ld_i64(&P.Uo[k + 1]), full_adopt); return SLIPDEV_INTERNAL; }   /* the published bounds hold */
    /* the pivot's record fields, read before the permutation swap below changes what row_perm answers */
    const int32_t plen = ent_len(pividx);
    const int pbits = ent_bits(pividx);
    slip_vm_drain();                                     /* the direct rows' and the records' write-through stores have left */
    slip_block_sync();
    SLIP_STAMP(5);

    /* ---- stage 1: publish the pivot.  Wave 0 moves the pivot's digits to their place in the L slab (if they are
     *      not there yet), builds the pivot record, swaps the permutation and publishes the column pointers; then
     *      the frontier moves and the next column may commit while this one still writes its bulk. ---- */
    if (wave == 0 && !early) {
        const int found = pividx - nU;                   /* position of the pivot inside L(:,k) */
        const int lp_ = slip_abs(plen);
        int pdirect; int64_t poff;
        if (use_tab) { pdirect = (int)(tab[3 * SLIP_TAB_CAP + pividx] >> 31); poff = Lnl + (int64_t)(tab[3 * SLIP_TAB_CAP + pividx] & 0x7FFFFFFFu); }
        else { const SlipEnt pe = slip_ld_ent(&P.Le[Lnz + found]); pdirect = row_direct(pivrow); poff = pe.off; }
        dig_t *dst = (dig_t *)(P.Llimbs + poff);
        const dig_t *src = pdirect ? (const dig_t *) dst : P.xd + (int64_t) pivrow * P.xcap;
        const int z = slip_publish_digits(dst, src, pdirect, lp_);
        const uint64_t lo64 = pdirect ? slip_ld_u64((const uint64_t *) dst) : *(const uint64_t *) src;
        slip_vm_drain();
        if (lane == 0) {
            SlipPiv pr; pr.off = poff; pr.len = plen; pr.bits = pbits; pr.ctz = z; pr.invlen = 0;
            pr.lo = lo64; pr.inv64 = 0; pr.pad = 0;
            if (lp_ <= 2) pr.inv64 = slip_inv64(pr.lo >> z);
            slip_st_piv(P.piv.at(k), pr);
            const int intermed = pivpos, intermed2 = slip_ld_i32(P.row_perm.at(k));
            slip_st_i32(P.row_perm.at(k), pivrow); slip_st_i32(P.row_perm.at(intermed), intermed2);
            slip_st_i32(P.pinv.at(pivrow), k); slip_st_i32(P.pinv.at(intermed2), intermed);
            slip_st_i32(P.sw_row.at(k), intermed2); slip_st_i32(P.sw_pos.at(k), intermed);
            slip_st_i64(&P.Up[k + 1], Unz + nUe); slip_st_i64(&P.Lp[k + 1], Lnz + nL);
            slip_st_i64(&P.Uo[k + 1], Unl + (int64_t) totU); slip_st_i64(&P.Lo[k + 1], Lnl + (int64_t) totL);
            slip_vm_drain();
            slip_st_frontier(st, k + 1, pivrow);
#ifdef SLIP_PROFILING
            P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 2] = (int32_t) slip_realtime();              /* time line 2: committed (by its worker) */
            /* slot 18: from the moment this worker learnt that column k-1 was committed to its own commit */
            if (t_last_) prof_[18] += slip_clock() - t_last_;
            {
                int32_t *tr = P.dbg + 8 * (int64_t) k; const unsigned long long nowc = slip_clock();
                tr[0] = t_last_ ? (int32_t)(nowc - t_last_) : -1; tr[1] = 0; tr[2] = 0; tr[3] = nrows;
                tr[4] = (int32_t) tr_sweep_; tr[5] = (int32_t)(nowc - tr_t2_); tr[7] = P.worker;
                        tr[6] = (int32_t) slip_realtime(); P.dbg[8 * (int64_t) P.n + k] = (int32_t) t_wait_[2];
            }
#endif
        }
    }
    SLIP_STAMP(6);

    /* ---- stage 2: the limbs.  One wave per entry, coalesced; x rows are stored padded to whole limbs.  Rows that
     *      already live in the L slab are not copied.  L is shared (write-through), U is only read by later launches. ---- */
    /* the lanes of a wave look at its entries side by side and only the entries that really need a copy (U entries,
     * L rows that are not already in the slab) are then walked one after the other */
    for (int base = wave; base < nE; base += nw * SLIP_WAVE) {
    uint64_t todo_e;
    {
        const int el = base + nw * lane;
        int need = 0;
        if (el < nE) need = (use_tab && el >= nUe) ? !(tab[3 * SLIP_TAB_CAP + (el - 1)] >> 31) : 1;
        todo_e = slip_ballot(need);
    }
    while (todo_e) {
        const int e = base + nw * slip_ctz64(todo_e);
        todo_e &= todo_e - 1;
        const int isU = e < nUe;
        const dig_t *srcx; dig_t *dst; int32_t xl; int src_shared = 0;
        if (use_tab && isU) {
            /* U(:,k): pivotal rows live in x; the pivot (last) may have been multiplied straight into the L slab */
            const int pt = e < nU ? e : pividx;
            xl = (int32_t) tab[1 * SLIP_TAB_CAP + pt];
            src_shared = (int)(tab[3 * SLIP_TAB_CAP + pt] >> 31);
            /* (the pivot is read where the column produced it, not from the L copy wave 0 may still be writing) */
            srcx = src_shared ? (const dig_t *)(P.Llimbs + Lnl + (int64_t)(tab[3 * SLIP_TAB_CAP + pt] & 0x7FFFFFFFu))
                              : P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + pt] * P.xcap;
            dst = (dig_t *)(P.Ulimbs + Unl + (int64_t) work[e]);
        } else if (use_tab) {
            const int pt = e - 1;
            if (tab[3 * SLIP_TAB_CAP + pt] >> 31) continue;                 /* multiplied straight into the slab */
            if (pt == pividx) continue;                                    /* copied at stage 1 */
            xl = (int32_t) tab[1 * SLIP_TAB_CAP + pt];
            srcx = P.xd + (int64_t) tab[0 * SLIP_TAB_CAP + pt] * P.xcap;
            dst = (dig_t *)(P.Llimbs + Lnl + (int64_t) tab[3 * SLIP_TAB_CAP + pt]);
        } else {
            const int64_t at = isU ? Unz + e : Lnz + (e - nUe);
            const int r = isU ? P.Ui[at] : slip_ld_i32(&P.Li[at]);                  /* this worker's own records: sc1 loads see its L2 */
            const SlipEnt en = isU ? P.Ue[at] : slip_ld_ent(&P.Le[at]);
            xl = en.len;
            src_shared = row_direct(r);
            srcx = row_digits(r);
            dst = isU ? (dig_t *)(P.Ulimbs + en.off) : (dig_t *)(P.Llimbs + en.off);
            if (!isU && (src_shared || r == pivrow)) continue;              /* in the slab already / copied at stage 1 */
        }
        const int lw = (slip_abs(xl) + 1) & ~1;
        if (isU) { for (int c = lane; c < lw; c += SLIP_WAVE) dst[c] = src_shared ? slip_ld_u32(srcx + c) : srcx[c]; }
#ifdef SLIP_BULK_SC1
        else     { for (int c = lane; c < lw; c += SLIP_WAVE) slip_st_u32(dst + c, srcx[c]); }
#else
        else     { for (int c = lane; c < lw; c += SLIP_WAVE) dst[c] = srcx[c]; }
#endif
    }
    }
    slip_vm_drain();
    slip_block_sync();
    if (tid == 0) {
        slip_agent_release();                            /* the column's plain stores (L entries, limbs) leave this XCD's L2 */
        slip_agent_add_i32(P.Lready.at(k), 1);             /* returning atomic: performed before the advance below reads the flags */
        slip_advance_ready(P, st);
        slip_agent_add_u64(&st->c_write, 4ull * (unsigned long long) nE + 8ull * (totU + totLexact) + 8ull * slip_limbs(plen));
        slip_agent_add_u64((unsigned long long *) &st->Lnl_exact, totLexact);
        slip_agent_add_u64((unsigned long long *) &st->Unl_exact, totU);
        slip_agent_max_u64(&st->c_maxdig, (unsigned long long) maxdig);
    }
    /* this column's counters go to the launch totals now, wave by wave (they used to ride in ten registers per lane from
     * column to column, in a kernel whose register allocation is at the limit) */
    {
        auto wsum64 = [&](unsigned long long v) -> unsigned long long {
            const unsigned long long a = slip_wave_sum_u32((uint32_t)(v & 0xFFFFFFull)), b = slip_wave_sum_u32((uint32_t)((v >> 24) & 0xFFFFFFull));
            const unsigned long long c = slip_wave_sum_u32((uint32_t)(v >> 48));
            return a + (b << 24) + (c << 48);
        };
        const unsigned long long t_read = wsum64(c_read), t_upd = wsum64(c_upd), t_src = wsum64(c_src), t_str = wsum64(c_str), t_mac = wsum64(c_mac);
        if (lane == 0) {
            if (t_read) slip_agent_add_u64(&st->c_read, t_read);
            if (t_upd) slip_agent_add_u64(&st->c_upd, t_upd);
            if (t_src) slip_agent_add_u64(&st->c_src, t_src);
            if (t_str) slip_agent_add_u64(&st->c_streamed, t_str);
            if (t_mac) slip_agent_add_u64(&st->c_macs, t_mac);
        }
    }
#ifdef SLIP_PROFILING
    if (tid == 0) P.dbg[18 * (int64_t) P.n + 6 * (int64_t) k + 4] = (int32_t) slip_realtime();          /* time line 4: the column ends */
    if (tid == 0) P.dbg[17 * (int64_t) P.n + k] |= (packaged ? 1 : 0) | (adopted ? 2 : 0) | (fastc ? 4 : 0) | (early ? 8 : 0) | (sv[SV_PKGVER] << 4) | (nrows << 16);
#endif
    SLIP_STAMP(7);
    SLIP_STAMP_FLUSH(st);
    slip_block_sync();
    return SLIPDEV_OK;
}

/* the kernel body of a column worker: draw columns from the ticket counter until none is left */
template <bool FAST>
SLIP_DEV void slip_factor_worker(const SlipParams &P, SlipState *st, uint32_t *lds)
{
    const int tid = slip_tid();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    if (tid == 0) { sv[SV_CUP] = 0; sv[SV_NOENG] = 0; }     /* the committer has not been seen yet */
    for (;;) {
        slip_block_sync();
        if (tid == 0) {
            const int t = slip_agent_add_i32(&st->ticket, 1);
            int k = P.k0 + (t - P.t0);
            /* a column beyond the one that stopped the factorisation can never commit */
            if (k < P.k_stop && (slip_ld_i64(&st->stop) >> 8) < (int64_t) k) k = P.k_stop;
            sv[SV_K] = k; sv[SV_TAG] = t + 1;
        }
        slip_block_sync();
        const int k = sv[SV_K], tag = sv[SV_TAG];
        if (k >= P.k_stop) break;
        const int status = slip_do_column<FAST>(P, st, k, tag, lds);
        if (status == SLIPDEV_OK) continue;
        if (status != SLIPDEV_ABORTED && tid == 0) slip_raise_stop(st, status == SLIPDEV_INTERNAL ? 0 : k, status);
        break;
    }
    slip_block_sync();
    (void) scan_tmp;
}

/* ------------------------------------------------------------------ */
/* REF forward / back substitution for one right-hand side             */
/* ------------------------------------------------------------------ */
template <bool FAST>
SLIP_DEV int slip_solve_rhs(const SlipParams &P, SlipState *st, const SlipSolveArgs &A, const int c, const int tag, uint32_t *lds)
{
    const int tid = slip_tid(), T = slip_nthreads(), lane = slip_lane(), wave = slip_wave();
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    const int n = P.n;
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    uint64_t *scan_tmp = (uint64_t *)(lds + SLIP_LDS_SCAN);
    uint32_t *work = lds + SLIP_LDS_WORK;
    uint32_t *bm = BM_LDS ? lds + SLIP_LDS_BITMAP : P.gbitmap;
    const int wcap = P.wcap;
    dig_t *b0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                        : P.gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    unsigned long long c_read = 0, c_upd = 0, c_src = 0, c_str = 0, c_mac = 0;

    if (tid == 0 && c == 0) P.dbg[24 * (int64_t) P.n + 3072 + 0] = (int32_t) slip_realtime();      /* solve time line (diagnostic words) */
    /* b2[pinv[i]] = b[i]  (SLIP_LU_solve.c:68-75): rows keep their ids, the bitmap is indexed by position */
    for (int w = tid; w < P.bm_words; w += T) bm[w] = 0;
    if (tid == 0) { sv[SV_ERR] = 0; sv[SV_CNT0] = 0; sv[SV_CNT0 + 1] = 0; sv[SV_CNT0 + 2] = 0; sv[SV_NROWS] = 0; sv[SV_F] = n; }
    slip_block_sync();
    for (int i = tid; i < n; i += T) {
        const int32_t bl = A.blen[(int64_t) c * n + i];
        SlipRow r; r.len = bl; r.h = -1; r.tag = tag; r.bits = 0;
        if (bl != 0) {
            const int pos = P.pinv.fixed(i), lb = slip_abs(bl);
            slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
            const dig_t *src = (const dig_t *)(A.blimbs + A.boff[(int64_t) c * n + i]);
            dig_t *X = P.xd + (int64_t) i * P.xcap;
            if (lb > P.xcap) sv[SV_ERR] = 1;
            else {
                const int lw = (lb + 1) & ~1;
                for (int d = 0; d < lw; d++) X[d] = d < lb ? src[d] : 0u;
                r.bits = 32 * lb - slip_clz32(src[lb - 1]);
            }
            P.xrow[i] = r;
        }
    }
    slip_block_sync();
    if (sv[SV_ERR]) return SLIPDEV_GROW_X;

    if (tid == 0 && c == 0) P.dbg[24 * (int64_t) P.n + 3072 + 1] = (int32_t) slip_realtime();      /* solve time line (diagnostic words) */
    /* forward substitution = the sweep over ALL pivot positions (slip_forward_sub.c:61-158) */
    unsigned long long tw_[3] = {0, 0, 0}, tl_ = 0;
    slip_sweep<FAST, false>(P, st, n, tag, lds, bm, b0, b1, b2, c_read, c_upd, c_src, c_str, c_mac, tw_, &tl_);
    slip_block_sync();
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;

    if (tid == 0 && c == 0) P.dbg[24 * (int64_t) P.n + 3072 + 2] = (int32_t) slip_realtime();      /* solve time line (diagnostic words) */
    /* x <- x * det (slip_array_mul.c:19), det = rho[n-1] */
    int npat, nUdummy;
    slip_pattern(P, lds, bm, n, &npat, &nUdummy);
    const int32_t *patl = (const int32_t *)(lds + SLIP_LDS_PAT);
    auto pat_at = [&](int t) -> int { return npat <= SLIP_PAT_CAP ? patl[t] : P.pat[t]; };
    slip_block_sync();
    {
        volatile int32_t *wcnt = &sv[SV_CNT0];
        if (tid == 0) *wcnt = 0;
        slip_block_sync();
        const SlipPiv Mdet = slip_ld_piv(P.piv.at(n - 1));
        for (int t0 = 0; t0 < npat; t0 += SLIP_WORK_CAP) {
            const int te = t0 + SLIP_WORK_CAP < npat ? t0 + SLIP_WORK_CAP : npat;
            for (int tb = t0; tb < te; tb += T) {
                const int t = tb + tid;
                int queue = 0, r = 0;
                if (t < te) do {
                    r = P.row_perm.fixed(pat_at(t));
                    const SlipRow xr = P.xrow[r];
                    if (xr.len == 0) break;
                    slip_u128 y = 0; int ys = 1; int done = 0;
                    if (slip_abs(xr.len) <= 2) {
                        const uint64_t xv = slip_limb0(P.xd + (int64_t) r * P.xcap);
                        if (slip_history_small(P, xr, xv, Mdet, -1, &y, &ys)) { slip_store_small(P, r, y, ys, xr.h, tag); done = 1; }
                    }
                    if (!done) queue = 1;
                } while (0);
                /* queue slots per wave (one LDS atomic per wave, not per row on the same counter) */
                const uint64_t qm = slip_ballot(queue);
                int qbase = 0;
                if (lane == 0 && qm) qbase = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(qm));
                qbase = (int) slip_bcast0_u32((uint32_t) qbase);
                if (queue) work[qbase + slip_popc64(qm & ((1ull << lane) - 1ull))] = (uint32_t) r;
            }
            slip_block_sync();
            slip_drain(P, lds, 4, 0, 0, n, 0, *wcnt, work, b0, b1, b2);
            if (tid == 0) *wcnt = 0;
            slip_block_sync();
        }
    }
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;

    if (tid == 0 && c == 0) P.dbg[24 * (int64_t) P.n + 3072 + 3] = (int32_t) slip_realtime();      /* solve time line (diagnostic words) */
    /* back substitution (slip_back_sub.c:36-52): positions descending; x_j /= U_jj (= rho_j, the last entry
     * of U(:,j)), then x_i -= U_ij x_j for the rows above */
    {
        volatile int32_t *wcnt = &sv[SV_CNT0];
        int cur = n;
        for (;;) {
            slip_block_sync();
            const int jp = slip_bitmap_prev(bm, cur);
            if (jp < 0) break;
            cur = jp;
            const int j = P.row_perm.fixed(jp);
            SlipRow xj = P.xrow[j];
            if (xj.tag != tag || xj.len == 0) continue;
            const SlipPiv Dj = slip_ld_piv(P.piv.at(jp));
            if (slip_abs(xj.len) <= 2 && slip_abs(Dj.len) <= 2) {
                const slip_u128 y = slip_divexact128((slip_u128) slip_limb0(P.xd + (int64_t) j * P.xcap), Dj.lo, Dj.ctz, Dj.inv64);
                slip_block_sync();                                 /* every thread has read the old value */
                if (tid == 0) slip_store_small(P, j, y, slip_sgn(xj.len) * slip_sgn(Dj.len), xj.h, tag);
            } else {
                slip_block_sync();
                if (wave == 0) { const int e = slip_divexact_out(&P, j, jp, b0, b1, b2); if (e && lane == 0) sv[SV_ERR] = e; }
            }
            slip_block_sync();
            if (sv[SV_ERR]) break;
            xj = P.xrow[j];
            const uint64_t xjv = slip_limb0(P.xd + (int64_t) j * P.xcap);
            const int64_t m0 = P.Up[jp], m1 = P.Up[jp + 1] - 1;   /* the pivot is the last entry */
            for (int64_t mb = m0; mb < m1; mb += SLIP_WORK_CAP) {
                const int64_t me = mb + SLIP_WORK_CAP < m1 ? mb + SLIP_WORK_CAP : m1;
                for (int64_t mm = mb; mm < me; mm += T) {
                    const int64_t m = mm + tid;
                    int queue = 0, qi = 0;
                    if (m < me) do {
                    const int i = P.Ui[m];
                    const SlipEnt ue = P.Ue[m];
                    const int pos = P.pinv.fixed(i);
                    SlipRow xi = P.xrow[i];
                    if (xi.tag != tag) {
                        xi.len = 0; xi.h = -1; xi.bits = 0; xi.tag = tag; P.xrow[i] = xi;
                        slip_atomic_or_u32(&bm[pos >> 5], 1u << (pos & 31));
                    }
                    if (ue.len == 0) break;
                    int done = 0;
                    if (slip_abs(ue.len) <= 2 && slip_abs(xj.len) <= 2 && slip_abs(xi.len) <= 2) {
                        const int bt = (xi.bits > ue.bits + xj.bits ? xi.bits : ue.bits + xj.bits) + 1;
                        if (bt <= 126) {
                            const slip_u128 p2 = (slip_u128) slip_limb0((const dig_t *)(P.Ulimbs + ue.off)) * xjv;
                            const int s2 = slip_sgn(ue.len) * slip_sgn(xj.len), sx = slip_sgn(xi.len);
                            slip_u128 mag; int sT;
                            if (xi.len == 0) { mag = p2; sT = -s2; }
                            else {
                                const slip_u128 xv = (slip_u128) slip_limb0(P.xd + (int64_t) i * P.xcap);
                                if (sx == s2) { if (xv >= p2) { mag = xv - p2; sT = sx; } else { mag = p2 - xv; sT = -sx; } }
                                else { mag = xv + p2; sT = sx; }
                            }
                            slip_store_small(P, i, mag, sT, xi.h, tag);
                            done = 1;
                        }
                    }
                    if (!done) { queue = 1; qi = i; }
                    } while (0);
                    const uint64_t qm = slip_ballot(queue);          /* queue slots per wave */
                    int qbase = 0;
                    if (lane == 0 && qm) qbase = slip_atomic_add_i32((int32_t *) wcnt, slip_popc64(qm));
                    qbase = (int) slip_bcast0_u32((uint32_t) qbase);
                    if (queue) {
                        const int at = qbase + slip_popc64(qm & ((1ull << lane) - 1ull));
                        work[2 * at] = (uint32_t)(m - m0); work[2 * at + 1] = (uint32_t) qi;
                    }
                }
                slip_block_sync();
                slip_drain(P, lds, 5, j, jp, n, m0, *wcnt, work, b0, b1, b2);
                if (tid == 0) *wcnt = 0;
                slip_block_sync();
                if (sv[SV_ERR]) break;
            }
            if (sv[SV_ERR]) break;
        }
    }
    slip_block_sync();
    if (sv[SV_ERR]) return sv[SV_ERR] >= 6 ? sv[SV_ERR] : SLIPDEV_GROW_X;

    if (tid == 0 && c == 0) P.dbg[24 * (int64_t) P.n + 3072 + 4] = (int32_t) slip_realtime();      /* solve time line (diagnostic words) */
    /* output: numerators in pivot-position order (the order SLIP_LU_solve returns before SLIP_permute_x); every
     * right-hand side owns a region of the output slab */
    {
        const int64_t obase = (int64_t) c * A.ostride;
        uint64_t run = 0;
        for (int p0 = 0; p0 < n; p0 += T) {
            const int pos = p0 + tid;
            int32_t xl = 0; int r = -1;
            if (pos < n && ((bm[pos >> 5] >> (pos & 31)) & 1u)) { r = P.row_perm.fixed(pos); xl = P.xrow[r].len; }
            uint64_t e0, e1, t0_, t1_;
            slip_block_scan2((uint64_t) slip_limbs(xl), 0, scan_tmp, &e0, &e1, &t0_, &t1_);
            if (pos < n) {
                const int64_t off = obase + (int64_t)(run + e0);
                A.olen[(int64_t) c * n + pos] = xl;
                A.ooff[(int64_t) c * n + pos] = off;
                if (xl != 0 && (int64_t)(run + e0) + slip_limbs(xl) <= A.ostride) {
                    const dig_t *src = P.xd + (int64_t) r * P.xcap;
                    dig_t *dst = (dig_t *)(A.olimbs + off);
                    const int lw = (slip_abs(xl) + 1) & ~1;
                    for (int d = 0; d < lw; d++) dst[d] = src[d];
                }
            }
            run += t0_;
        }
        if ((int64_t) run > A.ostride) return SLIPDEV_GROW_U;       /* output region too small: the host grows it */
    }
    slip_block_sync();
    return SLIPDEV_OK;
}

/* kernel body of the solves: every workgroup draws right-hand sides from the ticket counter (they are independent,
 * slip_forward_sub.c:61-158 per column of b); rhs_done[c] marks the finished ones so that a relaunch after a grow
 * only repeats the others.  A workgroup that draws no right-hand side HELPS (round 3): a single right-hand side is one
 * chain of n source steps on one workgroup, and the long update queues of its heavy steps (slip_drain opens them exactly
 * as in the factorisation: kind 1 forward, kind 5 backward) are taken by the waves of the helpers.  st->exited counts the
 * right-hand sides that are through; the helpers leave when it reaches nrhs. */
template <bool FAST>
SLIP_DEV void slip_solve_worker(const SlipParams &P, SlipState *st, const SlipSolveArgs &A, int32_t *rhs_done, uint32_t *lds)
{
    const int tid = slip_tid(), wave = slip_wave();
    volatile int32_t *sv = (volatile int32_t *)(lds + SLIP_LDS_VARS);
    for (;;) {
        slip_block_sync();
        if (tid == 0) {
            const int t = slip_agent_add_i32(&st->ticket, 1);
            sv[SV_K] = t - P.t0; sv[SV_TAG] = t + 1;
        }
        slip_block_sync();
        const int c = sv[SV_K], tag = sv[SV_TAG];
        if (c >= A.nrhs) break;
        if (!rhs_done[c]) {
            const int status = slip_solve_rhs<FAST>(P, st, A, c, tag, lds);
            if (tid == 0) {
                if (status == SLIPDEV_OK) rhs_done[c] = 1;
                else slip_raise_stop(st, c, status);
            }
        }
        if (tid == 0) slip_agent_add_i32(&st->exited, 1);
    }
    if (!P.farm) return;
    /* no right-hand side left for this workgroup: help until the last one is through */
    const bool BM_LDS = FAST || P.bitmap_in_lds, SCR_LDS = FAST || P.scratch_in_lds;
    const int wcap = P.wcap;
    dig_t *b0 = SCR_LDS ? lds + SLIP_LDS_BITMAP + (BM_LDS ? P.bm_words : 0) + wave * 3 * wcap
                        : P.gscratch + (int64_t) wave * 3 * wcap;
    dig_t *b1 = b0 + wcap, *b2 = b1 + wcap;
    unsigned long long spins = 0;
    for (;;) {
        slip_block_sync();
        if (tid == 0) {
            int s_ = slip_ld_i32(&st->exited) >= A.nrhs ? -1 : slip_farm_peek(P, st);
            if (++spins > SLIP_SPIN_LIMIT) s_ = -1;
            sv[SV_TMP3] = s_;
        }
        slip_block_sync();
        const int s_ = sv[SV_TMP3];
        if (s_ < 0) break;
        if (s_ > 0) slip_farm_help(P, st, lds, s_ - 1, b0, b1, b2);
        else slip_sleep();
    }
}

#endif /* SLIP_REF_LU_PIPE_COLS_H */
