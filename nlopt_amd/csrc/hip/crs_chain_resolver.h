/* crs_chain_resolver.h — the accept / reject chain of a device-resolved CRS2_LM window (crs.c:125-156) advanced by ONE
 * DEDICATED WAVEFRONT out of registers (crs_chain.hip).
 *
 * Why: rounds 2-4 had whichever workgroup had just evaluated a slot advance the chain under a lock, on a control block
 * in uncached device memory — the lock, `next`, `evald[j]`, `wp`, the two f values and the re-check after the unlock are five to
 * six DEPENDENT round trips to memory per slot, ≈ 3 us (profiles/r04_crs_forward_small_n.txt).  At n = 4096 a slot's gather takes
 * 20 us and hides that; at n = 512 the window's 128 slots are gathered and evaluated in ≈ 15 us and then wait 128 x 3 us for the
 * chain: 303 k evals/s, less than the conservative passes' 467 k.  Here the chain state (next, wp, the values that landed among the
 * worst rows) lives in the registers of one wavefront that does nothing else: it polls the NEXT 64 slots' result records with one
 * load per lane, resolves the whole run of evaluated slots it finds from registers (v_readlane), issues the rowstate stores of the
 * run, waits ONCE for them to land and publishes `next` / `pk` once — two round trips per RUN instead of five per slot.
 *
 * The record of slot a is two 64-bit words, ~bits(fT) and ~bits(fM), in the zeroed control area: a word is its own "evaluated" flag
 * (zero = not yet; the one f whose complement is zero is the all-ones NaN, stored as another NaN), so the evaluating workgroup
 * publishes with two plain stores and the resolver needs no second load.
 *
 * The decisions are crs_trial's (crs.c:125-156), statement for statement (stated sequentially in oracle/port_kernels.c and in
 * tools/chain_resolver_check.cpp); the host still verifies every one of them (crs_driver.c).  The wavefront gives up (halt: every
 * waiting slot proceeds, the host recomputes what it cannot verify) when nothing was evaluated for `timeout` ticks of the 100 MHz
 * clock since the last evaluation arrived — a launch can be slow, it cannot hang on the resolver.
 *
 * This header is compiled twice: by hipcc into crs_chain_kernel<..., RES = 1>, and by g++ into tools/chain_resolver_check.cpp,
 * where 64 threads play the wavefront in lockstep and a feeder thread plays the evaluating workgroups (CH_* primitives below). */
#ifndef NLA_CRS_CHAIN_RESOLVER_H
#define NLA_CRS_CHAIN_RESOLVER_H

#ifndef CH_EXTRA
#define CH_EXTRA 32                      /* accepted values that landed among the window's worst rows */
#endif

#ifndef CH_PRIMITIVES_DEFINED            /* the device's; the CPU check defines its own before including this file */
#define CH_DEV __device__ __forceinline__
CH_DEV int ch_lane() { return (int) (threadIdx.x & 63u); }
CH_DEV uint64_t ch_ballot(bool p) { return __ballot(p); }
CH_DEV uint64_t ch_readlane_u64(uint64_t v, uint32_t l)
{
    const int ls = __builtin_amdgcn_readfirstlane((int) l);
    const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, ls), hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (v >> 32), ls);
    return (uint64_t) lo | ((uint64_t) hi << 32);
}
/* every lane reads the value of the lane IT names (ds_bpermute: the LDS crossbar, no memory) */
CH_DEV uint64_t ch_shuffle_u64(uint64_t v, uint32_t src)
{
    const uint32_t lo = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (src << 2), (int) (uint32_t) v);
    const uint32_t hi = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (src << 2), (int) (uint32_t) (v >> 32));
    return (uint64_t) lo | ((uint64_t) hi << 32);
}
CH_DEV uint64_t ch_ld64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CH_DEV void ch_st32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
CH_DEV void ch_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
/* the light form: the resolver's stores go to UNCACHED memory as agent-scope atomics, so "have landed" (vmcnt) is all a release has to
 * mean — no L2 write-back.  Used below n = 2048, where every microsecond of the resolver's turn is on the window's dependency chains
 * (n = 512: +6.5 %, n = 64: +4 %); at the headline the heavy form measured 2.3 % FASTER (47.7 k against 46.7 k evals/s, same box, three
 * runs each — the write-back / invalidate pair changes what the gather finds in L2), so it stays there */
CH_DEV void ch_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
CH_DEV void ch_sleep() { __builtin_amdgcn_s_sleep(2); }
CH_DEV uint64_t ch_clock() { return (uint64_t) wall_clock64(); }
CH_DEV double ch_f_of_bits(uint64_t b) { return __longlong_as_double((long long) ~b); }
#endif

/* what the evaluating workgroup stores for f: never zero */
#ifdef CH_PRIMITIVES_DEFINED             /* (the CPU check's primitives have one release) */
CH_DEV void ch_landed() { ch_release(); }
#endif

CH_DEV uint64_t ch_bits_of_f(double f)
{
    uint64_t b;
    __builtin_memcpy(&b, &f, sizeof b);
    b = ~b;
    return b ? b : 1ull;                 /* f was the all-ones NaN: its neighbour is a NaN too, and compares the same */
}

/* rowstate word of a worst row the chain has overwritten: bit 0 set, kind (1 trial point, 2 mutation) << 1, producer slot << 3 */
CH_DEV uint32_t ch_rowstate_word(int kind, uint32_t slot) { return 1u | ((uint32_t) kind << 1) | (slot << 3); }

/* ctrl words this wavefront publishes (offsets into chain_ctrl, in 32-bit words: crs_chain.hip static_asserts them) */
#define CH_CTRL_NEXT 2
#define CH_CTRL_HALT 3
#define CH_CTRL_NACCEPT 4
#define CH_CTRL_WP 5
#define CH_CTRL_NEXTRA 6
#define CH_CTRL_PK 7

/* All 64 lanes call; every value below is wavefront-uniform except rt / rm (lane l: record of slot next + l) and xfb / xrow
 * (lane e: the e-th value that landed among the worst rows, as chain_ctrl::xf / xrow in the lock version). */
CH_DEV void chain_resolver_wave(uint32_t *ctrl_words, const uint64_t *recs, uint32_t *rowstate, int K, int nW,
                                const int64_t *W, const double *Wf, double f_best, int64_t i0, uint64_t timeout, bool light = false)
{
    const int lane = ch_lane();
    uint32_t next = 0, wp = 0, nextra = 0, naccept = 0, halt = 0, idle = 0;
    uint64_t xfb = 0;
    int64_t xrow = 0;
    /* the worst-row list in registers: lane l holds entries l, l + 64, l + 128, l + 192 (nW <= 256), so that the walk below never
     * waits for memory and every value it branches on is wavefront-uniform (v_readlane -> scalar registers) */
    uint64_t wfb[4];
    int64_t wrow[4];
    for (int q = 0; q < 4; ++q) {
        const int idx = lane + 64 * q;
        double f = 0.;
        wrow[q] = -1;
        if (idx < nW) { f = Wf[idx]; wrow[q] = W[idx]; }
        __builtin_memcpy(&wfb[q], &f, sizeof f);
    }
    auto list_f = [&](uint32_t idx) {
        const uint32_t q = idx >> 6;
        const uint64_t v = q == 0 ? wfb[0] : q == 1 ? wfb[1] : q == 2 ? wfb[2] : wfb[3];
        return ch_f_of_bits(~ch_readlane_u64(v, idx & 63u));
    };
    auto list_row = [&](uint32_t idx) {
        const uint32_t q = idx >> 6;
        const int64_t v = q == 0 ? wrow[0] : q == 1 ? wrow[1] : q == 2 ? wrow[2] : wrow[3];
        return (int64_t) ch_readlane_u64((uint64_t) v, idx & 63u);
    };
    const double f_last = nW > 0 ? list_f((uint32_t) nW - 1u) : 0.;
    const int64_t r_last = nW > 0 ? list_row((uint32_t) nW - 1u) : -1;
    uint64_t t0 = ch_clock();
    while (next < (uint32_t) K && !halt) {
        const uint32_t s = next + (uint32_t) lane;
        uint64_t rt = 0, rm = 0;
        if (s < (uint32_t) K) { rt = ch_ld64(recs + 2 * (size_t) s); rm = ch_ld64(recs + 2 * (size_t) s + 1); }
        /* a slot can be looked at as soon as f(T) is there: the evaluating workgroup publishes it BEFORE it forms and evaluates the
         * mutation (round 5), and four trials in five are accepted on f(T) alone — the mutation's share of an evaluation leaves the
         * critical path of every slot that waits for this one.  f(M) is asked for only behind a rejected trial; if it has not arrived
         * the run ends in front of that slot */
        const uint64_t ok = ch_ballot(rt != 0);
        const uint32_t run = (ok == ~0ull) ? 64u : (uint32_t) __builtin_ctzll(~ok);
        if (run == 0) {
            if ((++idle & 255u) == 0 && ch_clock() - t0 > timeout) { halt = 1; break; }
            ch_sleep();
            continue;
        }
        uint32_t i = 0;
        while (i < run && !halt) {
            /* FAST STEP (round 5): the common case — a reflection trial that beats the current worst row, which is simply the next entry
             * of the list — decided for ALL evaluated slots of the run at once instead of one slot per trip round the scalar walk below
             * (that walk is ~100 dependent scalar / cross-lane instructions per slot: at n = 64 a 256-slot window spent most of its
             * ~140 us in it).  While no accepted value has landed among the worst rows (nextra == 0), slot next + l accepted as a trial
             * point takes list entry wp + (l - i) PROVIDED every slot before it in the run did the same; so lane l tests its own f(T)
             * against that entry, and the run's leading lanes that pass are exactly the sequential walk's next decisions.  A lane fails —
             * and the scalar walk takes over at that slot — if its trial is not accepted (rejection: the mutation and a second block),
             * if the list ends, or if the accepted value MIGHT land among the worst rows or be a new best point (tested conservatively,
             * f >= the list's last / f <= the best, so that no row index is needed here; the scalar walk decides those exactly). */
#ifndef NLA_CHAIN_NO_FASTSTEP            /* (A/B builds) */
            if (nextra == 0 && wp < (uint32_t) nW) {
                const uint32_t rel = (uint32_t) lane - i;
                const bool in = (uint32_t) lane >= i && (uint32_t) lane < run && wp + rel < (uint32_t) nW;
                const uint32_t idx = wp + (in ? rel : 0u);
                const uint32_t q0 = wp >> 6;
                const uint64_t va = q0 == 0 ? wfb[0] : q0 == 1 ? wfb[1] : q0 == 2 ? wfb[2] : wfb[3];
                uint64_t fwb = ch_shuffle_u64(va, idx & 63u);
                if ((wp & 63u) + (run - i) > 64u) {                /* (uniform) the run's entries straddle two of the list's registers */
                    const uint64_t vb = q0 == 0 ? wfb[1] : q0 == 1 ? wfb[2] : wfb[3];
                    const uint64_t fb2 = ch_shuffle_u64(vb, idx & 63u);
                    if ((idx >> 6) != q0) fwb = fb2;
                }
                const double fw_l = ch_f_of_bits(~fwb), fT_l = ch_f_of_bits(rt);
                const bool ok = in && fT_l < fw_l && !(fT_l >= f_last) && !(fT_l <= f_best);
                const uint64_t m = ch_ballot(ok) >> i;
                const uint32_t P = (m == ~0ull) ? 64u : (uint32_t) __builtin_ctzll(~m);
                if (P) {
                    if ((uint32_t) lane >= i && (uint32_t) lane < i + P) ch_st32(&rowstate[wp + rel], ch_rowstate_word(1, next + (uint32_t) lane));
                    wp += P; naccept += P; i += P;
                    if (i >= run) break;
                }
            }
#endif
            const uint32_t j = next + i;
            const double fT = ch_f_of_bits(ch_readlane_u64(rt, i));
            const uint64_t rmi = ch_readlane_u64(rm, i);
            const double fM = ch_f_of_bits(rmi);
            /* the current worst: the next untouched row of the list, or a value that landed among them */
            double fw = -__builtin_huge_val();
            int64_t rw = -1;
            int xi = -1;
            if (wp < (uint32_t) nW) { fw = list_f(wp); rw = list_row(wp); }
            for (uint32_t e = 0; e < nextra; ++e) {
                const double xe = ch_f_of_bits(~ch_readlane_u64(xfb, e));
                const int64_t re = (int64_t) ch_readlane_u64((uint64_t) xrow, e);
                if (rw < 0 || xe > fw || (xe == fw && re > rw)) { fw = xe; rw = re; xi = (int) e; }
            }
            if (rw < 0) { halt = 1; break; }                    /* beyond the rows this launch knows */
            int kind = 0;
            double fnew = 0;
            if (fT < fw) { kind = 1; fnew = fT; }               /* crs.c:135 */
            else if (rmi == 0) break;                            /* rejected, and f(M) is still on its way: look again later */
            else if (fM < fw) { kind = 2; fnew = fM; }           /* the mutation of crs.c:139-146, accepted at :135 */
            if (kind) {
                if (xi >= 0) {                                   /* entry xi leaves the list: the last one takes its place */
                    const uint64_t lf = ch_readlane_u64(xfb, nextra - 1);
                    const int64_t lr = (int64_t) ch_readlane_u64((uint64_t) xrow, nextra - 1);
                    if (lane == xi) { xfb = lf; xrow = lr; }
                    --nextra;
                } else {
                    if (lane == 0) ch_st32(&rowstate[wp], ch_rowstate_word(kind, j));
                    ++wp;
                }
                ++naccept;
                /* the new value may itself be among the worst that are left */
                if (nW > 0 && (fnew > f_last || (fnew == f_last && rw > r_last))) {
                    if (nextra == CH_EXTRA) halt = 1;
                    else {
                        uint64_t fb;
                        __builtin_memcpy(&fb, &fnew, sizeof fb);
                        if (lane == (int) nextra) { xfb = fb; xrow = rw; }
                        ++nextra;
                    }
                }
                if (fnew < f_best || (fnew == f_best && rw < i0)) {           /* a new best: everything behind started from the old one */
                    halt = 2u | ((j + 1u) << 8);
                    if (lane == 0) ch_st32(&ctrl_words[CH_CTRL_HALT], halt);  /* at once: workgroups that draw a ticket for a later slot leave (crs_chain.hip) */
                }
            }
            i += (kind == 1) ? 1u : 2u;
        }
        if (i == 0 && !halt) {                                   /* the front slot waits for its f(M): nothing to publish */
            if ((++idle & 255u) == 0 && ch_clock() - t0 > timeout) { halt = 1; break; }
            ch_sleep();
            continue;
        }
        next += i;
        idle = 0;
        t0 = ch_clock();                                         /* the timeout counts from the last evaluation that arrived */
        if (halt) break;
        if (light) ch_landed(); else ch_release();               /* the run's rowstate stores have landed before `next` moves */
        if (lane == 0) {
            ch_st32(&ctrl_words[CH_CTRL_NEXT], next);
            ch_st32(&ctrl_words[CH_CTRL_PK], next | ((next - wp) << 16));
        }
    }
    if (light) ch_landed(); else ch_release();
    if (lane == 0) {
        if (halt) { ch_st32(&ctrl_words[CH_CTRL_NEXT], (uint32_t) K + 2u); ch_st32(&ctrl_words[CH_CTRL_PK], 0xffffffffu); }
        ch_st32(&ctrl_words[CH_CTRL_HALT], halt);               /* 0 ran to the end, 1 gave up (list exhausted, too many landed values, timeout), 2 | (j + 1) << 8: new best at slot j */
        ch_st32(&ctrl_words[CH_CTRL_NACCEPT], naccept);         /* for post-mortems: nothing on the device reads these three */
        ch_st32(&ctrl_words[CH_CTRL_WP], wp);
        ch_st32(&ctrl_words[CH_CTRL_NEXTRA], nextra);
    }
}

#endif
