/* isres_stochrank.h — the two kernels of the ISRES stochastic-ranking pipeline (isres.c:206-228 as a systolic pipeline; the comment in
 * isres_kernels.hip in front of the include describes it).  A header of its own because it is compiled twice: by hipcc as part of isres_kernels.hip (SR_KERNEL,
 * SR_SHARED_INT and SR_WAIT_VMCNT0 expand to the device constructs they replace — the machine code is what it was before the
 * kernels moved here), and by g++ into tools/stochrank_check.cpp, where the 64 lanes of a unit are threads in lockstep, the DPP shifts
 * and v_readfirstlane exchanges between barriers, all units of a small pipeline run at once, and the kernel must reproduce the
 * reference's double loop.  (Round 4's read-ahead variant — poll / load / store moved off the head of the 64-tick blocks — ran on the
 * MI355X in round 5: same results, 54.99 against 55.31 ms per generation at config 3, i.e. nothing: the tick is not waiting for
 * memory.  Deleted; profiles/r05_staged_ab.txt.) */
#ifndef NLA_ISRES_STOCHRANK_H
#define NLA_ISRES_STOCHRANK_H
#ifndef SR_KERNEL
#define SR_KERNEL __global__ __launch_bounds__(64)
#define SR_SHARED_INT(name) __shared__ int name
#define SR_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifndef SR_SEG_WORDS
#define SR_SEG_WORDS 638976ull           /* = NLA_MT_SEG_WORDS (include/nlopt_amd.h): 1024 regenerations of 624 words per segment of the device stream */
#endif
#ifndef SR_SETPRIO_HIGH
#define SR_SETPRIO_HIGH() __builtin_amdgcn_s_setprio(3)
#endif
#ifndef SR_CLOCK
#define SR_CLOCK() ((uint64_t) wall_clock64())       /* 100 MHz */
#endif
#ifndef SR_ANY                           /* does any lane of the wavefront say so? */
#define SR_ANY(p) (__ballot(p) != 0ull)
#endif
#ifndef SR_PF_K
#define SR_PF_K 32                       /* the tick of a block at which the next block's inputs are loaded (tuning builds: -DSR_PF_K=...) */
#endif
#define SR_UNWRITTEN 0xFFFFFFFFFFFFFFFFull    /* no element looks like this (bits 28-30 of the high word are never set: isres_pack, SR_PINF_HI) */
#define SR_GATE_TIMEOUT 400000000ull    /* 4 s without the block's bits arriving: the generator's launch failed or never ran */

#define SR_DPP_SHL1 0x130               /* wave_shl:1 — lane i reads lane i+1 */
#define SR_DPP_SHR1 0x138               /* wave_shr:1 — lane i reads lane i-1 */
#define SR_PINF_LO 0xFFFFF000u
#define SR_PINF_HI 0x0FFFFF00u
__device__ __forceinline__ uint32_t sr_dpp(uint32_t old, uint32_t src, const int ctrl_is_shl)
{
    return ctrl_is_shl ? (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) src, SR_DPP_SHL1, 0xf, 0xf, false)
                       : (uint32_t) __builtin_amdgcn_update_dpp((int) old, (int) src, SR_DPP_SHR1, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t sr_ld(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sr_st(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

SR_KERNEL void isres_stochrank_kernel(int64_t pop, int64_t nsweeps, uint64_t *__restrict__ streams,
                                                              int *__restrict__ progress, const uint64_t *bits /* written by mt_rankbits_kernel on another stream WHILE this kernel runs: neither
                                                                                                                 * read-only nor unaliased for the compiler, or it may lift a row's loads above the gate's acquire fence */,
                                                              int64_t rowwords, int *__restrict__ ticket, uint8_t *__restrict__ swapped_out,
                                                              const int *__restrict__ gate, uint64_t gate_g_rank0, int64_t gate_nrows)
{
    SR_SHARED_INT(s_unit);
    const int lane = threadIdx.x;
    SR_SETPRIO_HIGH();                          /* every tick is on the pipeline's serial path; whatever shares the SIMD (the generator of the bits) takes the slots left over */
    if (lane == 0) s_unit = atomicAdd(ticket, 1);
    __syncthreads();
    const int64_t unit = s_unit;
    const int64_t stage = unit * 64 + lane;
    const bool active = stage < nsweeps;
    const uint64_t *in = streams + (size_t) unit * (size_t) pop;
    uint64_t *out = streams + (size_t) (unit + 1) * (size_t) pop;
    (void) progress;                            /* (the counter protocol of rounds 2-4; the parameter stays in the launcher's signature) */
    const uint64_t *brow = bits + (size_t) (active ? stage : 0) * (size_t) rowwords;
    const int ipop = (int) pop, rw1 = (int) rowwords - 1;
    const int half = lane >> 5;                 /* row word of the window of block b starts at word b - 1 - half */
    const int cut = (63 - 2 * lane) & 63;       /* ... at this bit (1..63, never 0) */
    uint32_t c_lo = 0, c_hi = 0, o_lo = 0, o_hi = 0;         /* carry and output element of this stage ("-inf"), as two words */
    uint32_t vin_lo = 0, vin_hi = 0, ob_lo = 0, ob_hi = 0;   /* the unit's inputs / outputs of the current block, one per lane */
    uint32_t swv = 0;                                        /* bit 31: this stage swapped at least once */
    const uint32_t amask = active ? 0x80000000u : 0u;
    auto clampw = [&](int w) { return w < 0 ? 0 : (w > rw1 ? rw1 : w); };
    if (gate) {
        /* the rows of uniform bits are still being produced by mt_rankbits_kernel on another stream (isres_driver.c, "amd_isres_gated"): the
         * generator's wavefronts each add 1 to gate[c] when they have written all they owe to sweeps 64 c .. 64 c + 63 — this unit's rows —
         * and block c is complete when as many have done so as segments of the stream intersect the block's words (the formula of
         * nla_rankbits_gate_target, hip/mt_kernels.hip; gate_g_rank0 = stream index of the ranking's first word, gate_nrows = sweeps the
         * generator was asked for).  Wait BEFORE the first load of a row (a load ahead of the count could leave a stale line in this CU's
         * cache), then make the other stream's stores visible */
        const int *g = gate + unit;
        const uint64_t gb0 = gate_g_rank0 + 128ull * (uint64_t) (pop - 1) * (uint64_t) unit;
        const int64_t gr1 = 64 * (unit + 1) < gate_nrows ? 64 * (unit + 1) : gate_nrows;
        const uint64_t gb1 = gate_g_rank0 + 2ull * (uint64_t) (pop - 1) * (uint64_t) gr1;
        const int want = gb1 > gb0 ? (int) ((gb1 - 1) / SR_SEG_WORDS - gb0 / SR_SEG_WORDS + 1) : 0;
        /* bounded: a generator launch that faulted or never ran must not hang the device.  The unit that gives up says so in the word
         * behind the generator's ticket (gate[units + 2]); it and every later unit then run through on whatever bits there are, and the
         * host — which reads the word after the launch — waits for the generator's stream and ranks again without gates (isres_driver.c) */
        int *gate_err = const_cast<int *>(gate) + (gate_nrows + 63) / 64 + 2;
        if (lane == 0) {
            const uint64_t t0 = SR_CLOCK();
            unsigned spins = 0;
            while (__hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(8);
                if ((++spins & 1023u) == 0 && (__hip_atomic_load(gate_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || SR_CLOCK() - t0 > SR_GATE_TIMEOUT)) {
                    __hip_atomic_store(gate_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    /* HAND-OVER THROUGH THE ELEMENTS (round 5).  Rounds 2-4 passed a block of 64 elements from unit to unit behind a progress counter:
     * every block of a unit began with two dependent round trips to memory (lane 0 polls the upstream counter, then the lanes load
     * their inputs) and ended with a third (the output stores must have landed before the unit's own counter moves) — 2.8 us of stalls
     * around 2.9 us of ticks, on every block of every unit (measured by varying the block length: 64 elements 5.7 us per block, 32: 3.9,
     * 16: 3.1, 8: 3.0; profiles/r05_isres_handoff.txt).  Now the buffers between the units start out as SR_UNWRITTEN (the launcher
     * fills them), a producer just stores its 64 outputs — each a single 64-bit store, so an element is either there or not — and a
     * consumer loads its next block's 64 inputs HALF A BLOCK AHEAD and looks at them when it gets there: all written -> go; otherwise
     * load again until they are.  Nobody waits for a store, and the load on a unit's path has had 32 ticks to land.  A consumer that runs
     * too close behind its producer misses, waits a round trip, and from then on runs that much later — at the same rate.  Config 3:
     * 20.3 -> 14.7 ms per launch in the same source (17.7 ms with round 4's kernel), generation 46.3 -> 40.6 ms.
     * (The first load is issued BEFORE the row words below: at the top of the block loop the compiler's wait for this register is the
     * stricter of "as left by the code in front of the loop" and "as left by the previous block".) */
    const uint64_t pinf = ((uint64_t) SR_PINF_HI << 32) | SR_PINF_LO;
    uint64_t nxt = lane < ipop ? sr_ld(in + lane) : pinf;
    uint64_t wa = brow[clampw(-1 - half)], wb = brow[clampw(0 - half)], wp = brow[clampw(1 - half)];
    const int nblk = (ipop + 63) / 64 + 2;      /* the last output leaves lane 63 at tick pop + 126 */
    for (int b = 0; b < nblk; ++b) {
        const int tb = b * 64;
        /* u < PF bits of ticks tb .. tb+63 of this stage: row bits tb - 2 lane - 1 + k */
        const uint64_t win = (wa >> cut) | (wb << (64 - cut));
        const uint32_t wlo = (uint32_t) win, whi = (uint32_t) (win >> 32);
        wa = wb; wb = wp;
        /* the block's 64 inputs, one per lane (lane 0 consumes one per tick, the rest move down); past the end of the stream: +inf */
        uint64_t inb = pinf;
        if (tb < ipop) {
            const bool inr = tb + lane < ipop;
            inb = nxt;
            /* (the first look is at the load issued half a block ago and stands outside the retry loop: a loop header would wait for
             * every outstanding memory operation before each look) */
            if (SR_ANY(inr && inb == SR_UNWRITTEN)) {
                do {
                    __builtin_amdgcn_s_sleep(1);
                    if (inr) inb = sr_ld(in + tb + lane);
                } while (SR_ANY(inr && inb == SR_UNWRITTEN));
            }
            if (!inr) inb = pinf;
        }
        vin_lo = (uint32_t) inb; vin_hi = (uint32_t) (inb >> 32);
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            /* input: lane 0 from the unit's input block, the others from their left neighbour's output of the previous tick */
            const uint32_t x_lo = sr_dpp(vin_lo, o_lo, 0), x_hi = sr_dpp(vin_hi, o_hi, 0);
            vin_lo = sr_dpp(vin_lo, vin_lo, 1); vin_hi = sr_dpp(vin_hi, vin_hi, 1);
            /* by fval if u < PF (bit k of the window: isres.c:210) or both penalties are zero (bit 31 of both high words),
             * else by penalty (:211-212, :220) — sign bits instead of booleans: the tick stays straight-line integer code */
            const uint32_t usef = (((k & 32) ? whi : wlo) << (31 - (k & 31))) | (c_hi & x_hi);
            const int32_t df = (int32_t) (x_lo >> 12) - (int32_t) (c_lo >> 12);                    /* < 0: fval[carry] > fval[x]   (:213) */
            const int32_t dp = (int32_t) (x_hi & 0x0FFFFF00u) - (int32_t) (c_hi & 0x0FFFFF00u);    /* < 0: penalty[carry] > penalty[x] */
            const int32_t d = ((int32_t) usef < 0) ? df : dp;
            const bool swap = active && d < 0;
            swv |= (uint32_t) d & amask;
            o_lo = swap ? x_lo : c_lo;          /* emitted: the smaller of the pair */
            o_hi = swap ? x_hi : c_hi;
            c_lo = swap ? c_lo : x_lo;          /* kept: the larger */
            c_hi = swap ? c_hi : x_hi;
            /* lane 63's outputs move down one lane per tick: after tick tb + 62 lane l holds output tb - 128 + l of the unit */
            ob_lo = sr_dpp(o_lo, ob_lo, 1); ob_hi = sr_dpp(o_hi, ob_hi, 1);
            /* the row word used two blocks from now and, half a block ahead, the next block's inputs; the 64 outputs completed at
             * tick 62 are stored there and nobody waits for the stores as such.  (Measured and dropped: keeping the outputs in two
             * registers and storing them at tick 8 of the next block, so that the top of a block — which must drain the one counter
             * loads and stores share before it can use a loaded value — never meets a young store: 14.8 against 14.7 ms per launch.) */
            /* (also measured and dropped, call 18: a unit that missed its look-ahead falling back by s_sleep(24 .. 96) so that the next
             * ones hit — 14.3-15.9 against 14.1-14.3 ms per launch; the look-ahead's tick, 16 .. 48: no difference) */
            if (k == 12) wp = brow[clampw(b + 2 - half)];
            if (k == SR_PF_K) { if (tb + 64 < ipop) nxt = tb + 64 + lane < ipop ? sr_ld(in + tb + 64 + lane) : pinf; }
            if (k == 62) {
                const int base = tb - 128;
                if (base >= 0 && base + lane < ipop) sr_st(out + base + lane, ((uint64_t) ob_hi << 32) | ob_lo);
            }
        }
    }
    if (active) swapped_out[stage] = (uint8_t) (swv >> 31);
}

#endif
