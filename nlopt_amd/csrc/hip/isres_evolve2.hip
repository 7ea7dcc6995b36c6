/* isres_evolve2.hip — the ISRES evolve phase (mutation isres.c:234-252, differential variation :253-280) in
 * parallel over individuals.
 *
 * Why it is hard: every draw is an nlopt_nrand taken from ONE stream, and how many deviates individual k consumes
 * (1 + 2 per mutated coordinate + one more per out-of-bounds redraw, :245-248,270-273) depends on the deviates it
 * meets — so where individual k+1 starts reading depends on everything before it.  isres_kernels.hip walks that
 * chain with one workgroup (kept as the fallback); here the chain is cut open by MULTI-START LOOK-UP:
 *
 *   stage   (one workgroup per individual of a block of EVM consecutive individuals)  everything that does not
 *           depend on the stream position: parent rows, the differential step and which coordinates mutate
 *           (variation), compacted into a workspace.
 *   scan    (one workgroup per individual, one LANE per candidate start)  individual i's start is predicted from the
 *           block's exactly known first position, the deviates the individuals before it consume at least, and the
 *           running redraw rate; for EACH of the EVD stream positions around the prediction a lane walks the
 *           individual's coordinates sequentially (parent data broadcast from LDS, the deviate window shared) and
 *           records how many deviates the individual would consume FROM THAT START: E[i][d], plus the redraw count
 *           at every 1/64 of the coordinates (T) so that the write pass can start all of its lanes at once.
 *   chain   (one thread, tables in LDS)  start_0 is exact; start_{i+1} = start_i + E[i][start_i - base_i] — a table
 *           look-up per individual instead of the individual's whole arithmetic.  The walk stops where the true start
 *           leaves the predicted window (the next round re-anchors there), where a variation individual needs a row
 *           that an earlier individual of the same block rewrites (isres.c:260 reads the CURRENT physical row k+1),
 *           or where the deviates generated so far run out.
 *   write   (one wavefront per resolved individual)  the same arithmetic once more from the now exact start, lanes on
 *           contiguous coordinate chunks whose stream offsets come from T, producing the child's x and sigma rows.
 *
 * Everything is exact: the same expressions in the same order as the serial kernel (sigma' = sigma exp(taup z_k +
 * tau z), capped; x = x_parent + sigma' z redrawn while out of bounds; sigma_new = sigma + 0.2 (sigma' - sigma));
 * a wrong prediction costs a shorter round, never a different result.  Work is EVD x the serial arithmetic, spread
 * over EVM x EVD lanes per round.
 */
#include "dev_common.h"
#include <algorithm>
#include "../../../include/nlopt_amd.h"

#define EVD 256                 /* candidate starts per individual (window of stream positions) */
#define EVM 256                 /* a round's block is bm = 1 or 2 of these individuals (the variation phase's / the mutation phase's) */
#ifndef EVMX
#define EVMX 1024               /* the largest block: the mutation phase's (round 5: 512; round 6, with the chain kernel over segment tables: 768 / 1024 / 1536 measured,
                                 * 63 rounds of ~680 individuals per mutation phase at config 3 instead of 96 of ~450: -0.8 ms per generation) */
#endif
#ifndef EV2_MUT_BLOCK
#define EV2_MUT_BLOCK EVMX      /* (A/B builds: -DEV2_MUT_BLOCK=256) */
#endif
#ifndef EV2_VAR_BLOCK
#define EV2_VAR_BLOCK EVM       /* the variation phase's block (512 measured at the end of round 6: 37 -> 31 rounds, no time gained — half of its
                                 * rounds end at a row dependency, isres.c:260) */
#endif
#define EV2_MAXN 1150           /* LDS staging limit (same as the serial LDS kernel) */
#define EV2_ZW(n) (EVD + 3 * (n) + 65)      /* deviates staged per individual: window + 1 + 2n + room for n + 64 redraws */
static_assert(16 * EV2_ZW(EV2_MAXN) < 65535, "a segment's deviates fit the 16-bit segment table");
#define EV2_ZPAD 2                          /* doubles of LDS behind the staged deviates (the scan reads one deviate ahead) */

struct ev2_args {
    int n, ld, phase;
    int bm;                             /* individuals of this round's block: EVMX in the mutation phase, EVM in the variation phase */
    int64_t pop, survivors, zcount;
    double taup, tau;
    const double *lb, *ub, *z;
    const int32_t *irank, *inv;         /* inv[irank[k]] = k */
    double *X, *S;
    const double *x0c;                  /* copy of physical row 0 taken before the variation loop (isres.c:253) */
    int64_t *state;                     /* [0] next individual, [1] next deviate, [2] deviates ran out, [9] resolved in the last round,
                                           [10] stuck (fallback needed), [11] rounds, [12] first individual of the last round */
    double *rho;                        /* [2*phase] decayed redraw sum, [2*phase+1] decayed sum of the redraws EXPECTED (ws_mu) of the same individuals */
    double *ws_mu;                      /* per block slot: redraws the individual is expected to make (stage kernel; see the scan's prediction) */
    const double *mu_rp;                /* mutation phase: the same per PARENT, by rank position p < survivors (ev2_parent_mu_kernel, once per generation) */
    int32_t *ws_nact, *ws_act;          /* per block slot: number of mutated coordinates (-1: past the end), their indices */
    double *ws_xi, *ws_sg, *ws_xpre;    /* parent x / sigma of the mutated coordinates (compacted); x of the others */
    int16_t *T;                         /* EVM x 64 x EVD: redraws before coordinate chunk c when starting at d */
    int16_t *E;                         /* EVM x EVD: deviates consumed from candidate start d, -1 window exceeded, -2 deviates ran out */
    int64_t *ws_base, *ws_start;        /* window origin / exact start per slot */
    /* segment tables (round 6): built by the scan launch itself — the LAST workgroup of a segment's 16 individuals to finish composes
     * the segment's look-ups for every candidate start — so that the chain kernel only strings segments together */
    uint32_t *segcnt;                   /* per segment: scan workgroups that have finished (zeroed by the chain kernel) */
    uint16_t *SG;                       /* EVMX/16 x EVD: deviates the whole segment consumes from start d of its first individual (<= 16 EV2_ZW(EV2_MAXN) < 65535), 0xffff it cannot be crossed */
    int2 *SC;                           /* ... {individuals resolved | (why it stopped & 0xff) << 8, deviates consumed by them} */
    int32_t *SP;                        /* EVMX/16 x 16 x EVD: start of individual j of the segment, relative to the segment's start */
};

/* ---- stage ------------------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void ev2_stage_kernel(ev2_args A)
{
    __shared__ int s_cnt[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = blockIdx.x;
    const int n = A.n, ld = A.ld;
    /* (the words of the state a kernel needs are read together at its top: every dependent round trip to memory is 1-2 us of a
     * kernel that lasts 5-70 us and runs ~250 times per generation) */
    const int64_t st0 = A.state[0], st2 = A.state[2], st10 = A.state[10];
    if (st2 || st10) return;
    const int64_t k = st0 + i, kend = A.phase == 0 ? A.pop : A.survivors;
    if (k >= kend) { if (tid == 0) A.ws_nact[i] = -1; return; }
    const int64_t rk = A.irank[k];
    int32_t *act = A.ws_act + (size_t) i * n;
    double *wxi = A.ws_xi + (size_t) i * n, *wsg = A.ws_sg + (size_t) i * n, *wpre = A.ws_xpre + (size_t) i * n;
    /* (the mutation phase is not staged: its scan and write workgroups read the parent's rows themselves, ev2_scan0_kernel) */
    /* differential variation of survivor k, in place: x + 0.85 (x0 - physical row k+1) unless it is the last survivor;
     * coordinates that leave the box (or all of them, for the last survivor) are mutated from the survivor's own x, sigma */
    const double GAMMA = 0.85;
    const bool lastsurv = (k + 1 == A.survivors), self = (k + 1 == rk);
    const double *xr = A.X + (size_t) rk * ld, *sr = A.S + (size_t) rk * ld, *kr = A.X + (size_t) (k + 1) * ld;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + tid;
        double xv = 0, sv = 0, xnew = 0;
        bool mut = false;
        if (j < n) {
            xv = xr[j]; sv = sr[j];
            xnew = xv;
            if (!lastsurv) xnew = xv + GAMMA * (A.x0c[j] - (self ? xv : kr[j]));
            mut = lastsurv || xnew < A.lb[j] || xnew > A.ub[j];
            wpre[j] = xnew;
        }
        const unsigned long long bal = __ballot(mut);
        if (lane == 0) s_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base + __popcll(bal & ((1ull << lane) - 1));
        for (int w = 0; w < wave; ++w) off += s_cnt[w];
        if (mut) { act[off] = j; wxi[off] = xv; wsg[off] = sv; }
        __syncthreads();
        if (tid == 0) s_base += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
    }
    if (tid == 0) { A.ws_nact[i] = s_base; A.ws_mu[i] = (double) s_base; }       /* variation: no better guess than the running rate per mutated coordinate */
}

/* ---- segment tables (in the scan launch's tail) --------------------------------------------------------------------- */
#define EV2_SEG 16                     /* individuals per segment */
#define EV2_OUT 0x40000000             /* a window origin no start of a block can be within EVD of */
/* does slot q of the block stop every walk (past the end of the phase; variation: isres.c:260 reads the CURRENT physical row k + 1,
 * which must not be rewritten inside this block before k; a window origin out of all range)?  *base = its window origin otherwise */
/* what one workgroup of a launch stores and another loads (E, the window origins, the counts): relaxed agent-scope accesses — they go past
 * the caches that are not shared by all compute units, so the hand-over needs no release / acquire fence (a fence writes back and
 * invalidates a whole L2: measured, +37 us per scan launch with 512 workgroups fencing) */
__device__ __forceinline__ void ev2_st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ev2_st_agent64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ev2_ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ev2_ld_agent64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool ev2_slot_stops(const ev2_args &A, int q, int64_t st0, int64_t st1, long long *base)
{
    const int na = (int) ev2_ld_agent(reinterpret_cast<const uint32_t *>(A.ws_nact + q));
    *base = 0;
    if (na < 0) return true;
    const long long k = st0 + q, k1 = k + 1 < A.pop ? k + 1 : A.pop - 1;
    const long long o = A.inv ? A.inv[k1] : -1;
    const bool dep = A.phase == 1 && k + 1 < A.pop && o >= st0 && o < k;
    const long long b = (long long) ev2_ld_agent64(reinterpret_cast<const uint64_t *>(A.ws_base + q)), brel = b - st1;
    if (dep || brel >= EV2_OUT || brel <= -EV2_OUT) return true;
    *base = b;
    return false;
}
/* Every scan workgroup of a round ends here (256 threads = EVD candidate starts).  The last of a segment's 16 to arrive walks the segment
 * for every candidate start d of its first individual — start_{j+1} = start_j + E[j][start_j - base_j], the chain's own look-ups — and leaves
 * what the chain kernel needs to cross the segment in ONE look-up: the deviates consumed (SG), and for a walk that stops inside, how far it got
 * and why (SC); the individual starts along the way (SP) are what the chain kernel hands to the write pass.  (Rounds 3-5 did this in the chain
 * kernel — ONE workgroup that first copied the block's 128 KB of E into LDS: 37-45 us per round on the serial path of ~135 rounds per
 * generation, and a workgroup that needs a compute unit nobody else holds LDS on, §2.2.) */
__device__ __forceinline__ void ev2_segment_tail(const ev2_args &A, const int i, const int64_t st0, const int64_t st1, double *sm)
{
    __shared__ int s_last, s_stop[EV2_SEG];
    __shared__ long long s_sb[EV2_SEG];
    const int tid = threadIdx.x, seg = i / EV2_SEG, i0 = seg * EV2_SEG;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           /* this workgroup's E row / window origin / count have landed before it is counted */
    __syncthreads();
    if (tid == 0) s_last = __hip_atomic_fetch_add(&A.segcnt[seg], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == EV2_SEG - 1;
    __syncthreads();
    if (!s_last) return;
    int16_t *sE = reinterpret_cast<int16_t *>(sm);             /* 16 x EVD entries: 8 KB of the launch's dynamic LDS (its contents are done with) */
    {
        const uint64_t *src = reinterpret_cast<const uint64_t *>(A.E + (size_t) i0 * EVD);
        uint64_t *dst = reinterpret_cast<uint64_t *>(sE);
#pragma unroll
        for (int q = 0; q < EV2_SEG * EVD / 4 / 256; ++q) dst[q * 256 + tid] = ev2_ld_agent64(src + q * 256 + tid);
    }
    if (tid < EV2_SEG) { long long b; s_stop[tid] = ev2_slot_stops(A, i0 + tid, st0, st1, &b); s_sb[tid] = b; }
    __syncthreads();
    const int d = tid;
    int cnt = 0, why = 0;
    long long p = s_sb[0] + d;
    const long long p0 = p;
    int32_t *sp = A.SP + (size_t) seg * EV2_SEG * EVD + d;     /* [segment][individual][start]: a wavefront's stores are consecutive */
#pragma unroll 1
    for (int j = 0; j < EV2_SEG; ++j) {
        const long long dj = p - s_sb[j];
        if (s_stop[j]) { why = -11; break; }
        if (dj < 0 || dj >= EVD) { why = -10; break; }
        const int e = sE[j * EVD + (int) dj];
        if (e < 0) { why = e; break; }
        sp[j * EVD] = (int) (p - p0);
        p += e; ++cnt;
    }
    A.SG[seg * EVD + d] = cnt == EV2_SEG ? (uint16_t) (p - p0) : (uint16_t) 0xffffu;
    A.SC[seg * EVD + d] = make_int2(cnt | ((why & 0xff) << 8), (int) (p - p0));
}

/* ---- scan -------------------------------------------------------------------------------------------------------- */
/* PH0 (mutation phase, isres.c:234-252): nothing is staged — every coordinate of child k mutates from the rows of its parent
 * irank[k % survivors], which no child overwrites, so the workgroup reads them itself; the expectation of its predecessors' redraws comes
 * from the per-parent table mu_rp.  !PH0 (variation): from the stage kernel's compacted workspace. */
template <bool PH0>
__device__ __forceinline__ void ev2_scan_body(const ev2_args &A, const int i, double *sm)
{
    __shared__ long long s_red[4];
    __shared__ double s_mred[4];
    __shared__ long long s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = A.n;
    const int64_t st0 = A.state[0], st1 = A.state[1], st2 = A.state[2], st10 = A.state[10];
    const double rho_r = A.rho[2 * A.phase], rho_a = A.rho[2 * A.phase + 1];
    if (st2 || st10) return;
    int na;
    if (PH0) {
        const int64_t k = st0 + i;
        na = k < A.pop ? n : -1;
        if (tid == 0) { ev2_st_agent(reinterpret_cast<uint32_t *>(A.ws_nact + i), (uint32_t) na); A.ws_mu[i] = na < 0 ? 0.0 : A.mu_rp[k % A.survivors]; }     /* (the chain kernel's inputs; the count also the segment tail's) */
    } else
        na = A.ws_nact[i];
    if (na >= 0) {
    /* observed / expected redraws of the individuals resolved lately (decayed sums kept by the chain kernel); before anything was
     * resolved: the expectation as it is (mutation), nothing (variation: its "expectation" is the mutated-coordinate count) */
    const double rhoc = rho_a > 0 ? rho_r / rho_a : (PH0 ? 1.0 : 0.0);
    /* predicted start: the exact start of the block + what the individuals before this one consume at least
     * (1 + 2 per mutated coordinate) + the redraws expected of them */
    {
        long long acc = 0;
        double macc = 0;
        if (PH0) { acc = (tid == 0) ? (long long) i * n : 0; for (int q = tid; q < i; q += 256) macc += A.mu_rp[(st0 + q) % A.survivors]; }
        else for (int q = tid; q < i; q += 256) { acc += A.ws_nact[q]; macc += A.ws_mu[q]; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { acc += __shfl_xor(acc, m, 64); macc += __shfl_xor(macc, m, 64); }
        if (lane == 0) { s_red[wave] = acc; s_mred[wave] = macc; }
        __syncthreads();
        if (tid == 0) {
            const long long ab = s_red[0] + s_red[1] + s_red[2] + s_red[3];
            const double mb = s_mred[0] + s_mred[1] + s_mred[2] + s_mred[3];
            s_base = st1 + i + 2 * ab + (long long) floor(rhoc * mb) - EVD / 2;
            ev2_st_agent64(reinterpret_cast<uint64_t *>(A.ws_base + i), (uint64_t) s_base);
        }
        __syncthreads();
    }
    const int64_t base = s_base;
    double *xi = sm, *sg = sm + n, *lo = sm + 2 * n, *hi = sm + 3 * n, *smax = sm + 4 * n, *zw = sm + 5 * n;
    const double sqn = sqrt((double) n);
    if (PH0) {
        const int64_t ri = A.irank[(st0 + i) % A.survivors];
        const double *xr = A.X + (size_t) ri * A.ld, *sr = A.S + (size_t) ri * A.ld;
        for (int a = tid; a < na; a += 256) { xi[a] = xr[a]; sg[a] = sr[a]; lo[a] = A.lb[a]; hi[a] = A.ub[a]; smax[a] = (A.ub[a] - A.lb[a]) / sqn; }
    } else {
        const int32_t *act = A.ws_act + (size_t) i * n;
        const double *wxi = A.ws_xi + (size_t) i * n, *wsg = A.ws_sg + (size_t) i * n;
        for (int a = tid; a < na; a += 256) {
            const int j = act[a];
            xi[a] = wxi[a]; sg[a] = wsg[a]; lo[a] = A.lb[j]; hi[a] = A.ub[j]; smax[a] = (A.ub[j] - A.lb[j]) / sqn;
        }
    }
    const int ZW = EVD + 3 * na + 65;
    const int64_t avail = A.zcount - base;
    const int zwlen = (int) (avail < ZW ? (avail < 0 ? 0 : avail) : ZW);
    const bool zw_cut = avail < ZW;
    for (int q = tid; q < zwlen; q += 256) { const int64_t g = base + q; zw[q] = g >= 0 ? A.z[g] : 0.0; }
    __syncthreads();
    /* lane = candidate start d: the coordinates one after the other, exactly the serial loop (isres.c:236-251,266-277) */
    {
        const int d = tid;                                      /* blockDim.x == EVD */
        const int chunk = (na + 63) >> 6;
        int16_t *Ti = A.T + (size_t) i * 64 * EVD;
        int res = 0;
        if (base + d < 0 || d >= zwlen) res = zw_cut && base + d >= 0 ? -2 : -1;
        const double taup_rand = res == 0 ? A.taup * zw[d] : 0.0;
        int cur = d + 1, red = 0, cnext = 0, c = 0;
        for (int a = 0; a < na; ++a) {
            if (a == cnext) { if (c < 64) Ti[(size_t) c * EVD + d] = (int16_t) red; ++c; cnext += chunk; }
            if (res != 0) continue;
            if (cur + 1 >= zwlen) { res = zw_cut ? -2 : -1; continue; }
            /* the lane's chain is serial — deviate index -> sigma' (an fp64 exp) -> the first draw inside the box -> next index — and with
             * one wavefront per SIMD nothing hides its latency: the seven LDS reads of a coordinate are issued together, one wait in the
             * chain instead of three.  (Counting the draws with sigma' from v_exp_f32 and a slack around the bounds, exact expressions
             * only for draws inside the slack, was slower — 89 us against 70 us per round: with 5 % of the lanes redrawing every wavefront
             * goes round the draw loop twice, and the three-exit loop's mask bookkeeping costs more than the exp saves;
             * profiles/r04_isres_chain_walk.txt.) */
            /* (z2, the first redraw's deviate, comes with them — the staging area has room for the read past a window's end: with 5 % of the
             * draws outside the box nearly every step of a WAVEFRONT goes round the redraw loop once, and that round no longer waits for LDS) */
            double zs = zw[cur], z1 = zw[cur + 1], z2 = zw[cur + 2], sa = sg[a], sm_ = smax[a], xa = xi[a], l = lo[a], h = hi[a];
            asm volatile("" : "+v"(zs), "+v"(z1), "+v"(z2), "+v"(sa), "+v"(sm_), "+v"(xa), "+v"(l), "+v"(h));   /* (all eight reads issued here: the compiler would sink z1 and the bounds below the exp; the five that do not depend on the stream
             * position read one step ahead instead: 31.5-32.0 against 30.7-31.1 ms per generation, dropped) */
            double s2 = sa * exp(taup_rand + A.tau * zs);
            if (s2 > sm_) s2 = sm_;
            int t = 1;
            double xn = xa + s2 * z1;
            if (xn < l || xn > h) {
                t = 2;
                if (cur + 2 >= zwlen) res = zw_cut ? -2 : -1;
                else {
                    xn = xa + s2 * z2;
                    while (xn < l || xn > h) {
                        ++t;
                        if (cur + t >= zwlen) { res = zw_cut ? -2 : -1; break; }
                        xn = xa + s2 * zw[cur + t];
                    }
                }
            }
            cur += 1 + t; red += t - 1;
        }
        const int e = res != 0 ? res : 1 + 2 * na + red;
        const int e16 = e > 32767 ? -1 : e;
        /* read by another workgroup of this launch (the segment's tail): agent-scope stores, two entries a word */
        const int eo = __shfl_down(e16, 1, 64);
        if (!(d & 1)) ev2_st_agent(reinterpret_cast<uint32_t *>(A.E + (size_t) i * EVD + d), (uint32_t) (e16 & 0xffff) | ((uint32_t) eo << 16));
    }
    }
    ev2_segment_tail(A, i, st0, st1, sm);
}

__global__ __launch_bounds__(256) void ev2_scan_kernel(ev2_args A)
{
    extern __shared__ double sm[];
    ev2_scan_body<false>(A, (int) blockIdx.x, sm);
}

/* ---- chain ------------------------------------------------------------------------------------------------------- */
/* start_0 is exact; start_{i+1} = start_i + E[i][start_i - base_i].  Rounds 3-4 walked that chain through the block with one wavefront
 * (256 dependent look-ups: 33-39 us of a ~125 us round, the E table copied into LDS first).  But a look-up table composes: the
 * deviates the SEGMENT of individuals 16 s .. 16 s + 15 consumes is a function of the start of its first individual alone, and that
 * function can be tabulated for every candidate start by independent lanes (round 5: inside the chain kernel — ONE workgroup copying the
 * block's 128 KB of E into LDS, 37-45 us per round; round 6: by the scan launch itself, ev2_segment_tail above).  What is left here:
 *   B  (one thread)  strings the block's segments together from the block's exact first start, ONE look-up per segment in SG (32 KB in
 *      LDS); a segment that cannot be crossed whole ends the walk where its own table says (SC);
 *   C  (all at once)  every resolved individual's start is its segment's start + SP.
 * The same look-ups in the same tables as the serial walk, hence the same starts (config 3: 7-10 us per round, the round 143 -> 118 us,
 * 34.2 -> 32.3 ms per generation: profiles/r06_isres_segchain_ab.txt). */
__device__ __forceinline__ long long ev2_uniform64(long long v)
{
    const unsigned lo = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) (v & 0xffffffffll));
    const unsigned hi = (unsigned) __builtin_amdgcn_readfirstlane((int) (unsigned) ((unsigned long long) v >> 32));
    return (long long) (((unsigned long long) hi << 32) | lo);
}

__global__ __launch_bounds__(256) void ev2_chain_seg_kernel(ev2_args A)
{
    constexpr int NS = EVMX / EV2_SEG;
    __shared__ uint16_t sG[NS * EVD];
    __shared__ int s_segrel[NS], s_segstop[NS], s_segpos[NS + 1], s_r, s_why, s_end;
    __shared__ long long s_asum[4];
    __shared__ double s_msum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t st0 = A.state[0], st1 = A.state[1], st2 = A.state[2], st10 = A.state[10], st11 = A.state[11];
    const double rho_r = A.rho[2 * A.phase], rho_a = A.rho[2 * A.phase + 1];
    if (st2 || st10) { if (tid == 0) A.state[9] = 0; return; }     /* a skipped round resolves nothing (its scan counted nothing) */
    const int nseg = A.bm / EV2_SEG;
    if (tid < NS) A.segcnt[tid] = 0;                           /* for the next round's scan */
    const int64_t k0 = ev2_uniform64(st0), kend = A.phase == 0 ? A.pop : A.survivors;
    if (k0 >= kend) { if (tid == 0) A.state[9] = 0; return; }
    const long long pos0 = ev2_uniform64(st1);
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(A.SG);
        uint4 *dst = reinterpret_cast<uint4 *>(sG);
        for (int q = tid; q < nseg * EVD / 8; q += 256) dst[q] = src[q];
    }
    if (tid < nseg) {
        long long b;
        const bool stop = ev2_slot_stops(A, tid * EV2_SEG, st0, st1, &b);
        s_segstop[tid] = stop; s_segrel[tid] = stop ? 0 : (int) (b - pos0);
    }
    __syncthreads();
    if (tid == 0) {
        int p = 0, nf = 0, cnt = 0, why = 0;
        for (; nf < nseg; ++nf) {
            s_segpos[nf] = p;
            if (s_segstop[nf]) { why = -11; break; }
            const int d = p - s_segrel[nf];
            if (d < 0 || d >= EVD) { why = -10; break; }
            const int g = sG[nf * EVD + d];
            if (g == 0xffff) {                                       /* the walk ends inside this segment: how far it got */
                const int2 c = A.SC[nf * EVD + d];
                cnt = c.x & 0xff; why = (int) (int8_t) ((c.x >> 8) & 0xff);
                p += c.y;
                break;
            }
            p += g;
        }
        s_r = nf * EV2_SEG + cnt; s_why = nf < nseg ? why : 0; s_end = p;
    }
    __syncthreads();
    const int r = s_r, elast = s_why, pos = s_end;
    long long asum = 0;
    double msum = 0;
    {
        for (int q = tid; q < r; q += EVM) {                   /* (thread t: individuals t and t + 256: the order the redraw statistics have been summed in since round 5) */
            const int sg = q / EV2_SEG, j = q % EV2_SEG;
            const int d = s_segpos[sg] - s_segrel[sg];
            A.ws_start[q] = pos0 + s_segpos[sg] + A.SP[((size_t) sg * EV2_SEG + j) * EVD + d];
            asum += A.ws_nact[q]; msum += A.ws_mu[q];
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { asum += __shfl_xor(asum, m, 64); msum += __shfl_xor(msum, m, 64); }
    if (lane == 0) { s_asum[wave] = asum; s_msum[wave] = msum; }
    __syncthreads();
    if (tid != 0) return;
    asum = s_asum[0] + s_asum[1] + s_asum[2] + s_asum[3];
    msum = (s_msum[0] + s_msum[1]) + (s_msum[2] + s_msum[3]);
    const long long rsum = (long long) pos - r - 2 * asum;     /* = sum over the resolved of (consumed - 1 - 2 mutated) = their redraws */
    A.state[12] = k0;
    A.state[0] = k0 + r;
    A.state[1] = pos0 + pos;
    A.state[9] = r;
    A.state[11] = st11 + 1;
    if (elast == -2) A.state[2] = 1;
    else if (r == 0) A.state[10] = 1;                           /* not even the exactly-started first individual resolved: serial fallback */
#ifdef NLA_EV2_REASONS
    A.state[3] += elast == -10; A.state[4] += elast == -11; A.state[5] += elast == -1; A.state[6] += elast == 0; A.state[7] += r;
#endif
    A.rho[2 * A.phase] = 0.9 * rho_r + (double) rsum;
    A.rho[2 * A.phase + 1] = 0.9 * rho_a + msum;
}

/* ---- write ------------------------------------------------------------------------------------------------------- */
/* (a workgroup of 64 or of 256 threads: the chunk walk is the first wavefront's, loads and stores are everybody's) */
template <bool PH0>
__device__ __forceinline__ void ev2_write_body(const ev2_args &A, const int i, double *sm)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n = A.n, ld = A.ld;
    /* state[9] / [12] were written by the chain kernel of this round; a round that was skipped leaves state[9] = 0 */
    const int64_t st9 = A.state[9], st12 = A.state[12];
    if (i >= st9) return;
    const int na = PH0 ? n : A.ws_nact[i];
    const int64_t start = A.ws_start[i], wbase = A.ws_base[i];
    const int64_t k = st12 + i, rk = A.irank[k];
    double *xi = sm, *sg = sm + n, *lo = sm + 2 * n, *hi = sm + 3 * n, *smax = sm + 4 * n, *xo = sm + 5 * n, *so = sm + 6 * n, *zw = sm + 7 * n;
    const double sqn = sqrt((double) n);
    const int32_t *act = A.ws_act + (size_t) i * n;
    const double *wpre = A.ws_xpre + (size_t) i * n;
    if (PH0) {
        const int64_t ri = A.irank[k % A.survivors];
        const double *xr = A.X + (size_t) ri * ld, *sr = A.S + (size_t) ri * ld;
        for (int a = tid; a < na; a += nthr) { xi[a] = xr[a]; sg[a] = sr[a]; lo[a] = A.lb[a]; hi[a] = A.ub[a]; smax[a] = (A.ub[a] - A.lb[a]) / sqn; }
    } else {
        const double *wxi = A.ws_xi + (size_t) i * n, *wsg = A.ws_sg + (size_t) i * n;
        for (int a = tid; a < na; a += nthr) {
            const int j = act[a];
            xi[a] = wxi[a]; sg[a] = wsg[a]; lo[a] = A.lb[j]; hi[a] = A.ub[j]; smax[a] = (A.ub[j] - A.lb[j]) / sqn;
        }
    }
    const int ZW = EVD + 3 * na + 65;                         /* as long as the longest window the scan had for this individual */
    const int64_t avail = A.zcount - start;
    const int zwlen = (int) (avail < ZW ? avail : ZW);
    for (int q = tid; q < zwlen; q += nthr) zw[q] = A.z[start + q];
    __syncthreads();
    if (tid < 64) {
        /* the exact start is candidate dtrue of the scan's window: its lane left the redraw count before every chunk in T */
        const int lane = tid;
        const double ALPHA = 0.2;
        const int dtrue = (int) (start - wbase);
        const int chunk = (na + 63) >> 6;
        const int a0 = lane * chunk < na ? lane * chunk : na, a1 = a0 + chunk < na ? a0 + chunk : na;
        const double taup_rand = A.taup * zw[0];
        int cur = 1 + 2 * a0 + (a0 < na ? (int) A.T[((size_t) i * 64 + lane) * EVD + dtrue] : 0);
        for (int a = a0; a < a1; ++a) {
            const double xa = xi[a], sa = sg[a], l = lo[a], h = hi[a];
            double s2 = sa * exp(taup_rand + A.tau * zw[cur]);
            if (s2 > smax[a]) s2 = smax[a];
            int t = 1;
            double xn;
            for (;;) { xn = xa + s2 * zw[cur + t]; if (!(xn < l || xn > h)) break; ++t; }
            xo[a] = xn; so[a] = sa + ALPHA * (s2 - sa);
            cur += 1 + t;
        }
    }
    __syncthreads();
    double *xw = A.X + (size_t) rk * ld, *sw = A.S + (size_t) rk * ld;
    if (PH0) {
        for (int a = tid; a < na; a += nthr) { xw[a] = xo[a]; sw[a] = so[a]; }
    } else {
        for (int j = tid; j < n; j += nthr) xw[j] = wpre[j];      /* coordinates that stayed inside the box; sigma unchanged */
        __syncthreads();
        for (int a = tid; a < na; a += nthr) { const int j = act[a]; xw[j] = xo[a]; sw[j] = so[a]; }
    }
}

__global__ __launch_bounds__(64) void ev2_write_kernel(ev2_args A)
{
    extern __shared__ double sm[];
    ev2_write_body<false>(A, (int) blockIdx.x, sm);
}
__global__ __launch_bounds__(256) void ev2_write0_kernel(ev2_args A)
{
    extern __shared__ double sm[];
    ev2_write_body<true>(A, (int) blockIdx.x, sm);
}

/* Mutation phase, one launch per round in front of the chain kernel: workgroups 0 .. bm-1 scan the round's block (A: this round's
 * tables), workgroups bm .. 2 bm-1 — only when a round went before it in the batch — WRITE the individuals the PREVIOUS round
 * resolved (P: that round's tables; T, the window origins and the exact starts exist twice and alternate).  The children's rows and
 * the parents' rows are disjoint in this phase, so the previous round's write needs nothing the scan touches and leaves the serial
 * path: a round is scan -> chain instead of stage -> scan -> chain -> write (round 5; 4.8 + 8.1 us of kernels and two launch gaps of
 * a ~125 us round). */
__global__ __launch_bounds__(256) void ev2_scan0_kernel(ev2_args A, ev2_args P)
{
    extern __shared__ double sm[];
    /* (the launch takes what the scan and the stage took together — 84 us at config 3, of which the write workgroups cost nothing
     * measurable: a build without them ran the same; profiles/r05_isres_handoff.txt) */
    if ((int) blockIdx.x < A.bm) ev2_scan_body<true>(A, (int) blockIdx.x, sm);
    else ev2_write_body<true>(P, (int) blockIdx.x - A.bm, sm);
}

/* redraws a child of parent p (by rank position, p < survivors) is EXPECTED to make (isres.c:245-248 draws x again while it is outside
 * the box): with sigma' ~ sigma a draw of coordinate j leaves the box with p_j = Q((x_j - lb_j) / sigma_j) + Q((ub_j - x_j) / sigma_j), and
 * the draws are repeated p_j / (1 - p_j) times on average.  The parents differ (sd of the redraws per individual 5.8 at config 3, 5.1 of it
 * NOT explained by this expectation = the draws' own noise), and over the 256 individuals of a round a single rate for all of them
 * misses the true start by +-90 deviates: half of the rounds ended where the start left its +-128 window.  With the per-parent
 * expectation: 226 -> 183 rounds for the 42 857 children of a generation (tools/evolve_predict.py on a dump of the emulated device
 * reproduces the device's 226; 168 = every round a full block).  Once per generation, before the mutation phase. */
__global__ __launch_bounds__(256) void ev2_parent_mu_kernel(int n, int ld, int64_t survivors, const double *__restrict__ lb, const double *__restrict__ ub,
                                                             const int32_t *__restrict__ irank, const double *__restrict__ X, const double *__restrict__ S,
                                                             double *__restrict__ mu_rp)
{
    __shared__ double s_mu[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t p = blockIdx.x;
    if (p >= survivors) return;
    const int64_t ri = irank[p];
    const double *xr = X + (size_t) ri * ld, *sr = S + (size_t) ri * ld;
    double mu = 0;
    for (int j = tid; j < n; j += 256) {
        const double xv = xr[j], sv = sr[j];
        const double inv = 0.7071067811865476 / (sv > 1e-300 ? sv : 1e-300);
        const double q = 0.5 * (erfc((xv - lb[j]) * inv) + erfc((ub[j] - xv) * inv));
        mu += q < 0.999 ? q / (1.0 - q) : 999.0;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mu += __shfl_xor(mu, m, 64);
    if (lane == 0) s_mu[wave] = mu;
    __syncthreads();
    if (tid == 0) mu_rp[p] = s_mu[0] + s_mu[1] + s_mu[2] + s_mu[3];
}

/* inverse of the ranking permutation (variation's dependency test) */
__global__ __launch_bounds__(256) void ev2_inverse_kernel(int64_t pop, const int32_t *__restrict__ irank, int32_t *__restrict__ inv)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k < pop) inv[irank[k]] = (int32_t) k;
}

/* ---- launchers --------------------------------------------------------------------------------------------------- */
extern "C" size_t nla_isres_evolve2_ws_bytes(int n)
{
    /* nact | act | xi | sg | xpre | E | T | base | start, each region 256-byte aligned */
    size_t b = 0;
    auto add = [&](size_t x) { b += (x + 255) & ~(size_t) 255; };
    add(sizeof(int32_t) * EVMX); add(sizeof(int32_t) * EVMX * (size_t) n);
    add(sizeof(double) * EVMX * (size_t) n); add(sizeof(double) * EVMX * (size_t) n); add(sizeof(double) * EVMX * (size_t) n);
    add(sizeof(int16_t) * EVMX * EVD); add(sizeof(int16_t) * EVMX * 64 * EVD); add(sizeof(int64_t) * EVMX); add(sizeof(int64_t) * EVMX);
    add(sizeof(double) * EVMX);
    add(sizeof(int16_t) * EVMX * 64 * EVD); add(sizeof(int64_t) * EVMX); add(sizeof(int64_t) * EVMX);      /* the second set of T / base / start (mutation phase) */
    add(sizeof(uint32_t) * (EVMX / EV2_SEG)); add(sizeof(uint16_t) * (EVMX / EV2_SEG) * EVD); add(sizeof(int2) * (EVMX / EV2_SEG) * EVD);
    add(sizeof(int32_t) * (EVMX / EV2_SEG) * EVD * EV2_SEG);                                                /* the segment tables */
    return b;
}
extern "C" int nla_isres_evolve2_supported(int n) { return n >= 1 && n <= EV2_MAXN; }

extern "C" int nla_k_isres_inverse(int64_t pop, const int32_t *irank, int32_t *inv, void *stream)
{
    if (pop <= 0) return 0;
    hipLaunchKernelGGL(ev2_inverse_kernel, dim3((unsigned) ((pop + 255) / 256)), dim3(256), 0, (hipStream_t) stream, pop, irank, inv);
    NLA_LAUNCH_CHECK();
    return 0;
}

/* once per generation, before the mutation phase's rounds: mu_rp[p], p < survivors (device memory, survivors doubles) */
extern "C" int nla_k_isres_evolve_parent_mu(int n, int ld, int64_t survivors, const double *lb, const double *ub, const int32_t *irank,
                                            const double *X, const double *S, double *mu_rp, void *stream)
{
    if (survivors <= 0) return 0;
    hipLaunchKernelGGL(ev2_parent_mu_kernel, dim3((unsigned) survivors), dim3(256), 0, (hipStream_t) stream, n, ld, survivors, lb, ub, irank, X, S, mu_rp);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_evolve_rounds(int n, int ld, int phase, int64_t pop, int64_t survivors, int64_t zcount, double taup, double tau,
                                            const double *lb, const double *ub, const double *z, const int32_t *irank, const int32_t *inv,
                                            double *X, double *S, const double *x0c, int64_t *state, double *rho, void *ws, const double *mu_rp,
                                            int rounds, void *stream)
{
    if (!nla_isres_evolve2_supported(n)) return (int) hipErrorInvalidValue;
    if (phase == 0 && !mu_rp) return (int) hipErrorInvalidValue;
    ev2_args A;
    A.n = n; A.ld = ld; A.phase = phase; A.pop = pop; A.survivors = survivors; A.zcount = zcount; A.taup = taup; A.tau = tau;
    A.lb = lb; A.ub = ub; A.z = z; A.irank = irank; A.inv = inv; A.X = X; A.S = S; A.x0c = x0c; A.state = state; A.rho = rho; A.mu_rp = mu_rp;
    char *p = (char *) ws;
    auto take = [&](size_t x) { char *q = p; p += (x + 255) & ~(size_t) 255; return q; };
    A.ws_nact = (int32_t *) take(sizeof(int32_t) * EVMX);
    A.ws_act = (int32_t *) take(sizeof(int32_t) * EVMX * (size_t) n);
    A.ws_xi = (double *) take(sizeof(double) * EVMX * (size_t) n);
    A.ws_sg = (double *) take(sizeof(double) * EVMX * (size_t) n);
    A.ws_xpre = (double *) take(sizeof(double) * EVMX * (size_t) n);
    A.E = (int16_t *) take(sizeof(int16_t) * EVMX * EVD);
    A.T = (int16_t *) take(sizeof(int16_t) * EVMX * 64 * EVD);
    A.ws_base = (int64_t *) take(sizeof(int64_t) * EVMX);
    A.ws_start = (int64_t *) take(sizeof(int64_t) * EVMX);
    A.ws_mu = (double *) take(sizeof(double) * EVMX);
    int16_t *T2 = (int16_t *) take(sizeof(int16_t) * EVMX * 64 * EVD);
    int64_t *base2 = (int64_t *) take(sizeof(int64_t) * EVMX), *start2 = (int64_t *) take(sizeof(int64_t) * EVMX);
    A.segcnt = (uint32_t *) take(sizeof(uint32_t) * (EVMX / EV2_SEG));
    A.SG = (uint16_t *) take(sizeof(uint16_t) * (EVMX / EV2_SEG) * EVD);
    A.SC = (int2 *) take(sizeof(int2) * (EVMX / EV2_SEG) * EVD);
    A.SP = (int32_t *) take(sizeof(int32_t) * (EVMX / EV2_SEG) * EVD * EV2_SEG);
    hipStream_t st = (hipStream_t) stream;
    if (hipMemsetAsync(A.segcnt, 0, sizeof(uint32_t) * (EVMX / EV2_SEG), st) != hipSuccess) return (int) hipGetLastError();
    ev2_args B = A;                                            /* the other set of what a round's write still needs while the next round scans */
    B.T = T2; B.ws_base = base2; B.ws_start = start2;
    const size_t lds_tail = sizeof(int16_t) * EV2_SEG * EVD;   /* (the segment tail of a scan workgroup re-uses the launch's dynamic LDS) */
    const size_t lds_scan = std::max(sizeof(double) * (size_t) (5 * n + EV2_ZW(n) + EV2_ZPAD), lds_tail);
    const size_t lds_write = std::max(sizeof(double) * (size_t) (7 * n + EV2_ZW(n) + EV2_ZPAD), lds_tail);
    static bool attr_set = false;
    if (!attr_set) {
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(ev2_scan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(ev2_scan0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(ev2_write_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(ev2_write0_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void) hipGetLastError();
        attr_set = true;
    }
    A.bm = B.bm = phase == 0 ? EV2_MUT_BLOCK : EV2_VAR_BLOCK;
    if (phase == 0) {
        /* round r works on set r & 1; its launch also writes what round r - 1 resolved (the other set); the batch ends with the last
         * round's write, so every batch starts from a population that is up to date */
        for (int r = 0; r < rounds; ++r) {
            const ev2_args &C = (r & 1) ? B : A, &Pv = (r & 1) ? A : B;
            hipLaunchKernelGGL(ev2_scan0_kernel, dim3(r ? 2 * EV2_MUT_BLOCK : EV2_MUT_BLOCK), dim3(EVD), lds_write, st, C, Pv);
            hipLaunchKernelGGL(ev2_chain_seg_kernel, dim3(1), dim3(256), 0, st, C);
        }
        if (rounds > 0) hipLaunchKernelGGL(ev2_write0_kernel, dim3(EV2_MUT_BLOCK), dim3(256), lds_write, st, ((rounds - 1) & 1) ? B : A);
    } else
    for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(ev2_stage_kernel, dim3(EV2_VAR_BLOCK), dim3(256), 0, st, A);
        hipLaunchKernelGGL(ev2_scan_kernel, dim3(EV2_VAR_BLOCK), dim3(EVD), lds_scan, st, A);
        hipLaunchKernelGGL(ev2_chain_seg_kernel, dim3(1), dim3(256), 0, st, A);
        hipLaunchKernelGGL(ev2_write_kernel, dim3(EV2_VAR_BLOCK), dim3(64), lds_write, st, A);
    }
    NLA_LAUNCH_CHECK();
    return 0;
}
