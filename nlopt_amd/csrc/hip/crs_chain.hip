/* crs_chain.hip — CRS2_LM: a whole window of trials with the accept/reject chain RESOLVED INSIDE THE LAUNCH.
 *
 * The reference is one serial chain (src/algs/crs/crs.c:125-156): trial point = best + sum of n random rows, accept it over
 * the current worst row or not, maybe one mutation, next.  Trial b+1 may sample a row that trial b has just overwritten, so
 * the chain is a dependence chain through the population.  crs_kernels.hip breaks it by NEVER reading a row that may still
 * change (a slot stops at its first pick among the d worst rows, d = its distance from the window front): exact, nothing
 * wasted — but only the first ~sqrt(2N/n) slots of a window get through, 6 per pass at n = 4096, N = 1e5, and every pass
 * pays its fixed costs and the latency floor of a lone trial.  Speculating "every block is accepted and overwrites the k-th
 * worst row" lets a slot take a hazard row from the producer's trial point, but each rejection (3.6 % of the blocks) shifts
 * which block writes which row for everything behind it: measured 22.6 k evals/s against 31.1 k (half the bytes recomputed).
 *
 * Here the dependences are resolved exactly, by the kernel itself:
 *   gather     grid = K slots x coordinate chunks, as in crs_advance_kernel (same tiling, same accumulation order: x is the
 *              reference's bit for bit).  Workgroups draw (slot, chunk) from a ticket counter, front slot first.
 *   evaluate   the workgroup that completes the LAST chunk of a slot evaluates f of the trial point, forms the mutation the
 *              reference would try after a rejection (crs.c:139-146, words of the next stream block) and evaluates it too.
 *   resolve    ONE DEDICATED WAVEFRONT (the holder of ticket 0; crs_chain_resolver.h) advances the chain as far as the evaluated
 *              slots reach, in block order, out of its registers: f(T) < f(current worst)? else f(M) < f(worst)? — crs_trial's
 *              decisions on the window's list of worst rows (their f values come with the launch; a new value that lands
 *              among them is tracked) — and publishes, per worst row, who overwrote it (slot, trial or mutation) and how far
 *              the chain has got.  (Rounds 2-4 had the evaluating workgroups do this under a lock: five dependent round
 *              trips per slot; measured side by side on the MI355X in round 5 — n = 4096: 43.7 k -> 45.7 k evals/s, n = 512:
 *              316 k -> 676 k, profiles/r05_staged_ab.txt — and deleted.)
 *   new best   a trial that becomes the new best point ends the window: every later slot started its sum from the old best
 *              row (crs.c:69).  The resolver publishes the slot in ctrl->halt; workgroups that draw a ticket for a later
 *              slot after that leave at once (their slot's status says "not computed"), so the launch ends as soon as the
 *              ~8 slots in flight have drained instead of gathering the rest of the window for nothing.
 *   consume    a slot whose next pick is one of the worst rows ahead of it waits until that row's fate is known: overwritten
 *              by an earlier block -> read the writer's point (TX / TM of that slot, final before it was published);
 *              chain already past this slot's predecessors without touching the row -> read the row itself.
 * Every slot finishes in the one launch and nothing is guessed, so (nearly) every slot is consumed: the window can be as wide
 * as the per-pass costs want (up to 256) and the gather runs at its saturated bandwidth.  What remains unused: the slot of a
 * block that turned out to be a mutation block (its predecessor was rejected), and whatever is in flight when the best point
 * changes (the sums start from the best row; the resolver halts there).
 *
 * The host stays the authority: its in-order walk (crs_driver.c) recomputes every decision from the f values and accepts a
 * slot only if each row it took from a producer was in fact last written by that producer's point — the device's resolution
 * is a prediction that is right unless something the device does not model intervenes (a stop, a value re-entering the
 * worst rows twice); then the slot is recomputed.  Deadlock cannot occur: a slot waits only for blocks before it, their
 * workgroups hold earlier tickets and are therefore running or done, and the front slot never waits.
 *
 * Roofline: HBM, 8 n (n+1) algorithmic bytes per trial; the evaluation adds 24 n.
 *
 * Measured on MI355X (n = 4096, N = 1e5, Griewank; gpurun_out/r02g, r02h): 28 us per slot at 48 slots per launch (4.8 TB/s),
 * 32 us at 128, 40 us at 256 — a slot deeper in the window has more picks among the worst rows ahead of it (0.04 per row) and
 * each of them is a wait; 34.6 k evals/s at 48 slots against 31.0 k for the conservative passes of crs_kernels.hip.
 * TX, TM and the control block are UNCACHED device memory (nla_dev_malloc_uncached): what one workgroup stores another loads
 * without the L2 write-back / invalidate an agent-scope release / acquire pair costs per chunk and per pick (+5 %: 36.4 k).
 * Tried and dropped: agent-coherent (sc1) loads of the forwarded rows, a back-off in the polling loops, 8x larger Vitter batches
 * (the digest kernel runs under the gather on a quarter of the CUs: +2 %, kept), polling intervals 4 / 32 / 127 (no change).
 * What does matter: no scratch memory.  The worst-row lists travel as a 1.5 KB by-value kernel argument; indexing that argument
 * by name with a run-time j makes the compiler copy it to private memory — depending on innocuous details of the surrounding
 * control flow — and 1.5 KB of scratch per lane costs 45 % (fewer workgroups resident).  The kernel therefore reads the lists
 * through the kernarg segment pointer (first argument, offset 0) and never by name.
 * A waiting slot also stops waiting when the chain can no longer reach its row (see the polling loop): the full serialisation
 * behind a pick that nobody overwrites was the larger part of the wait time at a 9 % rejection rate.
 */
#include "crs_common.h"
#include "../nla_switches.h"
#include <stdlib.h>
#include "../../../include/nlopt_amd.h"

#define NLA_CHAIN_LIGHT_BELOW 2048        /* below: lighter fences around the resolver's turn and the evaluation (crs_chain_resolver.h, ch_landed) */
#define CH_EXTRA 32                      /* accepted values that landed among the window's worst rows */
#define CH_FWAVES 8                      /* f is reduced as a workgroup of 8 wavefronts reduces it (= NLA_FIN_WAVES of crs_kernels.hip, SH_WAVES of crs_shard.hip):
                                          * windows, conservative passes and column-sharded jobs give the same f bit for bit at every n */
#include <stddef.h>
#include <string.h>

/* control block of one launch (device memory, zeroed before the launch except `ticket`, which only grows) */
struct chain_ctrl {
    uint32_t ticket, lock, next, halt, naccept, wp, nextra, pk;     /* pk = next | (next - wp) << 16: what a waiting slot polls */
    double xf[CH_EXTRA]; int64_t xrow[CH_EXTRA];
};
/* behind the control block: fv[2K] doubles (fT, fM of every slot, for the resolver — the status records themselves may live in
 * pinned host memory), then the u32 arrays done[K], evald[K], rowstate[nW] */

/* the fv area holds one 16-byte record per slot (crs_chain_resolver.h: ~bits(fT), ~bits(fM)); evald is unused (the lock version's flag) */
#include "crs_chain_resolver.h"
static_assert(offsetof(chain_ctrl, next) == 4 * CH_CTRL_NEXT && offsetof(chain_ctrl, halt) == 4 * CH_CTRL_HALT && offsetof(chain_ctrl, naccept) == 4 * CH_CTRL_NACCEPT &&
              offsetof(chain_ctrl, wp) == 4 * CH_CTRL_WP && offsetof(chain_ctrl, nextra) == 4 * CH_CTRL_NEXTRA && offsetof(chain_ctrl, pk) == 4 * CH_CTRL_PK &&
              sizeof(chain_ctrl) % 16 == 0, "crs_chain_resolver.h addresses the control block by word");

#define NLA_KA_MAX 128                   /* list length that still travels as kernel arguments (2 KB of the 4 KB there are) */
struct chain_lists { int inl; int64_t W[NLA_KA_MAX]; double Wf[NLA_KA_MAX]; };

__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
/* system scope (sc0 sc1): what crosses to / comes from ANOTHER device or process — the column-sharded instance below */
__device__ __forceinline__ uint32_t ld_sys(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys_f64(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<uint64_t *>(p), (uint64_t) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* ---- the window on a COLUMN-SHARDED population (SH = true): W ranks = W devices (or, on a test box, W processes sharing one), rank r
 * holds columns [c0, c0 + nc) of every row.  Coordinates never mix in the gather-sum (crs.c:63-121), so a rank's workgroups form ITS
 * columns of every slot's trial point — 1 / W of the rows' bytes — and store the finished chunk into the slot's row of EVERY rank's TX
 * (peer-mapped memory: hipIpcOpenMemHandle; over xGMI between devices), then raise the chunk's flag word on every rank.  TX rows are
 * full width on every rank.  The workgroup that completes a slot's LOCAL chunks waits for the flags of all the slot's chunks, then
 * evaluates f(T), forms the mutation from the whole best row (a full-width copy every rank keeps) and evaluates it — the single-GPU
 * reduction on the same numbers, so f is the same bits on every rank — and every rank's resolver takes the same decisions.  Nothing
 * else crosses: 8 n bytes per slot and rank against 8 n (n + 1) / W gathered.
 *   flags[a][c]      (seq << 2): chunk c (numbered across the ranks) of slot a of launch `seq` is in this rank's TX.  Never cleared: a
 *                    waiter accepts any value >= its own launch's (a peer may be one launch ahead, never two: it cannot finish a
 *                    launch without this rank's chunks)
 *   stopw[seq & 1][r] (seq << 2) | rank r's stop bits for launch seq (force_stop, maxtime: per-process conditions every rank must act
 *                    on together); the resolver's workgroup writes this rank's word to every peer when the launch starts and ORs
 *                    all ranks' words into status[K] when the chain is done
 * Every wait gives up: a slot behind a new best point is abandoned (ctrl->halt), and a chunk that does not arrive within 1.5 s sets
 * the launch's failure word (status[K].t) — the host ends the run with an error instead of hanging the device. */
#define NLA_SH_MAXW 8
struct chain_shard {
    int world, rank, c0, ldf, chunks_total, chunk0, pad0, pad1;
    double *peerTX[NLA_SH_MAXW];
    uint32_t *peerflags[NLA_SH_MAXW];
    uint32_t *peerstop[NLA_SH_MAXW];
    const double *xbest, *lbf, *ubf;
};
#define NLA_SH_WAIT_TICKS 150000000ull   /* 1.5 s of the 100 MHz clock (the resolver's own give-up is 2 s after the last arrival) */

template <int VEC, int U, int WAVES, int OBJ, bool SH = false>
__global__ __launch_bounds__(WAVES * 64) void crs_chain_kernel(
    const chain_lists L_first_kernel_argument,  /* read through the kernarg segment below, never by name: indexing the by-value copy
                                                 * with a run-time j makes the compiler move all 1.5 KB of it to scratch memory */
    int n, int ld, const double *__restrict__ X, int64_t i0, double f_best, const int32_t *__restrict__ jn_ring,
    const int32_t *__restrict__ pos_ring, const int32_t *__restrict__ last_ring, const uint32_t *__restrict__ words_ring,
    uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *__restrict__ Wd, const double *__restrict__ Wfd, int nW,
    int slot_mask, int chunks, const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ TX,
    double *__restrict__ TM, chain_ctrl *__restrict__ ctrl, uint32_t ticket_base, nla_crs_slot_status *__restrict__ status,
    uint32_t *__restrict__ fwcnt, uint32_t *__restrict__ fwrec, int fwcap, double sign, uint64_t resolver_timeout,
    int ncols, const chain_shard *__restrict__ S, uint32_t seq, uint32_t stopbits)
{
    typedef const __attribute__((address_space(4))) chain_lists *kernarg_lists;
    const kernarg_lists Lk = (kernarg_lists) __builtin_amdgcn_kernarg_segment_ptr();       /* explicit arguments start at offset 0 */
    (void) L_first_kernel_argument;
    typedef typename VecT<VEC>::T V;
    static_assert(U <= 64, "one lane per row of a batch");
    __shared__ V sacc[64];
    __shared__ int32_t srow[NLA_ADV_RCAP];
    __shared__ int s_turn, s_ticket, s_last, s_state;
    __shared__ uint32_t s_nrec, s_halt;
    __shared__ double scratch[2 * CH_FWAVES];
    volatile __attribute__((address_space(3))) int *turn = (volatile __attribute__((address_space(3))) int *) &s_turn;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    double *fv = reinterpret_cast<double *>(ctrl + 1);
    uint32_t *done = reinterpret_cast<uint32_t *>(fv + 2 * (size_t) K), *rowstate = done + 2 * (size_t) K;     /* (done[K], K unused words, rowstate[nW]) */
    const int64_t *W = Lk->inl ? (const int64_t *) Lk->W : Wd;
    const double *Wf = Lk->inl ? (const double *) Lk->Wf : Wfd;
    /* SH: a workgroup takes ticket after ticket until none is left (the launcher caps the grid: ranks that SHARE a device must both be
     * resident — a rank's waiting workgroups may not fill the chip — and with the tickets handed out in order the holder of the earliest
     * unfinished one is always running, whatever the cap).  Otherwise: one ticket per workgroup, as ever. */
#define CH_NEXT do { if constexpr (SH) continue; else return; } while (0)
    for (;;) {
    if constexpr (SH) __syncthreads();    /* the previous ticket's LDS contents are done with */
    if (threadIdx.x == 0) { s_ticket = (int) (atomicAdd(&ctrl->ticket, 1u) - ticket_base); s_nrec = 0; s_halt = ld_agent(&ctrl->halt); }
    __syncthreads();
    if constexpr (SH) { if (s_ticket > K * chunks) return; }
    /* the first workgroup to run is the resolver: wavefront 0 advances the chain for the whole launch, the others leave.  Every
     * slot's workgroups hold later tickets, so whatever they wait for is running already */
    if (s_ticket == 0) {
        if (wave == 0) {
            if constexpr (SH) {          /* this rank's stop bits for the launch, to every rank (lane r: rank r) */
                if (lane < S->world) st_sys(S->peerstop[lane] + (seq & 1u) * NLA_SH_MAXW + (uint32_t) S->rank, (seq << 2) | (stopbits & 3u));
            }
            chain_resolver_wave(reinterpret_cast<uint32_t *>(ctrl), reinterpret_cast<const uint64_t *>(fv), rowstate, K, nW, W, Wf, f_best, i0,
                                resolver_timeout, !SH && n < NLA_CHAIN_LIGHT_BELOW);
            if constexpr (SH) {
                /* the chain is done, so every rank's kernel of this launch has started (slot 0 is never abandoned: its chunks came from
                 * all of them): their stop words are there or on their way */
                uint32_t bits = 0, bad = 0;
                if (lane < S->world) {
                    const uint32_t *wp_ = S->peerstop[S->rank] + (seq & 1u) * NLA_SH_MAXW + (uint32_t) lane;
                    const uint64_t t0 = wall_clock64();
                    uint32_t v;
                    while (((v = ld_sys(wp_)) >> 2) != seq) {
                        if (wall_clock64() - t0 > NLA_SH_WAIT_TICKS) { bad = 1; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    bits = v & 3u;
                }
                const uint32_t forced = __ballot((bits & 1u) != 0) != 0, timed = __ballot((bits & 2u) != 0) != 0;
                const uint32_t failed = (__ballot(bad != 0) != 0 || ld_agent(&ctrl->lock) != 0) ? 1u : 0u;
                if (lane == 0) { status[K].fT = forced ? 1. : 0.; status[K].fM = timed ? 1. : 0.; status[K].t = (int32_t) failed; status[K].pad = 0; }
            }
        }
        return;
    }
    const int wg = s_ticket - 1;
    const int a = wg / chunks, chunk = wg % chunks;             /* front slot first: producers before consumers */
    {
        /* a new best point at slot j ended the window (ctrl->halt = 2 | (j + 1) << 8): this slot started from the old best row and
         * will be dropped — do not gather it.  (read once, by the thread that drew the ticket: the whole workgroup stays or leaves) */
        const uint32_t h = s_halt;
        if ((h & 2u) && (uint32_t) a >= (h >> 8)) {
            if (threadIdx.x == 0) { status[a].t = 0; if (chunk == 0) fwcnt[a] = 0; }
            CH_NEXT;
        }
    }
    const uint64_t block = first_block + (uint64_t) a;
    const uint32_t rb = (uint32_t) (block % ring_blocks);
    const int q = (int) (block & (uint64_t) slot_mask);
    const int32_t *p = pos_ring + (size_t) rb * (size_t) n;
    const int jn = jn_ring[rb];
    const int64_t rbase = p[n - 1];                             /* last pick: i += iurand(Nleft); i += i == i0  (crs.c:109) */
    int64_t al = rbase + (rbase >= i0 ? 1 : 0) + (int64_t) last_ring[rb];
    al += (al == i0) ? 1 : 0;
    auto pick_row = [&](int t) -> int32_t {
        int64_t r;
        if (t < n - 1) { r = p[t]; r += (r >= i0 ? 1 : 0); } else r = al;
        return (int32_t) r;
    };
    const int nun = a < nW ? a : nW;
    /* the picks of the staged segment that are rows W[j], j < nun, become -(j+1): "ask what happened to worst row j" */
    auto mark_hazards = [&](int cnt) {
        int hit[(256 + 63) / 64];
#pragma unroll
        for (int it = 0; it < (256 + 63) / 64; ++it) {
            const int j = it * 64 + lane;
            hit[it] = -1;
            if (j < nun) {
                const int64_t r = W[j];
                int lo = 0, hi = cnt - 1;
                while (r != i0 && lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int32_t pv = srow[mid];
                    if (pv == (int32_t) r) { hit[it] = mid; break; }
                    if (pv < (int32_t) r) lo = mid + 1; else hi = mid - 1;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < (256 + 63) / 64; ++it)
            if (hit[it] >= 0) srow[hit[it]] = -(it * 64 + lane + 1);
    };
    const int col = (chunk * 64 + lane) * VEC;
    const bool active = col < (SH ? ncols : n);
    const size_t colc = active ? (size_t) col : 0;
    const double *Xc = X + colc;
    /* rows of TX / TM: SH — full width (stride ldf), this rank's columns start at c0; otherwise the population's own layout */
    const size_t tld = SH ? (size_t) S->ldf : (size_t) ld;
    const size_t tc0 = SH ? (size_t) S->c0 : 0;
    double *accrow = TX + (size_t) q * tld + tc0 + colc;
    if (wave == 0) sacc[lane] = ldv<VEC>(Xc + (size_t) i0 * (size_t) ld);       /* x := best (crs.c:69) */
    const double hneg = -(0.5 * n);         /* x -= xi*(0.5*n)  ==  x += xi*(-(0.5*n)), exactly */
    const uint32_t lane_off = (uint32_t) (colc * sizeof(double));
    V v[U];
    for (int seg0 = 0; seg0 < n; seg0 += NLA_ADV_RCAP) {
        const int cnt = (n - seg0 < NLA_ADV_RCAP) ? n - seg0 : NLA_ADV_RCAP;
        nla_lds_barrier();
        for (int i = threadIdx.x; i < cnt; i += WAVES * 64) srow[i] = pick_row(seg0 + i);
        nla_lds_barrier();
        if (wave == 0 && nun > 0) mark_hazards(cnt);
        if (threadIdx.x == 0) *turn = 0;
        nla_lds_barrier();
        const int nb = (cnt + U - 1) / U;
        auto issue = [&](int b) {
            const int base = b * U;
            const int mine = base + lane < cnt ? base + lane : cnt - 1;
            const int32_t myrow = srow[mine];
#pragma unroll
            for (int u = 0; u < U; ++u) {        /* unconditional: lanes past the end hold the last row */
                const int64_t r = (int64_t) __builtin_amdgcn_readlane(myrow, u);
                const char *rowp = reinterpret_cast<const char *>(X + (size_t) r * (size_t) ld);   /* wave-uniform */
                if (r < 0) {
                    /* worst row j of the launch's list: overwritten by a block before this one?  Wait until the chain says so
                     * (rowstate[j]) or has passed every block before this slot without touching it */
                    const int j = (int) (-r - 1);
                    uint32_t rs;
                    for (;;) {
                        rs = ld_agent(&rowstate[j]);
                        if (rs) break;
                        /* ... or cannot touch it any more: rows are overwritten in list order, one per accepted block at most, so
                         * with wp rows gone and a - next blocks still to be decided before this slot, row j >= wp + (a - next)
                         * stays as it is */
                        const uint32_t pk = ld_agent(&ctrl->pk);         /* next | (next - wp) << 16, one word */
                        /* next >= a  or  next - wp >= a - j, as one branch: both differences negative <=> keep waiting */
                        if ((int32_t) (((pk & 0xffffu) - (uint32_t) a) & ((pk >> 16) - (uint32_t) (a - j))) >= 0) { rs = ld_agent(&rowstate[j]); break; }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    /* TX / TM are uncached memory (nla_dev_malloc_uncached): nothing stale to drop, so no agent-scope acquire
                     * (= an L2 invalidate per pick) here.  The wavefront-scope fence orders the loads below after the poll */
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    /* a writer at or behind this slot's own block does not count: the row is read as it is before that */
                    int pj = 0, kind = 0;                    /* kind 0: the row as it is (no block before this one wrote it) */
                    if (rs && (int) (rs >> 3) < a) {
                        pj = (int) (rs >> 3); kind = (int) ((rs >> 1) & 3u);
                        const int qk = (int) ((first_block + (uint64_t) pj) & (uint64_t) slot_mask);
                        rowp = reinterpret_cast<const char *>((kind == 1 ? TX : TM) + (size_t) qk * tld + tc0);
                    } else rowp = reinterpret_cast<const char *>(X + (size_t) W[j] * (size_t) ld);
                    if (chunk == 0 && lane == 0 && base + u < cnt) {     /* what was decided, for the host to verify */
                        const uint32_t k = atomicAdd(&s_nrec, 1u);
                        if ((int) k < fwcap) fwrec[(size_t) a * (size_t) fwcap + k] = (uint32_t) j | ((uint32_t) pj << 8) | ((uint32_t) kind << 16);
                    }
                }
                v[u] = *reinterpret_cast<const V *>(rowp + lane_off);    /* (non-temporal loads here: +0.8 % at n = 4096, -0.7 % at 2048, -1.5 % at 512, same box — left plain) */
            }
        };
        if (wave < nb) issue(wave);
        for (int ph = wave; ph < nb; ph += WAVES) {
            while (*turn != ph) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            V acc = sacc[lane];
            const int base = ph * U, tb = seg0 + base;
            if (base + U <= cnt && !(jn >= tb && jn < tb + U)) {
#pragma unroll
                for (int u = 0; u < U; ++u) add_row(acc, v[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + u < cnt) acc_row(acc, v[u], (tb + u == jn) ? hneg : 1.0);
            }
            sacc[lane] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *turn = ph + 1;
            if (ph + WAVES < nb) issue(ph + WAVES);
        }
        if (wave == 0) {
            while (*turn != nb) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    }
    if (wave == 0 && active) {              /* x[k] *= 2.0 / n, then clamp (crs.c:116-120) */
        V acc = sacc[lane];
        const double s = 2.0 / n;
        double rx, ry = 0.;
        if constexpr (VEC == 1) {
            double a0 = *reinterpret_cast<double *>(&acc);
            rx = nla_clamp_box(a0 * s, lb[col], ub[col]);
            *accrow = rx;
        } else {
            double2 a2 = *reinterpret_cast<double2 *>(&acc), r2;
            r2.x = rx = nla_clamp_box(a2.x * s, lb[col], ub[col]);
            r2.y = ry = nla_clamp_box(a2.y * s, lb[col + 1], ub[col + 1]);
            *reinterpret_cast<double2 *>(accrow) = r2;
        }
        if constexpr (SH) {              /* the same chunk into the slot's row on every other rank */
            const int me = S->rank, Wn = S->world;
            for (int r = 0; r < Wn; ++r) {
                if (r == me) continue;
                double *dst = S->peerTX[r] + (size_t) q * tld + tc0 + colc;
                st_sys_f64(dst, rx);
                if constexpr (VEC == 2) st_sys_f64(dst + 1, ry);
            }
        }
        (void) ry;
        /* the stores must have LANDED before this wave's lane 0 counts the chunk as done below (the barrier alone does not wait
         * for them: a workgroup-scope release drops vmcnt).  TX is uncached memory, so landed = visible to every CU: no L2
         * write-back, no agent-scope release */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    /* this chunk of the trial point is final; the workgroup that completes the slot evaluates it */
    __syncthreads();
    if constexpr (SH) {                   /* the chunk has landed in every rank's TX (wave 0's wait above, then the barrier): raise its flag everywhere */
        if (threadIdx.x < (unsigned) S->world)
            st_sys(S->peerflags[threadIdx.x] + (size_t) a * (size_t) S->chunks_total + (size_t) (S->chunk0 + chunk), seq << 2);
    }
    if (threadIdx.x == 0) {
        if (chunk == 0) fwcnt[a] = s_nrec;
        s_last = (atomicAdd(&done[a], 1u) == (uint32_t) (chunks - 1));
    }
    __syncthreads();
    if (!s_last) CH_NEXT;
    if constexpr (SH) {
        /* the last LOCAL chunk of the slot: wait for the other ranks' (their workgroups hold tickets as early as this one's and never
         * wait for this slot), unless the slot lies behind a new best point — then some rank may have left it alone */
        if (wave == 0) {
            const uint32_t *fl = S->peerflags[S->rank] + (size_t) a * (size_t) S->chunks_total;
            const int nch = S->chunks_total;
            const uint64_t t0 = wall_clock64();
            int state = 0;
            while (!state) {
                bool all = true;
                for (int c = lane; c < nch; c += 64) all = all && ((int32_t) (ld_sys(fl + c) - (seq << 2)) >= 0);
                if (__ballot(!all) == 0) state = 1;
                else {
                    const uint32_t h = ld_agent(&ctrl->halt);
                    if ((h & 2u) && (uint32_t) a >= (h >> 8)) state = 2;
                    else if (wall_clock64() - t0 > NLA_SH_WAIT_TICKS) state = 3;
                    else __builtin_amdgcn_s_sleep(4);
                }
            }
            if (lane == 0) s_state = state;
        }
        __syncthreads();
        if (s_state != 1) {
            if (threadIdx.x == 0) {
                status[a].t = 0;
                if (s_state == 3) st_agent(&ctrl->lock, 1u);          /* a peer's chunk never came: the launch has failed (status[K].t) */
            }
            CH_NEXT;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");                 /* system scope: the peers' stores */
    } else
    {
        /* TX is uncached memory and its chunks had landed before they were counted: nothing stale to drop.  Below n = 2048 — where the
         * evaluation is a link of the window's dependency chains — the fence only orders the loads behind the count; from n = 2048 the
         * agent-scope acquire stays (measured: the headline is 2.3 % faster WITH it; crs_chain_resolver.h, ch_landed) */
        if (n < NLA_CHAIN_LIGHT_BELOW) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    {
        const int tid = threadIdx.x;
        const double *x = TX + (size_t) q * tld;
        const double *xb = SH ? S->xbest : X + (size_t) i0 * (size_t) ld;
        const double *lbm = SH ? S->lbf : lb, *ubm = SH ? S->ubf : ub;      /* the mutation clamps whole points */
        const uint32_t *w = words_ring + (size_t) ((block + 1) % ring_blocks) * 2 * (size_t) n;
        double *m = TM + (size_t) q * tld;
        auto getx = [&](int i) { return __builtin_nontemporal_load(x + i); };
        const double fT = sign * nla_block_objective_as<OBJ, WAVES, CH_FWAVES>(n, getx, scratch);
        uint64_t *rec = reinterpret_cast<uint64_t *>(fv) + 2 * (size_t) a;
        /* f(T) is published AT ONCE (the record's first word is its own flag; TX of the slot has landed: the chunks' waits above): the
         * resolver accepts four trials in five on f(T) alone, and whoever waits for this slot need not wait for its mutation too */
#ifndef NLA_CHAIN_LATE_FT                /* (A/B builds: both words at the end, as before round 5) */
        if (tid == 0) __hip_atomic_store(rec, ch_bits_of_f(fT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        auto mut = [&](int i) {        /* p_i = best_i (1+w) - w p_i, clamp (crs.c:140-145) */
            const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
            const double wv = nla_urand_from(0., 1., ww.x, ww.y);
            return nla_clamp_box(xb[i] * (1 + wv) - wv * getx(i), lbm[i], ubm[i]);
        };
        for (int i = tid; i < n; i += WAVES * 64) m[i] = mut(i);
        const double fM = sign * nla_block_objective_as<OBJ, WAVES, CH_FWAVES>(n, mut, scratch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        /* every wave's part of m[] has landed before lane 0 publishes the slot */
        __syncthreads();
        if (tid == 0) {
            status[a].fT = fT; status[a].fM = fM; status[a].t = n; status[a].pad = 0;
            /* the record's second word: f(M) — TM of the slot has landed (the wait above) */
#ifdef NLA_CHAIN_LATE_FT
            __hip_atomic_store(rec, ch_bits_of_f(fT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            __hip_atomic_store(rec + 1, ch_bits_of_f(fM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (!SH) return;
    }
#undef CH_NEXT
}

extern "C" size_t nla_crs_chain_ctrl_bytes(int K, int nW)
{
    return sizeof(chain_ctrl) + sizeof(double) * 2 * (size_t) K + sizeof(uint32_t) * (2 * (size_t) K + (size_t) nW + 8);
}

static bool chain_vec2(int n, int ld)
{
    static int force1 = -1;              /* experiment switch: NLA_CHAIN_VEC1=1 -> 64-coordinate chunks everywhere */
    if (force1 < 0) { const char *s = NLA_DBG_ENV("NLA_CHAIN_VEC1"); force1 = (s && atoi(s) > 0) ? 1 : 0; }
    return !force1 && (n % 2 == 0) && (ld % 2 == 0) && n >= 128;
}

extern "C" int nla_crs_chain_chunks(int n, int ld)
{
    const bool vec2 = chain_vec2(n, ld);
    const int cpw = vec2 ? 128 : 64;
    return (n + cpw - 1) / cpw;
}

extern "C" uint32_t nla_crs_chain_tickets(int n, int ld, int K)
{
    return (uint32_t) nla_crs_chain_chunks(n, ld) * (uint32_t) K + 1u;         /* the slots' workgroups + the resolver's */
}

extern "C" int nla_k_crs_chain_lean(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                                    const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                                    uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                                    const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                                    nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero, void *stream);
extern "C" int nla_k_crs_chain(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                               const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                               uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                               const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                               nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, void *stream)
{
    return nla_k_crs_chain_lean(obj, n, ld, X, i0, f_best, jn_ring, pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, w_on_host,
                                slot_mask, lb, ub, TX, TM, ctrl, ticket_base, status, fwcnt, fwrec, fwcap, 0, stream);
}

/* ctrl_is_zero != 0: the caller has cleared the control block behind its ticket word on this stream already (nla_k_crs_commit_zero).
 * (A doorbell in pinned memory rung by the last workgroup, for a host that spins instead of synchronising the stream, was measured here
 * in round 5 and removed: with a system-scope fence per workgroup n = 512 ran at 612 k evals/s against 687 k with the synchronisation,
 * with one fence by the ringing workgroup 920 k against 980 k — profiles/r05_lean_windows_ab.txt.) */
static int chain_launch(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                        const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                        uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                        const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                        nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero, void *stream,
                        int ncols, const chain_shard *S, int ldf, uint32_t seq, uint32_t stopbits, int grid_cap);
extern "C" int nla_k_crs_chain_lean(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                                    const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                                    uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                                    const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                                    nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero, void *stream)
{
    return chain_launch(obj, n, ld, X, i0, f_best, jn_ring, pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, w_on_host, slot_mask,
                        lb, ub, TX, TM, ctrl, ticket_base, status, fwcnt, fwrec, fwcap, ctrl_is_zero, stream, n, nullptr, ld, 0u, 0u, 0);
}

/* ---- column-sharded windows (SH instance of the kernel) ------------------------------------------------------ */
/* chunks a rank with `ncols` columns (even-padded where n >= 128) contributes per slot */
extern "C" int nla_crs_chain_sh_chunks(int n, int ncols)
{
    const int cpw = (n >= 128 && ncols % 2 == 0) ? 128 : 64;
    return (ncols + cpw - 1) / cpw;
}
/* bytes of the per-rank table the kernel reads (device memory, filled by nla_crs_chain_sh_table) */
extern "C" size_t nla_crs_chain_sh_table_bytes(void) { return sizeof(chain_shard); }
/* fills the host image of that table: peerTX / peerflags / peerstop [r] = rank r's TX, flags, stop words AS MAPPED IN THIS PROCESS (the own
 * entries: the own buffers); xbest / lbf / ubf: whole best row and whole bounds on this device */
extern "C" int nla_crs_chain_sh_table(void *host_image, int world, int rank, int c0, int ldf, int chunks_total, int chunk0, void *const *peerTX,
                                      void *const *peerflags, void *const *peerstop, const double *xbest, const double *lbf, const double *ubf)
{
    if (world < 2 || world > NLA_SH_MAXW || rank < 0 || rank >= world) return (int) hipErrorInvalidValue;
    chain_shard t;
    memset(&t, 0, sizeof t);
    t.world = world; t.rank = rank; t.c0 = c0; t.ldf = ldf; t.chunks_total = chunks_total; t.chunk0 = chunk0;
    for (int r = 0; r < world; ++r) { t.peerTX[r] = (double *) peerTX[r]; t.peerflags[r] = (uint32_t *) peerflags[r]; t.peerstop[r] = (uint32_t *) peerstop[r]; }
    t.xbest = xbest; t.lbf = lbf; t.ubf = ubf;
    memcpy(host_image, &t, sizeof t);
    return 0;
}
extern "C" size_t nla_crs_chain_sh_stop_bytes(void) { return sizeof(uint32_t) * 2 * NLA_SH_MAXW; }
/* one window on this rank's columns.  X: N x ld (the slice), ncols columns of it gathered (nc, even-padded); TX / TM: rows of ldf doubles
 * (whole points), TX peer-mapped on every rank; lb / ub: the slice's bounds; table: device copy of the nla_crs_chain_sh_table image;
 * seq: the launch's number (1, 2, ...: the same on every rank); stopbits: bit 0 force_stop, bit 1 maxtime as THIS rank sees them;
 * grid_cap >= 2: at most that many workgroups (ranks sharing one device: all of them must fit the chip together), 0: one per ticket.
 * status[K] comes back with (fT != 0: some rank's force_stop, fM != 0: some rank's clock, t != 0: a rank's chunks did not arrive). */
extern "C" int nla_k_crs_chain_sh(int obj, int n, int ncols, int ld, int ldf, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                                  const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                                  uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                                  const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                                  nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero,
                                  const void *table, uint32_t seq, uint32_t stopbits, int grid_cap, void *stream)
{
    if (!table || ncols < 1 || ncols > ld || ldf < n || ldf % 16 != 0) return (int) hipErrorInvalidValue;
    return chain_launch(obj, n, ld, X, i0, f_best, jn_ring, pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, w_on_host, slot_mask,
                        lb, ub, TX, TM, ctrl, ticket_base, status, fwcnt, fwrec, fwcap, ctrl_is_zero, stream, ncols, (const chain_shard *) table, ldf, seq, stopbits, grid_cap);
}
/* workgroups of one launch and tickets they draw (every workgroup but the resolver's draws one ticket past the last) */
static unsigned chain_sh_grid(int n, int ncols, int K, int grid_cap)
{
    const long total = (long) nla_crs_chain_sh_chunks(n, ncols) * K + 1;
    return (unsigned) ((grid_cap >= 2 && grid_cap < total) ? grid_cap : total);
}
extern "C" uint32_t nla_crs_chain_sh_tickets(int n, int ncols, int K, int grid_cap)
{
    return (uint32_t) nla_crs_chain_sh_chunks(n, ncols) * (uint32_t) K + chain_sh_grid(n, ncols, K, grid_cap);
}

static int chain_launch(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                        const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                        uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                        const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                        nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero, void *stream,
                        int ncols, const chain_shard *S, int ldf, uint32_t seq, uint32_t stopbits, int grid_cap)
{
    if (K <= 0) return 0;
    if (K > 256 || nW > 256 || nW < 0 || obj < 0) return (int) hipErrorInvalidValue;
    const double sign = nla_obj_sign(&obj);
    /* rows of TX / TM start on a 128-byte line: a line that holds the end of one slot's row and the start of the next one's could
     * sit in a CU's vector L1 from the read of the first and serve a stale start of the second (consumers take no L1 invalidate) */
    if (ld % 16 != 0 || ((uintptr_t) TX | (uintptr_t) TM) % 128 != 0) return (int) hipErrorInvalidValue;
    (void) ldf;
    hipStream_t st = (hipStream_t) stream;
    chain_lists L;
    L.inl = 0;
    if (w_on_host) {                                          /* the lists travel as kernel arguments: no copy in front of the launch */
        if (nW > NLA_KA_MAX) return (int) hipErrorInvalidValue;
        L.inl = 1;
        for (int j = 0; j < nW; ++j) { L.W[j] = W[j]; L.Wf[j] = Wf[j]; }
        W = nullptr; Wf = nullptr;
    }
    const bool sh = S != nullptr;
    const bool vec2 = sh ? (n >= 128 && ncols % 2 == 0 && ld % 2 == 0) : chain_vec2(n, ld);
    const int chunks = sh ? nla_crs_chain_sh_chunks(n, ncols) : nla_crs_chain_chunks(n, ld);
    if (sh && (ldf % 16 != 0)) return (int) hipErrorInvalidValue;
    const dim3 grid(sh ? chain_sh_grid(n, ncols, K, grid_cap) : (unsigned) ((long) chunks * K) + 1u);
    const uint64_t res_timeout = 200000000ull;                /* 2 s of the 100 MHz clock without a single evaluation arriving */
    chain_ctrl *c = (chain_ctrl *) ctrl;
    /* everything but the ticket counter starts from zero */
    if (!ctrl_is_zero) {
        hipError_t e = hipMemsetAsync((char *) ctrl + sizeof(uint32_t), 0, nla_crs_chain_ctrl_bytes(K, nW) - sizeof(uint32_t), st);
        if (e != hipSuccess) return (int) e;
    }
#define CHAIN(VEC, UU, WV, O) hipLaunchKernelGGL((crs_chain_kernel<VEC, UU, WV, O>), grid, dim3(WV * 64), 0, st, L, n, ld, X, i0, f_best, jn_ring, \
        pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, slot_mask, chunks, lb, ub, TX, TM, c, ticket_base, status,         \
        fwcnt, fwrec, fwcap, sign, res_timeout, ncols, S, seq, stopbits)
#define CHAIN_SH(VEC, UU, WV, O) hipLaunchKernelGGL((crs_chain_kernel<VEC, UU, WV, O, true>), grid, dim3(WV * 64), 0, st, L, n, ld, X, i0, f_best, jn_ring, \
        pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, slot_mask, chunks, lb, ub, TX, TM, c, ticket_base, status,         \
        fwcnt, fwrec, fwcap, sign, res_timeout, ncols, S, seq, stopbits)
/* rows in flight per workgroup = wavefronts x U (tuning builds override: NLOPT_AMD_VARIANT="name:-DNLA_CHAIN_MID_W=8 ...") */
/* Measured on the MI355X, N = 1e5 (profiles/r05_lean_windows_ab.txt; was 4 x 16 from n = 512, 2 x 16 from 128, 1 x 16 below): the time of a
 * window below n = 2048 is the DEPTH of its dependency chains (a slot that picked one of the worst rows ahead of it waits for the chain,
 * then finishes its gather, is evaluated, and lets the next one go) times what one such step takes — so what counts is how fast ONE
 * slot gets through its rows, not how many slots are resident: 8 x 32 rows in flight per workgroup from n = 512 (n = 512: 688 -> 798 k
 * evals/s, n = 1024: 290 -> 378 k), 4 wavefronts below (n = 256: 1.03 -> 1.13 M, n = 64: 1.53 -> 1.70 M; 8 there: 1.10 / 1.55 M) */
#ifndef NLA_CHAIN_MID_W
#define NLA_CHAIN_MID_W 8
#endif
#ifndef NLA_CHAIN_MID_U
#define NLA_CHAIN_MID_U 32
#endif
#ifndef NLA_CHAIN_LOW_W
#define NLA_CHAIN_LOW_W 4
#endif
#ifndef NLA_CHAIN_LOW_U
#define NLA_CHAIN_LOW_U 16
#endif
#ifndef NLA_CHAIN_TINY_W
#define NLA_CHAIN_TINY_W 4
#endif
#define CHAIN_SHAPE(O)                                                                   \
    if (vec2) {                                                                          \
        if (n >= 2048) CHAIN(2, 32, 8, O); else if (n >= 512) CHAIN(2, NLA_CHAIN_MID_U, NLA_CHAIN_MID_W, O); else CHAIN(2, NLA_CHAIN_LOW_U, NLA_CHAIN_LOW_W, O); \
    } else {                                                                             \
        if (n >= 2048) CHAIN(1, 32, 8, O); else if (n >= 512) CHAIN(1, NLA_CHAIN_MID_U, NLA_CHAIN_MID_W, O); else if (n >= 128) CHAIN(1, NLA_CHAIN_LOW_U, NLA_CHAIN_LOW_W, O); else CHAIN(1, 16, NLA_CHAIN_TINY_W, O); \
    }
/* the sharded instance in three shapes (a rank's share of a row is short: the 8 x 32 tiling from n = 512 on, 4 x 16 below) */
#define CHAIN_SHAPE_SH(O)                                                                \
    if (vec2) { if (n >= 512) CHAIN_SH(2, 32, 8, O); else CHAIN_SH(2, 16, 4, O); } else CHAIN_SH(1, 16, 4, O);
    if (sh) { NLA_OBJ_DISPATCH(obj, CHAIN_SHAPE_SH) }
    else { NLA_OBJ_DISPATCH(obj, CHAIN_SHAPE) }
#undef CHAIN_SHAPE_SH
#undef CHAIN_SH
#undef CHAIN_SHAPE
#undef CHAIN
    NLA_LAUNCH_CHECK();
    return 0;
}
