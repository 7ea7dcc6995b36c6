/* isres_kernels.hip — Improved Stochastic Ranking Evolution Strategy (NLOPT_GN_ISRES) on gfx950.
 *
 * Reference loops replaced (src/algs/isres/isres.c): the initial population :122-128, the
 * per-candidate evaluation of f and the constraint penalties :134-166, the selection :202-229
 * (sort by f when everything is feasible; otherwise Runarsson & Yao's stochastic ranking — up to
 * pop sweeps of pop-1 adjacent compare-exchanges, each drawing one uniform, 2.5e9 serial steps at
 * pop = 5e4: the reference's hot spot), the standard mutation :234-252 and the differential
 * variation :253-280.  All randomness comes from the reference's single MT19937 stream at the
 * reference's offsets; the word stream is produced by mt_kernels.hip.
 *
 * Data layout in HBM: X and S (step sizes) are pop x ld fp64 row-major, ld = n rounded up to even;
 * F/PEN/GPEN/FEAS are pop-vectors.  Ranking works on 64-bit packed elements, laid out so that each of the two
 * compares of a ranking step is one 32-bit compare after one shift / mask:
 *   low word   [0,12) individual, low 12 bits | [12,32) dense rank of f
 *   high word  [0,8) individual, high 8 bits  | [8,28) dense rank of penalty | bit 31: penalty == 0
 * (dense rank = number of strictly smaller values, so integer compares reproduce the reference's
 * fp64 compares including ties); populations up to 2^20 — larger ones only without constraints, where no generation ranks
 * stochastically (the sort by f needs no packed elements).
 */
#include "dev_common.h"
#include "../../../include/nlopt_amd.h"

#define ISRES_IDX_BITS 20
#define ISRES_IDX_MASK 0xFFFFFu
__host__ __device__ __forceinline__ uint64_t isres_pack(uint32_t idx, uint32_t rf, uint32_t rp, bool pzero)
{
    const uint32_t lo = (rf << 12) | (idx & 0xFFFu), hi = (pzero ? 0x80000000u : 0u) | (rp << 8) | (idx >> 12);
    return ((uint64_t) hi << 32) | lo;
}
__host__ __device__ __forceinline__ uint32_t isres_unpack_idx(uint64_t e) { return ((uint32_t) e & 0xFFFu) | ((((uint32_t) (e >> 32)) & 0xFFu) << 12); }

/* ------------------------------------------------------------------------------------------------
 * initial population (isres.c:122-128): xs[k][j] = urand(lb_j, ub_j) k-major from the stream,
 * sigma[k][j] = (ub_j - lb_j)/sqrt(n); row 0 := the caller's x.  One wavefront per individual.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void isres_init_kernel(int n, int ld, const double *__restrict__ lb, const double *__restrict__ ub,
                                                          const uint32_t *__restrict__ words, int64_t k_first, int64_t count,
                                                          const double *__restrict__ x0, double *__restrict__ X, double *__restrict__ S)
{
    const int lane = threadIdx.x & 63;
    const int64_t kl = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (kl >= count) return;
    const int64_t k = k_first + kl;
    const uint32_t *w = words + (size_t) kl * 2 * (size_t) n;
    const double sq = sqrt((double) n);
    for (int j = lane; j < n; j += 64) {
        const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * j);
        X[(size_t) k * ld + j] = (k == 0) ? x0[j] : nla_urand_from(lb[j], ub[j], ww.x, ww.y);
        S[(size_t) k * ld + j] = (ub[j] - lb[j]) / sq;
    }
}

/* ------------------------------------------------------------------------------------------------
 * evaluation (isres.c:138-166): f by one wavefront per individual; every constraint by ONE lane
 * in the reference's sequential order (so the penalties are bit-identical to the host callbacks),
 * combined in constraint order: inequalities (gval > tol -> infeasible, max(g,0)^2), the
 * inequality-only snapshot gpenalty, then equalities (|h| > tol -> infeasible, h^2).
 * ---------------------------------------------------------------------------------------------- */
__device__ __forceinline__ double isres_constraint_value(const nla_dev_constraint &c, int n, const double *x)
{
    /* NLA_CON_BLOCKSUM: sum_{i in block q of Q} x_i - 1 (objfuncs.h nla_con_blocksum_seq) */
    const unsigned lo = (unsigned) (((unsigned long long) c.q * (unsigned) n) / c.Q);
    const unsigned hi = (unsigned) (((unsigned long long) (c.q + 1) * (unsigned) n) / c.Q);
    double s = 0;
    for (unsigned i = lo; i < hi; ++i) s += x[i];
    return s - 1;
}

template <int OBJ>
__global__ __launch_bounds__(256) void isres_eval_kernel(int n, int ld, const double *__restrict__ X, int64_t pop, int m, int p,
                                                          const nla_dev_constraint *__restrict__ con, double *__restrict__ F,
                                                          double *__restrict__ PEN, double *__restrict__ GPEN, int32_t *__restrict__ FEAS, double sign)
{
    const int lane = threadIdx.x & 63;
    const int64_t k = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= pop) return;
    const double *x = X + (size_t) k * (size_t) ld;
    double f = 0;
    if constexpr (OBJ >= 0) f = nla_wave_objective<OBJ>(n, [&](int i) { return x[i]; });   /* OBJ < 0: f comes from a user kernel */
    double pen = 0, gpen = 0;
    int feas = 1;
    const int mp = m + p;
    for (int c0 = 0; c0 < mp; c0 += 64) {
        const int c = c0 + lane;
        const double v = (c < mp) ? isres_constraint_value(con[c], n, x) : 0.0;
        const double tol = (c < mp) ? con[c].tol : 0.0;
        const int cnt = mp - c0 < 64 ? mp - c0 : 64;
        for (int u = 0; u < cnt; ++u) {                 /* in constraint order, every lane the same chain */
            double g = __shfl(v, u, 64);
            const double t = __shfl(tol, u, 64);
            if (c0 + u == m) gpen = pen;
            if (c0 + u < m) {
                if (g > t) feas = 0;
                if (g < 0) g = 0;
                pen += g * g;
            } else {
                if (fabs(g) > t) feas = 0;
                pen += g * g;
            }
        }
    }
    if (p == 0) gpen = pen;
    if (lane == 0) { if (OBJ >= 0) F[k] = sign * f; PEN[k] = pen; GPEN[k] = gpen; FEAS[k] = feas; }
}

/* ------------------------------------------------------------------------------------------------
 * dense ranks by counting (O(pop^2) compares, LDS-tiled): rf = #{F_j < F_k}, rp = #{PEN_j < PEN_k},
 * and the position of k in the stable sort by f, ps = #{F_j < F_k or (F_j == F_k and j < k)} —
 * the all-feasible ranking of isres.c:204 (glibc's qsort_r is a stable merge sort, qsort_r.c:190).
 * Outputs: elems[k] = packed element of individual k; sorted[ps] = k.
 * ---------------------------------------------------------------------------------------------- */
#define RC_W 8                           /* wavefronts per workgroup: wavefront w counts over the tiles w, w + 8, ... of the population */
#define RC_T 256                         /* individuals per tile (each wavefront stages its own: no workgroup barrier inside the loop) */
/* lane = individual (64 per workgroup), the j range dealt over the workgroup's 8 wavefronts, partial counts summed through LDS: 782
 * workgroups / 6256 wavefronts at pop = 5e4 where one thread per individual and 256 per workgroup gave 196 / 784 — the kernel sits in
 * front of the ranking pipeline on the generation's critical path (4.8 ms at config 3 in that shape, rounds 1-4) */
typedef double rc_d8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(RC_W * 64) void isres_rank_count_kernel(int64_t pop, const double *__restrict__ F, const double *__restrict__ PEN,
                                                                     uint64_t *__restrict__ elems, int32_t *__restrict__ sorted)
{
    __shared__ uint32_t part[3][RC_W][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const int64_t k = (int64_t) blockIdx.x * 64 + lane;
    const bool live = k < pop;
    const double fk = live ? F[k] : 0.0, pk = live ? PEN[k] : 0.0;
    uint32_t rf = 0, rp = 0, ps = 0;
    /* the values everybody is compared with come through the SCALAR unit (round 5; tiles staged in LDS before: two broadcast LDS reads per
     * comparison kept the kernel at a fifth of its instruction rate): the index j is the same for the whole wavefront, so F[j], PEN[j] are
     * scalar loads — 8 values per instruction out of the constant cache — and the comparisons take them as scalar operands */
    for (int64_t j0 = (int64_t) wave * RC_T; j0 < pop; j0 += (int64_t) RC_W * RC_T) {
        const int cnt = (int) (pop - j0 < RC_T ? pop - j0 : RC_T);
        const double *Fj = F + j0, *Pj = PEN + j0;
        const int64_t kd = k - j0;                                      /* "j < k" inside the tile, in 32 bits */
        const int kk = kd < 0 ? 0 : kd > RC_T ? RC_T : (int) kd;
        int i = 0;
        if ((j0 & 7) == 0) {       /* RC_T and the tile origin are multiples of 8: the wide loads are aligned */
            rc_d8 fa = *(const rc_d8 *) (Fj), pa = *(const rc_d8 *) (Pj);
            for (; i + 8 <= cnt; i += 8) {
                const rc_d8 fv = fa, pv = pa;
                if (i + 16 <= cnt) { fa = *(const rc_d8 *) (Fj + i + 8); pa = *(const rc_d8 *) (Pj + i + 8); }   /* the next 8, in flight */
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const double fj = fv[u], pj = pv[u];
                    rf += fj < fk;
                    rp += pj < pk;
                    ps += (uint32_t) (fj < fk) | ((uint32_t) (fj == fk) & (uint32_t) (i + u < kk));
                }
            }
        }
        for (; i < cnt; ++i) {
            const double fj = Fj[i], pj = Pj[i];
            rf += fj < fk;
            rp += pj < pk;
            ps += (uint32_t) (fj < fk) | ((uint32_t) (fj == fk) & (uint32_t) (i < kk));
        }
    }
    part[0][wave][lane] = rf; part[1][wave][lane] = rp; part[2][wave][lane] = ps;
    __syncthreads();
    if (wave == 0 && live) {
        rf = rp = ps = 0;
        for (int w = 0; w < RC_W; ++w) { rf += part[0][w][lane]; rp += part[1][w][lane]; ps += part[2][w][lane]; }
        if (pop <= (1 << ISRES_IDX_BITS)) elems[k] = isres_pack((uint32_t) k, rf, rp, pk == 0);
        sorted[ps] = (int32_t) k;
    }
}

/* ------------------------------------------------------------------------------------------------
 * the ranking's uniforms (isres.c:210: u = nlopt_urand(0,1), one per step, unconditionally) reduced
 * to the only thing the algorithm asks of them: the bit u < PF.  Step (sweep i, position j) uses
 * stream words 2(i(pop-1)+j), +1.  Rows of `words` = whole sweeps; output row i = ceil((pop-1)/64)
 * 64-bit words, bit j of the row = (u_{i,j} < 0.45).  One wavefront per (sweep, 2048 steps).
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void isres_bits_kernel(const uint32_t *__restrict__ words, int64_t row_first, int nrows, int64_t popm1,
                                                         int64_t rowwords, uint64_t *__restrict__ bits)
{
    const int lane = threadIdx.x;
    const int64_t segs = (popm1 + 2047) / 2048;
    const int64_t r = blockIdx.x / segs, seg = blockIdx.x - r * segs;
    if (r >= nrows) return;
    const uint32_t *w = words + (size_t) r * 2 * (size_t) popm1;
    for (int it = 0; it < 32; ++it) {
        const int64_t j = seg * 2048 + (int64_t) it * 64 + lane;
        bool b = false;
        if (j < popm1) {
            const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * j);
            b = nla_urand_from(0., 1., ww.x, ww.y) < 0.45;                   /* PF, isres.c:72 */
        }
        const uint64_t mask = __ballot(b);
        if (lane == 0 && seg * 2048 + (int64_t) it * 64 < popm1) bits[(size_t) (row_first + r) * rowwords + seg * 32 + it] = mask;
    }
}

/* ------------------------------------------------------------------------------------------------
 * stochastic ranking as a systolic pipeline.  One sweep of the reference's bubble pass
 * (isres.c:208-226) is a stream transducer with one element of state: it holds a carry c, reads the
 * next element x, and — by fval if (u < PF or both penalties are zero), else by penalty — either
 * emits x and keeps c (the reference's swap) or emits c and keeps x.  Sweep i+1 consumes exactly
 * the stream sweep i emits, so the pop sweeps are pop chained stages: stage i handles its j-th
 * compare two ticks after stage i-1 handled its (j+1)-th.  Lane = stage (64 consecutive sweeps per
 * wavefront, elements move lane-to-lane by DPP wave shift); wavefront u hands its output stream to
 * wavefront u+1 through a global buffer, 64 elements per publication (device-coherent stores, then a
 * progress counter).  Units are claimed by ticket so that a unit only ever waits for units that are
 * already running.  3*pop ticks instead of pop^2 steps.
 *
 * Every tick is on the serial path of the whole pipeline (pop + 2 sweeps + 63 units ticks end to end), so
 * the tick is ~25 straight-line instructions with no per-lane phase logic:
 *  - sentinels instead of phases: a stage starts with the carry "-inf" (all zero: never greater than
 *    anything, so it is emitted untouched and the first real element becomes the carry — the reference's
 *    first read), and the unit's input is followed by "+inf" elements (nothing is greater: every stage
 *    emits its carry when the first one arrives — the flush — and then passes +inf on).  Compares that
 *    involve a sentinel never swap, whatever their bit;
 *  - the unit's 64 inputs of a block sit one per lane and move down one lane per tick (wave_shl), lane 0
 *    always holds the current one and the DPP shift that brings the left neighbour's output leaves it
 *    there (`old` operand); lane 63's outputs are collected the same way, one lane per tick;
 *  - the u < PF bits of a lane's next 64 ticks are one 64-bit window cut from two row words once per
 *    block (the cut position 63 - 2 lane (mod 64) never changes); a tick tests bit k of it, k static.
 * ---------------------------------------------------------------------------------------------- */
#include "isres_stochrank.h"

/* final order: irank[pos] = individual of the element at pos */
__global__ __launch_bounds__(256) void isres_unpack_kernel(int64_t pop, const uint64_t *__restrict__ stream, int32_t *__restrict__ irank)
{
    const int64_t k = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (k < pop) irank[k] = (int32_t) isres_unpack_idx(stream[k]);
}

/* ------------------------------------------------------------------------------------------------
 * normal deviates (mt19937ar.c:216-232, Box-Muller polar): during the evolve phase every draw is an
 * nlopt_nrand, so the stream is a sequence of 4-word attempts, each accepted (s < 1) or not
 * independently of everything else.  Attempt a of the phase uses words 4a..4a+3.  The accepted
 * deviates, in order, are what the serial algorithm consumes one by one: compaction in two passes
 * (count per workgroup, host-free scan, write).  zatt[i] = attempt index of the i-th deviate.
 * ---------------------------------------------------------------------------------------------- */
#define NR_PER_WG 1024
__device__ __forceinline__ bool isres_attempt(const uint32_t *w, double &z)
{
    const uint4 q = *reinterpret_cast<const uint4 *>(w);
    const double v1 = nla_urand_from(-1., 1., q.x, q.y), v2 = nla_urand_from(-1., 1., q.z, q.w);
    const double s = v1 * v1 + v2 * v2;
    if (s >= 1.0) return false;
    z = (s == 0) ? 0.0 : 0.0 + v1 * sqrt(-2 * log(s) / s) * 1.0;
    return true;
}

__global__ __launch_bounds__(256) void isres_nrand_count_kernel(const uint32_t *__restrict__ words, int64_t nattempts, int32_t *__restrict__ counts)
{
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    int c = 0;
    for (int r = 0; r < NR_PER_WG / 256; ++r) {
        const int64_t a = (int64_t) blockIdx.x * NR_PER_WG + r * 256 + threadIdx.x;
        double z;
        if (a < nattempts && isres_attempt(words + 4 * a, z)) ++c;
    }
    atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
}

/* exclusive scan of counts[0..nwg) in place (one workgroup), total added to *ztotal */
__global__ __launch_bounds__(1024) void isres_scan_kernel(int32_t *__restrict__ counts, int nwg, int64_t *__restrict__ ztotal)
{
    __shared__ int64_t s_part[1024];
    __shared__ int64_t s_base;
    const int tid = threadIdx.x;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nwg; b0 += 1024) {
        const int i = b0 + tid;
        const int64_t v = i < nwg ? counts[i] : 0;
        s_part[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int64_t add = tid >= off ? s_part[tid - off] : 0;
            __syncthreads();
            s_part[tid] += add;
            __syncthreads();
        }
        const int64_t incl = s_part[tid], base = s_base;
        if (i < nwg) counts[i] = (int32_t) (base + incl - v);
        __syncthreads();
        if (tid == 1023) s_base = base + incl;
        __syncthreads();
    }
    if (tid == 0) *ztotal += s_base;
}

__global__ __launch_bounds__(256) void isres_nrand_write_kernel(const uint32_t *__restrict__ words, int64_t nattempts, int64_t attempt_base,
                                                                 const int32_t *__restrict__ offsets, int64_t zbase,
                                                                 double *__restrict__ z, int64_t *__restrict__ zatt)
{
    __shared__ int s_wave[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t base = zbase + offsets[blockIdx.x];
    for (int r = 0; r < NR_PER_WG / 256; ++r) {
        const int64_t a = (int64_t) blockIdx.x * NR_PER_WG + r * 256 + threadIdx.x;
        double zv = 0;
        const bool ok = a < nattempts && isres_attempt(words + 4 * a, zv);
        const uint64_t m = __ballot(ok);
        const int before = __popcll(m & ((1ull << lane) - 1));
        __syncthreads();
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wave[w];
        if (ok) { z[base + wbase + before] = zv; zatt[base + wbase + before] = attempt_base + a; }
        base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    }
}

/* ------------------------------------------------------------------------------------------------
 * the evolve phase (isres.c:234-280) by ONE wavefront: the individuals are a serial chain through
 * the position in the deviate sequence (how many deviates individual k consumes depends on how
 * often its coordinates had to be redrawn, :245-248,270-273).  Inside an individual the lanes take
 * contiguous coordinate chunks, each sequentially from an assumed starting position; the chunk
 * consumptions are prefix-summed into new starting positions and the chunks recomputed until the
 * positions stop changing (a shift only matters if it changes some later chunk's redraw count,
 * which is rare: a few rounds).  Mutation reads the parent's rows and writes the child's;
 * variation updates a survivor in place from saved copies of its own row, of physical row 0 as it
 * was after the mutation loop (:253) and of the CURRENT physical row k+1 (:260).
 * state[0] = next individual, state[1] = next deviate, state[2] = 1 if the deviates ran out.
 * ---------------------------------------------------------------------------------------------- */
struct isres_evolve_args {
    int n, ld, phase;              /* phase 0: mutation k in [survivors, pop), 1: variation k in [0, survivors) */
    int64_t pop, survivors, zcount;
    double taup, tau;
    const double *lb, *ub, *z;
    const int32_t *irank;
    double *X, *S, *scratch;       /* scratch: 3*ld doubles (x0 copy | own x | own sigma) */
    int64_t *state;
};

__global__ __launch_bounds__(64) void isres_evolve_kernel(isres_evolve_args A)
{
    const int lane = threadIdx.x;
    const int n = A.n, ld = A.ld;
    const int chunk = (n + 63) / 64;
    const int j0 = lane * chunk, j1 = (j0 + chunk < n) ? j0 + chunk : n;
    const double ALPHA = 0.2, GAMMA = 0.85;
    const double sqn = sqrt((double) n);
    double *x0c = A.scratch, *xown = A.scratch + ld, *sown = A.scratch + 2 * ld;
    volatile double *Xv = A.X;
    volatile double *Sv = A.S;
    int64_t k = A.state[0], pos = A.state[1];
    int64_t kend = A.phase == 0 ? A.pop : A.survivors;
    if (A.state[14] > 0 && A.state[14] < kend) kend = A.state[14];      /* stop before this individual (fallback use) */
    if (A.phase == 1 && k == 0) {           /* memcpy(x0, xs, n) before the first survivor (isres.c:253) */
        for (int j = lane; j < n; j += 64) x0c[j] = Xv[j];
        __threadfence();
    }
    bool ranout = false;
    for (; k < kend && !ranout; ++k) {
        const int64_t rk = A.irank[k];
        const int64_t ri = A.phase == 0 ? A.irank[k % A.survivors] : rk;
        const bool lastsurv = (k + 1 == A.survivors);
        if (pos >= A.zcount) { ranout = true; break; }
        const double taup_rand = A.taup * A.z[pos];                 /* one deviate per individual, always */
        if (A.phase == 1) {                 /* in-place update: work from copies of the survivor's own rows */
            for (int j = lane; j < n; j += 64) { xown[j] = Xv[(size_t) rk * ld + j]; sown[j] = Sv[(size_t) rk * ld + j]; }
            __threadfence();
        }
        const volatile double *xp = A.phase == 0 ? Xv + (size_t) ri * ld : xown;
        const volatile double *sp = A.phase == 0 ? Sv + (size_t) ri * ld : sown;
        const volatile double *xk1 = Xv + (size_t) (k + 1) * ld;    /* physical row k+1 (variation only) */
        const bool self = (k + 1 == rk);
        int64_t shift = 0;                  /* deviates consumed by the chunks before this lane's */
        int64_t total = 0;
        for (int round = 0; round < 66; ++round) {
            int64_t cur = pos + 1 + shift, used = 0;
            bool over = false;
            for (int j = j0; j < j1; ++j) {
                const double xi = xp[j], sigi = sp[j];
                double xnew = xi;
                bool mutate = true;
                if (A.phase == 1) {
                    if (!lastsurv) {
                        const double other = self ? xown[j] : xk1[j];
                        xnew = xi + GAMMA * (x0c[j] - other);
                    }
                    mutate = lastsurv || xnew < A.lb[j] || xnew > A.ub[j];
                }
                double snew = sigi;
                if (mutate) {
                    if (cur + 1 >= A.zcount) { over = true; break; }
                    const double sigmamax = (A.ub[j] - A.lb[j]) / sqn;
                    double sg = sigi * exp(taup_rand + A.tau * A.z[cur]);
                    if (sg > sigmamax) sg = sigmamax;
                    int64_t t = 1;
                    for (;;) {
                        if (cur + t >= A.zcount) { over = true; break; }
                        xnew = xi + sg * A.z[cur + t];
                        if (!(xnew < A.lb[j] || xnew > A.ub[j])) break;
                        ++t;
                    }
                    if (over) break;
                    snew = sigi + ALPHA * (sg - sigi);
                    cur += 1 + t; used += 1 + t;
                }
                Xv[(size_t) rk * ld + j] = xnew;
                Sv[(size_t) rk * ld + j] = snew;
            }
            if (__ballot(over)) { ranout = true; break; }
            /* exclusive prefix sum of `used` over the lanes */
            int64_t incl = used;
            for (int off = 1; off < 64; off <<= 1) {
                const int64_t o = __shfl_up(incl, off, 64);
                if (lane >= off) incl += o;
            }
            const int64_t nshift = incl - used;
            total = __shfl(incl, 63, 64);
            const bool changed = nshift != shift;
            shift = nshift;
            if (!__ballot(changed)) break;
        }
        if (ranout) {
            if (A.phase == 1) {             /* undo the partial in-place update */
                __threadfence();
                for (int j = lane; j < n; j += 64) { Xv[(size_t) rk * ld + j] = xown[j]; Sv[(size_t) rk * ld + j] = sown[j]; }
            }
            break;
        }
        pos += 1 + total;
        __threadfence();
    }
    if (lane == 0) { A.state[0] = k; A.state[1] = pos; A.state[2] = ranout ? 1 : 0; }
}

/* LDS-staged variant for n <= ISRES_EVOLVE_LDS_MAXN, one workgroup of EV_T threads: per individual
 * the parent rows and a window of the deviate sequence are brought into LDS with coalesced loads,
 * the step-size factor exp(taup_rand + tau z) of EVERY deviate of the window is computed once in
 * parallel (which deviate a coordinate ends up using is only known after the fixpoint; the factor
 * does not depend on that), every round of the fixpoint works out of LDS with one short coordinate
 * chunk per thread, and the child rows are written back once, coalesced.  Same contract and state
 * as isres_evolve_kernel. */
#define ISRES_EVOLVE_LDS_MAXN 1150
#define EV_T 256
#define EV_D 32                             /* shift band of the one-pass path */
__device__ __forceinline__ int ev_block_exscan(int v, int *s_w, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    __syncthreads();                        /* s_w free again */
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < EV_T / 64; ++w) { const int t = s_w[w]; if (w < wave) base += t; }
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + incl - v;
}

__global__ __launch_bounds__(EV_T) void isres_evolve_lds_kernel(isres_evolve_args A)
{
    extern __shared__ double sm[];
    __shared__ int s_w[EV_T / 64];
    __shared__ int8_t s_tab[2][EV_T][EV_D];     /* per-thread shift maps of the one-pass path, ping-pong */
    const int tid = threadIdx.x;
    const int n = A.n, ld = A.ld;
    const int chunk = (n + EV_T - 1) / EV_T;
    const int j0 = tid * chunk < n ? tid * chunk : n, j1 = (j0 + chunk < n) ? j0 + chunk : n;
    const double ALPHA = 0.2, GAMMA = 0.85;
    const double sqn = sqrt((double) n);
    const int ZW = 3 * n + 64;              /* deviates staged per individual: 1 + 2n + room for redraws */
    double *xp = sm, *sp = sm + n, *xo = sm + 2 * n, *so = sm + 3 * n, *x0c = sm + 4 * n, *xk1 = sm + 5 * n, *lbs = sm + 6 * n,
           *ubs = sm + 7 * n, *smax = sm + 8 * n, *zw = sm + 9 * n, *gw = sm + 9 * n + ZW;
    int64_t k = A.state[0], pos = A.state[1];
    int64_t kend = A.phase == 0 ? A.pop : A.survivors;
    if (A.state[14] > 0 && A.state[14] < kend) kend = A.state[14];      /* stop before this individual (fallback use) */
    for (int j = tid; j < n; j += EV_T) { lbs[j] = A.lb[j]; ubs[j] = A.ub[j]; smax[j] = (A.ub[j] - A.lb[j]) / sqn; }
    if (A.phase == 1) {                     /* memcpy(x0, xs, n) before the first survivor (isres.c:253); kept for resumes */
        if (k == 0) for (int j = tid; j < n; j += EV_T) A.scratch[j] = A.X[j];
        __threadfence();
        __syncthreads();
        for (int j = tid; j < n; j += EV_T) x0c[j] = __builtin_nontemporal_load(A.scratch + j);
    }
    __syncthreads();
    bool ranout = false;
    int64_t rounds_total = 0;
    int64_t c_eval = 0, c_scan = 0, c_fin = 0, c_all = 0, c_stage = 0;
    double rho = -1;                        /* redraws per coordinate, running estimate (one-pass path) */
    for (; k < kend; ++k) {
        const int64_t rk = A.irank[k];
        const int64_t ri = A.phase == 0 ? A.irank[k % A.survivors] : rk;
        const bool lastsurv = (k + 1 == A.survivors);
        const bool self = (k + 1 == rk);
        if (pos >= A.zcount) { ranout = true; break; }
        const int64_t q0 = __builtin_readcyclecounter();
        {   /* stage: parent (or own) rows, physical row k+1, deviate window; rows may have been rewritten
             * by an earlier individual of this phase on this CU: bypass the L1 */
            const double *xr = A.X + (size_t) ri * ld, *sr = A.S + (size_t) ri * ld;
            const double *kr = A.X + (size_t) (k + 1) * ld;
            const bool needk1 = A.phase == 1 && !lastsurv;
            for (int j = tid; j < n; j += EV_T) {
                xp[j] = __builtin_nontemporal_load(xr + j);
                sp[j] = __builtin_nontemporal_load(sr + j);
                if (needk1) xk1[j] = __builtin_nontemporal_load(kr + j);
            }
            const int64_t zlast = A.zcount - 1;
            for (int i = tid; i < ZW; i += EV_T) { const int64_t g = pos + i; zw[i] = A.z[g <= zlast ? g : zlast]; }
        }
        __syncthreads();
        const double taup_rand = A.taup * zw[0];                    /* one deviate per individual, always */
        if (A.phase == 0) for (int i = tid; i < ZW; i += EV_T) gw[i] = exp(taup_rand + A.tau * zw[i]);
        __syncthreads();
        auto zat = [&](int64_t idx) -> double { const int64_t r = idx - pos; return r < ZW ? zw[r] : A.z[idx]; };
        auto gat = [&](int64_t idx) -> double {
            const int64_t r = idx - pos;
            return (A.phase == 0 && r < ZW) ? gw[r] : exp(taup_rand + A.tau * zat(idx));
        };
        const int64_t q1 = __builtin_readcyclecounter();
        c_stage += q1 - q0;
        int shift = 0, total = 0;
        bool resolved = false;
        /* One-pass path (mutation, one coordinate per thread).  Coordinate j draws at window index
         * 1 + 2j + s, s = redraws before it — unknown, but close to rho*j.  Every thread evaluates its
         * coordinate for all EV_D shifts around that guess (cheap: the exp factors are tabulated) and
         * publishes the map  shift in -> shift out;  composing the maps in coordinate order (a prefix
         * "scan" under composition, log2(EV_T) steps) yields every coordinate's true shift.  If the truth
         * leaves the guessed band anywhere, the iterative fixpoint below takes over for this individual. */
        if (A.phase == 0 && chunk == 1 && rho >= 0) {
            const int j = tid;
            const int b = (int) floor(rho * j) - EV_D / 2, bnext = (int) floor(rho * (j + 1)) - EV_D / 2;
            const int zavail = (int) ((A.zcount - pos) < ZW ? (A.zcount - pos) : ZW);
            int8_t m[EV_D];
            if (j < n) {
                const double xi = xp[j], sigi = sp[j], sm_ = smax[j], lo = lbs[j], hi = ubs[j];
#pragma unroll 4
                for (int d = 0; d < EV_D; ++d) {
                    const int sft = b + d, idx = 1 + 2 * j + sft;
                    int out = -1;
                    if (sft >= 0 && idx + 1 < zavail) {
                        double sg = sigi * gw[idx];
                        if (sg > sm_) sg = sm_;
                        int tt = 1;
                        for (;;) {
                            const double xn = xi + sg * zw[idx + tt];
                            if (!(xn < lo || xn > hi)) break;
                            ++tt;
                            if (idx + tt >= zavail) { tt = -1; break; }
                        }
                        if (tt > 0) { const int dn = sft + (tt - 1) - bnext; if (dn >= 0 && dn < EV_D) out = dn; }
                    }
                    m[d] = (int8_t) out;
                }
            } else {
#pragma unroll
                for (int d = 0; d < EV_D; ++d) { const int dn = b + d - bnext; m[d] = (int8_t) ((b + d >= 0 && dn >= 0 && dn < EV_D) ? dn : -1); }
            }
            const int64_t q2 = __builtin_readcyclecounter();
            c_eval += q2 - q1;
            int cb = 0;
#pragma unroll
            for (int d = 0; d < EV_D; ++d) s_tab[0][tid][d] = m[d];
            for (int off = 1; off < EV_T; off <<= 1) {
                __syncthreads();
                if (tid >= off) {
#pragma unroll
                    for (int d = 0; d < EV_D; ++d) { const int q = s_tab[cb][tid - off][d]; m[d] = (int8_t) (q < 0 ? -1 : s_tab[cb][tid][q]); }
                }
#pragma unroll
                for (int d = 0; d < EV_D; ++d) s_tab[cb ^ 1][tid][d] = m[d];
                cb ^= 1;
            }
            __syncthreads();
            const int64_t q3 = __builtin_readcyclecounter();
            c_scan += q3 - q2;
            /* prefix maps applied to the one known input: shift 0 before coordinate 0 */
            const int d0 = EV_D / 2;        /* 0 - b_0 */
            const int mine = tid == 0 ? d0 : s_tab[cb][tid - 1][d0];
            const int last = s_tab[cb][EV_T - 1][d0];
            const int bad = __syncthreads_or((mine < 0 || last < 0) ? 1 : 0);
            if (!bad) {
                resolved = true;
                shift = b + mine;
                total = ((int) floor(rho * EV_T) - EV_D / 2) + last;
                if (j < n) {
                    const int idx = 1 + 2 * j + shift;
                    const double xi = xp[j], sigi = sp[j];
                    double sg = sigi * gw[idx];
                    if (sg > smax[j]) sg = smax[j];
                    double xn;
                    int tt = 1;
                    for (;;) { xn = xi + sg * zw[idx + tt]; if (!(xn < lbs[j] || xn > ubs[j])) break; ++tt; }
                    xo[j] = xn;
                    so[j] = sigi + ALPHA * (sg - sigi);
                }
                total += 2 * n;             /* deviates after the individual's own: 2 per coordinate + redraws */
                ++rounds_total;
                __syncthreads();
            }
            c_fin += __builtin_readcyclecounter() - q3;
        }
        for (int round = 0; !resolved && round < EV_T + 2; ++round) {
            int64_t cur = pos + 1 + shift;
            int used = 0;
            bool over = false;
            for (int j = j0; j < j1; ++j) {
                const double xi = xp[j], sigi = sp[j];
                double xnew = xi;
                bool mutate = true;
                if (A.phase == 1) {
                    if (!lastsurv) xnew = xi + GAMMA * (x0c[j] - (self ? xp[j] : xk1[j]));
                    mutate = lastsurv || xnew < lbs[j] || xnew > ubs[j];
                }
                double snew = sigi;
                if (mutate) {
                    if (cur + 1 >= A.zcount) { over = true; break; }
                    const double sigmamax = smax[j];
                    double sg = sigi * gat(cur);
                    if (sg > sigmamax) sg = sigmamax;
                    int t = 1;
                    for (;;) {
                        if (cur + t >= A.zcount) { over = true; break; }
                        xnew = xi + sg * zat(cur + t);
                        if (!(xnew < lbs[j] || xnew > ubs[j])) break;
                        ++t;
                    }
                    if (over) break;
                    snew = sigi + ALPHA * (sg - sigi);
                    cur += 1 + t; used += 1 + t;
                }
                xo[j] = xnew;
                so[j] = snew;
            }
            const int nshift = ev_block_exscan(used, s_w, total);
            const int flags = (over ? 2 : 0) | (nshift != shift ? 1 : 0);
            shift = nshift;
            ++rounds_total;
            const int any = __syncthreads_or(flags);
            if (any & 2) { ranout = true; break; }
            if (!(any & 1)) break;
        }
        if (ranout) break;                  /* nothing of this individual has been written */
        {
            double *xw = A.X + (size_t) rk * ld, *sw = A.S + (size_t) rk * ld;
            for (int j = tid; j < n; j += EV_T) { xw[j] = xo[j]; sw[j] = so[j]; }
        }
        pos += 1 + total;
        if (A.phase == 0) { const double r1 = (double) (total - 2 * n) / n; rho = rho < 0 ? r1 : 0.75 * rho + 0.25 * r1; }
        if (A.phase == 1) __threadfence();  /* a later survivor may read this row as "physical row k+1" */
        __syncthreads();
        c_all += __builtin_readcyclecounter() - q0;
    }
    if (tid == 0) { A.state[0] = k; A.state[1] = pos; A.state[2] = ranout ? 1 : 0; A.state[3] += rounds_total; A.state[4] += c_stage; A.state[5] += c_eval; A.state[6] += c_scan; A.state[7] += c_fin; A.state[8] += c_all; }
}

/* ------------------------------------------------------------------------------------------------
 * launchers
 * ---------------------------------------------------------------------------------------------- */
extern "C" int nla_k_isres_init(int n, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t k_first,
                                int64_t count, const double *x0, double *X, double *S, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(isres_init_kernel, dim3((unsigned) ((count + 3) / 4)), dim3(256), 0, (hipStream_t) stream, n, ld, lb, ub, words, k_first, count, x0, X, S);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_eval(int obj, int n, int ld, const double *X, int64_t pop, int m, int p, const nla_dev_constraint *con,
                                double *F, double *PEN, double *GPEN, int32_t *FEAS, void *stream)
{
    if (pop <= 0) return 0;
    const dim3 grid((unsigned) ((pop + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t) stream;
    const double sign = nla_obj_sign(&obj);
#define CALL(O) hipLaunchKernelGGL((isres_eval_kernel<O>), grid, block, 0, st, n, ld, X, pop, m, p, con, F, PEN, GPEN, FEAS, sign)
    if (obj < 0) { CALL(-1); }                 /* constraints only: f by a user-supplied kernel (userobj.c) */
    else NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_rank_count(int64_t pop, const double *F, const double *PEN, uint64_t *elems, int32_t *sorted, void *stream)
{
    if (pop <= 0) return 0;
    /* above 2^20 individuals only `sorted` (the stable sort by f: all a generation without penalties needs, isres.c:203-204) is
     * meaningful: the packed elements have 20 bits per field, and the driver admits such populations only without constraints */
    hipLaunchKernelGGL(isres_rank_count_kernel, dim3((unsigned) ((pop + 63) / 64)), dim3(RC_W * 64), 0, (hipStream_t) stream, pop, F, PEN, elems, sorted);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_bits(const uint32_t *words, int64_t row_first, int nrows, int64_t pop, uint64_t *bits, void *stream)
{
    const int64_t popm1 = pop - 1;
    if (nrows <= 0 || popm1 <= 0) return 0;
    const int64_t rowwords = (popm1 + 63) / 64, segs = (popm1 + 2047) / 2048;
    hipLaunchKernelGGL(isres_bits_kernel, dim3((unsigned) (segs * nrows)), dim3(64), 0, (hipStream_t) stream, words, row_first, nrows, popm1, rowwords, bits);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_stochrank_gated(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits,
                                           int *ticket, uint8_t *swapped, int32_t *irank, const int *gate, uint64_t gate_g_rank0, int64_t gate_nrows,
                                           void *stream);
/* elements a unit of the pipeline hands on at a time: a launch makes pop + 2 sweeps + (this - 1) units serial ticks */
extern "C" int nla_isres_stochrank_handoff(void) { return 64; }
extern "C" int nla_k_isres_stochrank(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits,
                                     int *ticket, uint8_t *swapped, int32_t *irank, void *stream)
{
    return nla_k_isres_stochrank_gated(pop, nsweeps, streams, progress, bits, ticket, swapped, irank, NULL, 0, 0, stream);
}
extern "C" int nla_k_isres_stochrank_gated(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits,
                                           int *ticket, uint8_t *swapped, int32_t *irank, const int *gate, uint64_t gate_g_rank0, int64_t gate_nrows,
                                           void *stream)
{
    static_assert(SR_SEG_WORDS == NLA_MT_SEG_WORDS, "the pipeline's gate targets count the generator's segments");
    hipStream_t st = (hipStream_t) stream;
    if (pop <= 0) return 0;
    const int64_t units = (nsweeps + 63) / 64;
    const int64_t rowwords = (pop - 1 + 63) / 64;
    if (units > 0 && pop > 1) {
        /* the units hand over through the elements themselves (hip/isres_stochrank.h): every buffer between two units starts out unwritten */
        hipError_t e = hipMemsetAsync(streams + (size_t) pop, 0xFF, sizeof(uint64_t) * (size_t) units * (size_t) pop, st);
        if (e != hipSuccess) return (int) e;
        hipLaunchKernelGGL(isres_stochrank_kernel, dim3((unsigned) units), dim3(64), 0, st, pop, nsweeps, streams, progress, bits, rowwords, ticket,
                           swapped, gate, gate_g_rank0, gate_nrows);
        NLA_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(isres_unpack_kernel, dim3((unsigned) ((pop + 255) / 256)), dim3(256), 0, st, pop,
                       streams + (size_t) (pop > 1 ? units : 0) * (size_t) pop, irank);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_nrand(const uint32_t *words, int64_t nattempts, int64_t attempt_base, int32_t *counts, int64_t *ztotal,
                                 int64_t zbase, double *z, int64_t *zatt, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    if (nattempts <= 0) return 0;
    const int nwg = (int) ((nattempts + NR_PER_WG - 1) / NR_PER_WG);
    hipLaunchKernelGGL(isres_nrand_count_kernel, dim3(nwg), dim3(256), 0, st, words, nattempts, counts);
    hipLaunchKernelGGL(isres_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nwg, ztotal);
    hipLaunchKernelGGL(isres_nrand_write_kernel, dim3(nwg), dim3(256), 0, st, words, nattempts, attempt_base, counts, zbase, z, zatt);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_isres_evolve(int n, int ld, int phase, int64_t pop, int64_t survivors, int64_t zcount, double taup, double tau,
                                  const double *lb, const double *ub, const double *z, const int32_t *irank, double *X, double *S,
                                  double *scratch, int64_t *state, void *stream)
{
    isres_evolve_args A;
    A.n = n; A.ld = ld; A.phase = phase; A.pop = pop; A.survivors = survivors; A.zcount = zcount; A.taup = taup; A.tau = tau;
    A.lb = lb; A.ub = ub; A.z = z; A.irank = irank; A.X = X; A.S = S; A.scratch = scratch; A.state = state;
    if (n <= ISRES_EVOLVE_LDS_MAXN) {
        const size_t lds = sizeof(double) * (size_t) (9 * n + 2 * (3 * n + 64));
        static bool attr_set = false;
        if (!attr_set) {
            (void) hipFuncSetAttribute(reinterpret_cast<const void *>(isres_evolve_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256 - 2 * EV_T * EV_D);
            (void) hipGetLastError();
            attr_set = true;
        }
        hipLaunchKernelGGL(isres_evolve_lds_kernel, dim3(1), dim3(EV_T), lds, (hipStream_t) stream, A);
    } else {
        hipLaunchKernelGGL(isres_evolve_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, A);
    }
    NLA_LAUNCH_CHECK();
    return 0;
}
