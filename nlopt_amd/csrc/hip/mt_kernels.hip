/* mt_kernels.hip — the MT19937 word stream on gfx950.
 *
 * The reference draws every random number from one serial generator (src/util/mt19937ar.c).
 * Here the same word stream is produced in parallel: the stream is cut into segments of
 * NLA_MT_SEG_REGENS regenerations (638,976 words); the block array at the start of every segment
 * is obtained by GF(2) jump-ahead (mt_jump_kernel, polynomials from ../mt_host.c) in log2(#seg)
 * doubling rounds, then one wavefront per segment regenerates its 1024 blocks out of LDS and
 * writes the tempered words with coalesced 256-byte stores (mt_generate_kernel).
 *
 * Bound: integer VALU + LDS latency (3 dependent phases of 227 words per regeneration,
 * mt19937ar.c:108-117); HBM traffic is 4 B/word written once.
 */
#include "dev_common.h"
#include "../../../include/nlopt_amd.h"

#define MT_N 624
#define MT_M 397
#define MT_DEG 19937

__device__ __forceinline__ uint32_t mt_twist(uint32_t hi, uint32_t lo, uint32_t far)
{
    uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y)       /* mt19937ar.c:125-128 */
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* one wavefront (= one 64-thread workgroup) per segment */
__global__ __launch_bounds__(64) void mt_generate_kernel(const uint32_t *__restrict__ seg_states, uint64_t seg_first,
                                                          uint64_t g_first, uint64_t count, uint32_t *__restrict__ out)
{
    __shared__ uint32_t mt[MT_N];
    const int lane = threadIdx.x;
    const uint64_t seg = seg_first + blockIdx.x;
    const uint64_t g_end = g_first + count;
    const uint64_t g0 = seg * NLA_MT_SEG_WORDS;

    for (int i = lane; i < MT_N; i += 64) mt[i] = seg_states[(size_t) blockIdx.x * MT_N + i];
    __syncthreads();

    for (int r = 0; r < NLA_MT_SEG_REGENS; ++r) {
        const uint64_t gb = g0 + (uint64_t) r * MT_N;
        if (gb >= g_end) break;
        if (gb + MT_N > g_first) {
            for (int i = lane; i < MT_N; i += 64) {
                const uint64_t g = gb + i;
                if (g >= g_first && g < g_end) out[g - g_first] = mt_temper(mt[i]);
            }
        }
        /* regenerate in place: three phases whose inputs are all older than their outputs */
        for (int k = lane; k < MT_N - MT_M; k += 64) {                       /* 0 .. 226 */
            uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]);
            mt[k] = v;      /* lanes read before any lane of this step writes (lockstep); k+1 of
                               the last lane of a step is written only by the next step */
        }
        __syncthreads();
        for (int k = MT_N - MT_M + lane; k < 2 * (MT_N - MT_M); k += 64) {     /* 227 .. 453 */
            uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
            mt[k] = v;
        }
        __syncthreads();
        for (int k = 2 * (MT_N - MT_M) + lane; k < MT_N - 1; k += 64) {        /* 454 .. 622 */
            uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
            mt[k] = v;
        }
        __syncthreads();
        if (lane == 0) mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        __syncthreads();
    }
}

/* the same with the segment length as an argument (nla_k_mt_generate_seg; "amd_mlsl_seg_regens"): a stream user that needs few words per
 * fill — MLSL draws 8 M words per iteration, 13 segments of the default length: 13 wavefronts walking 1024 regenerations each, ~3 ms of
 * latency on an idle device — can cut its stream into shorter segments and get as many more wavefronts.  A kernel of its own so that
 * mt_generate_kernel's loop keeps its compile-time bound (and its machine code). */
__global__ __launch_bounds__(64) void mt_generate_seg_kernel(const uint32_t *__restrict__ seg_states, uint64_t seg_first,
                                                              uint64_t g_first, uint64_t count, uint32_t *__restrict__ out, int seg_regens)
{
    __shared__ uint32_t mt[MT_N];
    const int lane = threadIdx.x;
    const uint64_t seg = seg_first + blockIdx.x;
    const uint64_t g_end = g_first + count;
    const uint64_t g0 = seg * ((uint64_t) MT_N * (uint64_t) seg_regens);

    for (int i = lane; i < MT_N; i += 64) mt[i] = seg_states[(size_t) blockIdx.x * MT_N + i];
    __syncthreads();

    for (int r = 0; r < seg_regens; ++r) {
        const uint64_t gb = g0 + (uint64_t) r * MT_N;
        if (gb >= g_end) break;
        if (gb + MT_N > g_first) {
            for (int i = lane; i < MT_N; i += 64) {
                const uint64_t g = gb + i;
                if (g >= g_first && g < g_end) out[g - g_first] = mt_temper(mt[i]);
            }
        }
        for (int k = lane; k < MT_N - MT_M; k += 64) {                       /* 0 .. 226 (the three phases of mt_generate_kernel) */
            uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]);
            mt[k] = v;
        }
        __syncthreads();
        for (int k = MT_N - MT_M + lane; k < 2 * (MT_N - MT_M); k += 64) {     /* 227 .. 453 */
            uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
            mt[k] = v;
        }
        __syncthreads();
        for (int k = 2 * (MT_N - MT_M) + lane; k < MT_N - 1; k += 64) {        /* 454 .. 622 */
            uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]);
            mt[k] = v;
        }
        __syncthreads();
        if (lane == 0) mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        __syncthreads();
    }
}

/* dst = block array J words after src, g = t^J mod phi:  y_j = XOR_{i : g_i} x[i + j], j < 624, over the untempered word
 * sequence x[0 .. 19937 + 623] that continues src.  One 640-thread workgroup per state.  The sequence is walked in WINDOWS of
 * JUMP_CHUNK polynomial bits: the window holds x[c .. c + JUMP_CHUNK + 624) — thread j's terms of the chunk — and is extended
 * from its own last 624 words for the next chunk (x[m+624] = x[m+397] ^ A(x[m], x[m+1]): 227 independent words per step).
 * A window of FOUR states is 43 KB of LDS (the whole sequence of one state was 82 KB): three workgroups = twelve states per CU, and
 * the scalar walk over the polynomial's 10^4 set bits is paid once for the four states' reads.  g is wave-uniform, so the bit test is a scalar branch and the LDS reads of a
 * wavefront are 64 consecutive words. */
#define JUMP_CHUNK 2048                          /* polynomial bits per window: a multiple of 64 */
#define JUMP_WIN (JUMP_CHUNK + MT_N)
#define JUMP_STATES 4                            /* states per workgroup: the scalar walk over the polynomial's set bits is shared by their reads */
__global__ __launch_bounds__(640) void mt_jump_kernel(const uint64_t *__restrict__ poly, const uint32_t *__restrict__ src,
                                                       uint32_t *__restrict__ dst, int count)
{
    __shared__ uint32_t x[JUMP_STATES][JUMP_WIN];
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * JUMP_STATES;
    uint32_t acc[JUMP_STATES];
#pragma unroll
    for (int q = 0; q < JUMP_STATES; ++q) {
        acc[q] = 0;
        /* (a workgroup past the end of the list repeats the last state: same work, nothing stored) */
        const int sq = s0 + q < count ? s0 + q : count - 1;
        if (tid < MT_N) x[q][tid] = src[(size_t) sq * MT_N + tid];
    }
    __syncthreads();
    for (int c = 0; c < NLA_MT_POLYWORDS * 64; c += JUMP_CHUNK) {
        /* extend the windows: words 624 .. JUMP_WIN-1 from the 624 before them, 227 at a time */
        for (int base = MT_N; base < JUMP_WIN; base += (MT_N - MT_M)) {
            const int j = base + tid;
            if (tid < (MT_N - MT_M) && j < JUMP_WIN) {
#pragma unroll
                for (int q = 0; q < JUMP_STATES; ++q) x[q][j] = mt_twist(x[q][j - MT_N], x[q][j - MT_N + 1], x[q][j - (MT_N - MT_M)]);
            }
            __syncthreads();
        }
        if (tid < MT_N) {
            const int w1 = (c + JUMP_CHUNK) / 64 < NLA_MT_POLYWORDS ? (c + JUMP_CHUNK) / 64 : NLA_MT_POLYWORDS;
            for (int w = c / 64; w < w1; ++w) {
                uint64_t gw = poly[w];                       /* uniform -> scalar load */
                const int off = (w * 64 - c) + tid;
                while (gw) {
                    const int bb = __builtin_ctzll(gw);
                    gw &= gw - 1;
#pragma unroll
                    for (int q = 0; q < JUMP_STATES; ++q) acc[q] ^= x[q][off + bb];
                }
            }
        }
        __syncthreads();
        /* the next window starts JUMP_CHUNK words further on: its first 624 words are this window's last 624 */
        uint32_t keep[JUMP_STATES];
#pragma unroll
        for (int q = 0; q < JUMP_STATES; ++q) keep[q] = tid < MT_N ? x[q][JUMP_CHUNK + tid] : 0;
        __syncthreads();
        if (tid < MT_N) {
#pragma unroll
            for (int q = 0; q < JUMP_STATES; ++q) x[q][tid] = keep[q];
        }
        __syncthreads();
    }
    if (tid < MT_N) {
#pragma unroll
        for (int q = 0; q < JUMP_STATES; ++q) if (s0 + q < count) dst[(size_t) (s0 + q) * MT_N + tid] = acc[q];
    }
}

/* ------------------------------------------------------------------------------------------------
 * The ranking's uniforms never need to exist as words in memory: ISRES asks of u = nlopt_urand(0,1) (isres.c:210, one per
 * ranking step, two stream words each) only the bit u < PF.  One wavefront per segment regenerates its blocks as
 * mt_generate_kernel does and turns every block's 312 steps into bits on the spot: step s of the ranking (s = sweep * (pop-1) +
 * position, counted from the ranking's first word g_rank0) sets bit (position) of row (sweep) of `bits` (rows of `rowwords`
 * 64-bit words, zeroed by the launcher's caller).  No 4 B/word written, no 4 B/word read back, all segments of a generation's
 * ranking in one launch (7800 wavefronts at pop = 5e4).
 *
 * Round 5 (the kernel of rounds 2-4 cost ~700 instructions per regeneration, a third of them scalar 64-bit divisions and a
 * lane-0 scatter of every ballot through three atomic ORs; 9.6 ms alone at config 3, 17 ms beside the ranking pipeline):
 *  - the bit is decided in integers on the FIRST word alone: u = ((a >> 5) 2^26 + (b >> 6)) / 2^53 < 0.45  <=>  (a >> 5, b >> 6) <
 *    (0x3999999, 0x2666667) lexicographically (0.45 * 2^53 = 4053239664633446.5 exactly; nla_urand_from's expression is exact in
 *    double, checked value by value in tests/test_host_logic.py), so the second word is tempered and looked at only when a >> 5
 *    equals 0x3999999 (once in 2^27 steps) — half the tempering, no conversion to double;
 *  - 64 consecutive steps are one ballot; the ballots are strung together in a wavefront-uniform 64-bit accumulator and leave as whole
 *    words with plain (device-scope) stores — an atomic OR only for the first and last word of a wavefront's range, which it may
 *    share with its neighbours; (row, column) advance by addition, one 64-bit division per wavefront;
 *  - no workgroup barriers: the workgroup is one wavefront and a wavefront's LDS operations execute in the order they are issued;
 *  - IN-ORDER GATES: segments are claimed by ticket (front first), and a wavefront that has produced everything it owes to a block
 *    of 64 sweeps adds 1 to gate[block] (its stores landed first).  Block c is complete when gate[c] equals the number of segments
 *    that intersect its words (nla_rankbits_gate_target): the ranking pipeline's unit c (hip/isres_stochrank.h) waits for exactly
 *    that and starts while later blocks are still being generated — one launch, no flag kernels in between.
 * ---------------------------------------------------------------------------------------------- */
#define RB_T_HI 0x3999999u               /* (a >> 5) below this: u < 0.45; above: not; equal: (b >> 6) < RB_T_LO decides */
#define RB_T_LO 0x2666667u

/* the output cursor of a wavefront (every member wavefront-uniform) */
struct rb_out {
    unsigned long long acc;              /* bits gathered for word accw of row `row` */
    long long row;
    int col;                             /* next step = (row, col) */
    int accw;                            /* -1: nothing gathered */
    int shared;                          /* the word may also hold bits this store does not carry (first word of the range, a word flushed early): OR it in */
};
__device__ __forceinline__ void rb_flush(rb_out &o, long long rowwords, unsigned long long *__restrict__ bits, int lane)
{
    if (o.accw >= 0 && lane == 0) {
        unsigned long long *w = bits + (size_t) o.row * (size_t) rowwords + (size_t) o.accw;
        if (o.shared) { if (o.acc) __hip_atomic_fetch_or(w, o.acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else __hip_atomic_store(w, o.acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    o.accw = -1; o.acc = 0; o.shared = 0;
}
/* append the bits of `cnt` consecutive steps (bit i of bm = i-th of them) */
__device__ __forceinline__ void rb_put(rb_out &o, unsigned long long bm, int cnt, int popm1, long long rowwords, unsigned long long *__restrict__ bits, int lane)
{
    while (cnt > 0) {
        const int room = popm1 - o.col;
        const int take = cnt < room ? cnt : room;
        const int sh = o.col & 63, w = o.col >> 6;
        const unsigned long long piece = take >= 64 ? bm : (bm & ((1ULL << take) - 1ULL));
        if (o.accw != w) { const int keep = o.accw < 0 ? o.shared : 0; rb_flush(o, rowwords, bits, lane); o.accw = w; o.shared = keep; }
        o.acc |= piece << sh;
        if (sh + take >= 64) {                                   /* word w is complete */
            rb_flush(o, rowwords, bits, lane);
            if (sh + take > 64) { o.accw = w + 1; o.acc = piece >> (64 - sh); }
        }
        o.col += take;
        bm = take >= 64 ? 0ULL : bm >> take;
        cnt -= take;
        if (o.col >= popm1) { rb_flush(o, rowwords, bits, lane); o.col = 0; ++o.row; }
    }
}

__global__ __launch_bounds__(64) void mt_rankbits_kernel(const uint32_t *__restrict__ seg_states, uint64_t seg_first, uint64_t g_rank0,
                                                          uint64_t g_first, uint64_t count, int64_t popm1_, int64_t rowwords,
                                                          unsigned long long *__restrict__ bits, int *__restrict__ gate, int *__restrict__ ticket)
{
    extern __shared__ uint32_t rb_pad[];          /* dynamic LDS asked for by the launcher only to bound the wavefronts per CU */
    __shared__ uint32_t mt[MT_N + 64];            /* (+ one scratch slot per lane: the regeneration's branch-free stores) */
    __shared__ int s_blk;
    const int lane = threadIdx.x;
    (void) rb_pad;
    /* front segment first: a block of sweeps is complete when ALL its segments are, so they must not be started in any other order */
    if (ticket) { if (lane == 0) s_blk = atomicAdd(ticket, 1); __syncthreads(); } else if (lane == 0) s_blk = (int) blockIdx.x;
    if (!ticket) __syncthreads();
    const int blk = __builtin_amdgcn_readfirstlane(s_blk);
    const uint64_t seg = seg_first + (uint64_t) blk;
    const uint64_t g_end = g_first + count;
    const uint64_t g0 = seg * NLA_MT_SEG_WORDS;
    const int popm1 = (int) popm1_;
    const int par = (int) (g_rank0 & 1);          /* blocks start at even word indices: a step is (2k + par, 2k + par + 1) of its block */
    /* this wavefront's words, and the blocks of 64 sweeps (2 * 64 * popm1 words each, counted from g_rank0) they belong to */
    const uint64_t lo_w = g0 > g_first ? g0 : g_first, hi_w = g0 + NLA_MT_SEG_WORDS < g_end ? g0 + NLA_MT_SEG_WORDS : g_end;
    const uint64_t bw = 128ULL * (uint64_t) popm1;
    long long next_c = 0, last_c = -1;
    uint64_t next_end = 0;
    if (hi_w > lo_w) { next_c = (long long) ((lo_w - g_rank0) / bw); last_c = (long long) ((hi_w - 1 - g_rank0) / bw); next_end = g_rank0 + (uint64_t) (next_c + 1) * bw; }
    rb_out o = { 0ULL, 0, 0, -1, 1 };             /* (the first word of the range may be shared with the previous segment's wavefront) */
    if (hi_w > lo_w) {
        /* (row, column) of this wavefront's first step — the one 64-bit division it pays: its steps are consecutive from there.  A step
         * belongs to the wavefront that holds its FIRST word; first words sit at g = g_rank0 (mod 2) */
        const uint64_t gs = lo_w + ((lo_w ^ g_rank0) & 1ULL);
        const long long s0 = (long long) ((gs - g_rank0) >> 1);
        o.row = s0 / popm1; o.col = (int) (s0 - o.row * popm1);
    }

    for (int i = lane; i < MT_N; i += 64) mt[i] = seg_states[(size_t) blk * MT_N + i];
    asm volatile("" ::: "memory");

    for (int r = 0; r < NLA_MT_SEG_REGENS; ++r) {
        const uint64_t gb = g0 + (uint64_t) r * MT_N;
        uint32_t w_last = 0;
        if (gb >= g_end) break;
        const bool touches = gb + MT_N > g_first;
        if (touches) {
            const int npairs = par ? MT_N / 2 - 1 : MT_N / 2;      /* pairs wholly inside the block */
            constexpr int NIT = (MT_N / 2 + 63) / 64;
            /* all of the block's first words in flight at once, one wait */
            uint32_t a[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) { const int k = it * 64 + lane; a[it] = mt[k < npairs ? 2 * k + par : 0]; }
            const bool full = gb >= g_first && gb + MT_N + 1 <= g_end;      /* (uniform) every pair of the block is a step of the range */
            bool low[NIT];
            unsigned long long eqm = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const uint32_t hi = mt_temper(a[it]) >> 5;
                low[it] = hi < RB_T_HI;
                eqm |= __ballot(hi == RB_T_HI);
            }
            if (eqm) {                                             /* once in 2^21 blocks: the second word decides */
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int k = it * 64 + lane;
                    if (k < npairs && (mt_temper(a[it]) >> 5) == RB_T_HI) low[it] = (mt_temper(mt[2 * k + par + 1]) >> 6) < RB_T_LO;
                }
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int k = it * 64 + lane;
                const uint64_t g = gb + 2 * (uint64_t) k + (uint64_t) par;
                const bool valid = k < npairs && (full || (g >= g_first && g + 1 < g_end));
                const unsigned long long vm = __ballot(valid), bm = __ballot(valid && low[it]);
                if (vm) {                                           /* the valid lanes are one run of consecutive steps */
                    const int lo = __builtin_ctzll(vm), cnt = __builtin_popcountll(vm);
                    if (o.accw == (o.col >> 6) && o.col + cnt < popm1) {
                        /* the common case: the run continues the word being gathered and stays inside the row — at most one word completes */
                        const int sh = o.col & 63;
                        const unsigned long long run = bm >> lo;
                        o.acc |= run << sh;
                        if (sh + cnt >= 64) {
                            const int w = o.accw;
                            rb_flush(o, rowwords, bits, lane);
                            o.accw = w + 1; o.acc = sh ? run >> (64 - sh) : 0ULL;
                        }
                        o.col += cnt;
                    } else rb_put(o, bm >> lo, cnt, popm1, rowwords, bits, lane);
                }
            }
            if (par) w_last = mt_temper(mt[MT_N - 1]);             /* first word of the step that straddles into the next block */
        }
        asm volatile("" ::: "memory");
        /* regenerate in place (as mt_generate_kernel; one wavefront: its LDS operations execute in program order).  Every phase: all its
         * loads, then all its stores — the loads of a phase only read words no store of the same phase has written yet */
        {
            /* branch-free: a lane past the end of a phase recomputes the phase's last word from clamped indices and stores it into a slot
             * of its own behind the state (mt[MT_N + lane]) */
            constexpr int D = MT_N - MT_M;                          /* 227 */
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k0 = lane + 64 * q, k = k0 < D ? k0 : D - 1; v[q] = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]); }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k0 = lane + 64 * q; mt[k0 < D ? k0 : MT_N + lane] = v[q]; }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k0 = D + lane + 64 * q, k = k0 < 2 * D ? k0 : 2 * D - 1; v[q] = mt_twist(mt[k], mt[k + 1], mt[k - D]); }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k0 = D + lane + 64 * q; mt[k0 < 2 * D ? k0 : MT_N + lane] = v[q]; }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int k0 = 2 * D + lane + 64 * q, k = k0 < MT_N - 1 ? k0 : MT_N - 2; v[q] = mt_twist(mt[k], mt[k + 1], mt[k - D]); }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int k0 = 2 * D + lane + 64 * q; mt[k0 < MT_N - 1 ? k0 : MT_N + lane] = v[q]; }
            asm volatile("" ::: "memory");
        }
        if (lane == 0) mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        asm volatile("" ::: "memory");
        if (par && touches) {
            const uint64_t g = gb + MT_N - 1;                       /* (word 623 of block r, word 0 of block r + 1) */
            if (g >= g_first && g + 1 < g_end) {
                const uint32_t hi = w_last >> 5;
                bool b = hi < RB_T_HI;
                if (hi == RB_T_HI) b = (mt_temper(mt[0]) >> 6) < RB_T_LO;
                rb_put(o, b ? 1ULL : 0ULL, 1, popm1, rowwords, bits, lane);
            }
        }
        /* every word of this wavefront below gb + 624 has been turned into bits: the blocks of sweeps that end there have all it owes them */
        if (gate && next_c <= last_c && next_end <= gb + MT_N) {
            const int keep = o.accw;                                /* (a word flushed before it is full: what follows is ORed in) */
            const long long krow = o.row;
            rb_flush(o, rowwords, bits, lane);
            if (keep >= 0 && krow == o.row) { o.accw = keep; o.shared = 1; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        /* the stores have landed before the count says so */
            while (next_c <= last_c && next_end <= gb + MT_N) {
                if (lane == 0) __hip_atomic_fetch_add(gate + next_c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ++next_c; next_end += bw;
            }
        }
    }
    o.shared = 1;                                                   /* the last word may be shared with the next segment's wavefront */
    rb_flush(o, rowwords, bits, lane);
    if (gate && next_c <= last_c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (; next_c <= last_c; ++next_c) if (lane == 0) __hip_atomic_fetch_add(gate + next_c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

/* how many wavefronts of mt_rankbits_kernel add 1 to gate[c]: the segments that intersect the words of sweeps 64 c .. min(64 c + 64, nrows) - 1 */
extern "C" int nla_rankbits_gate_target(uint64_t g_rank0, int64_t popm1, int64_t nrows, int64_t c)
{
    const uint64_t b0 = g_rank0 + 128ULL * (uint64_t) popm1 * (uint64_t) c;
    const int64_t r1 = 64 * (c + 1) < nrows ? 64 * (c + 1) : nrows;
    const uint64_t b1 = g_rank0 + 2ULL * (uint64_t) popm1 * (uint64_t) r1;
    if (b1 <= b0) return 0;
    return (int) ((b1 - 1) / NLA_MT_SEG_WORDS - b0 / NLA_MT_SEG_WORDS + 1);
}

extern "C" int nla_k_mt_jump(const uint64_t *poly, const uint32_t *src_states, uint32_t *dst_states, int count, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(mt_jump_kernel, dim3((unsigned) ((count + JUMP_STATES - 1) / JUMP_STATES)), dim3(640), 0, (hipStream_t) stream, poly, src_states, dst_states, count);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_mt_generate(const uint32_t *seg_states, uint64_t seg_first, int nseg,
                                 uint64_t g_first, uint64_t count, uint32_t *out, void *stream)
{
    if (nseg <= 0 || count == 0) return 0;
    hipLaunchKernelGGL(mt_generate_kernel, dim3(nseg), dim3(64), 0, (hipStream_t) stream,
                       seg_states, seg_first, g_first, count, out);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_mt_generate_seg(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_first, uint64_t count,
                                     uint32_t *out, int seg_regens, void *stream)
{
    if (seg_regens < 1 || seg_regens > NLA_MT_SEG_REGENS || (seg_regens & (seg_regens - 1))) return (int) hipErrorInvalidValue;
    if (nseg <= 0 || count == 0) return 0;
    hipLaunchKernelGGL(mt_generate_seg_kernel, dim3(nseg), dim3(64), 0, (hipStream_t) stream, seg_states, seg_first, g_first, count, out, seg_regens);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_mt_rankbits_gated(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first,
                                       uint64_t count, int64_t popm1, int64_t rowwords, uint64_t *bits, int *gate, int *ticket, int waves_per_cu,
                                       void *stream)
{
    if (nseg <= 0 || count == 0 || popm1 <= 0) return 0;
    if (popm1 > 0x7fffffff) return (int) hipErrorInvalidValue;
    /* waves_per_cu > 0: dynamic LDS that nobody touches bounds how many wavefronts a CU holds (160 KB per CU), so that the segments are
     * worked off front to back in waves of that size instead of all at once (in-order gates: the first blocks complete early) */
    size_t pad = 0;
    if (waves_per_cu > 0 && waves_per_cu < 32) {
        pad = (size_t) (160 * 1024) / (size_t) waves_per_cu;
        pad = pad > 4096 ? pad - 4096 : 0;                   /* (the kernel's own 2.5 KB + allocation granularity) */
        if (pad > 64 * 1024 - 4096) pad = 64 * 1024 - 4096;  /* (static + dynamic <= 64 KB without the opt-in attribute) */
    }
    hipLaunchKernelGGL(mt_rankbits_kernel, dim3(nseg), dim3(64), pad, (hipStream_t) stream, seg_states, seg_first, g_rank0, g_first, count,
                       popm1, rowwords, (unsigned long long *) bits, gate, ticket);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_mt_rankbits(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first,
                                 uint64_t count, int64_t popm1, int64_t rowwords, uint64_t *bits, void *stream)
{
    return nla_k_mt_rankbits_gated(seg_states, seg_first, nseg, g_rank0, g_first, count, popm1, rowwords, bits, NULL, NULL, 0, stream);
}
