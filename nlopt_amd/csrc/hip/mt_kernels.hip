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
 * 64-bit words, zeroed by the launcher's caller).  64 consecutive steps are one ballot; lane 0 ORs its pieces into the one to
 * three words they fall into (row ends do not fall on word boundaries).  No 4 B/word written, no 4 B/word read back — and,
 * with no word buffer to size passes by, ALL segments of a generation's ranking run in one launch (7800 wavefronts at
 * pop = 5e4 instead of five passes of 1678), which is what this latency-bound generator needs.
 * ---------------------------------------------------------------------------------------------- */
/* lane 0: OR the bits of `left` consecutive ranking steps starting at step s (bit i of bm = step s + i) into the rows of `bits`.
 * (row, j) of a step follow from those of the previous call by addition — the steps of a wavefront are consecutive — so the
 * one 64-bit division is paid once per wavefront */
struct mt_rowpos { int64_t s, row, j; int ready; };
__device__ __forceinline__ void mt_scatter_bits(mt_rowpos &P, int64_t s, unsigned long long bm, int left, int64_t popm1, int64_t rowwords,
                                                unsigned long long *__restrict__ bits)
{
    int64_t row, j;
    if (!P.ready) { row = s / popm1; j = s - row * popm1; P.ready = 1; }
    else { row = P.row; j = P.j + (s - P.s); while (j >= popm1) { j -= popm1; ++row; } }
    P.s = s; P.row = row; P.j = j;
    while (left > 0) {
        const int take = (int) (popm1 - j < (int64_t) left ? popm1 - j : (int64_t) left);      /* steps left in this row */
        const unsigned long long piece = take >= 64 ? bm : (bm & ((1ULL << take) - 1ULL));
        unsigned long long *w = bits + (size_t) row * (size_t) rowwords + (size_t) (j >> 6);
        const int sh = (int) (j & 63);
        if (piece << sh) atomicOr(w, piece << sh);
        if (sh && (piece >> (64 - sh))) atomicOr(w + 1, piece >> (64 - sh));
        bm = take >= 64 ? 0ULL : bm >> take;
        left -= take;
        j += take;
        if (j >= popm1) { j = 0; ++row; }
    }
}

__global__ __launch_bounds__(64) void mt_rankbits_kernel(const uint32_t *__restrict__ seg_states, uint64_t seg_first, uint64_t g_rank0,
                                                          uint64_t g_first, uint64_t count, int64_t popm1, int64_t rowwords,
                                                          unsigned long long *__restrict__ bits)
{
    __shared__ uint32_t mt[MT_N];
    const int lane = threadIdx.x;
    const uint64_t seg = seg_first + blockIdx.x;
    const uint64_t g_end = g_first + count;
    const uint64_t g0 = seg * NLA_MT_SEG_WORDS;
    const int par = (int) (g_rank0 & 1);          /* blocks start at even word indices: a step is (2k + par, 2k + par + 1) of its block */
    mt_rowpos P = { 0, 0, 0, 0 };

    for (int i = lane; i < MT_N; i += 64) mt[i] = seg_states[(size_t) blockIdx.x * MT_N + i];
    __syncthreads();

    for (int r = 0; r < NLA_MT_SEG_REGENS; ++r) {
        const uint64_t gb = g0 + (uint64_t) r * MT_N;
        uint32_t w_last = 0;
        if (gb >= g_end) break;
        const bool touches = gb + MT_N > g_first;
        if (touches) {
            const int npairs = par ? MT_N / 2 - 1 : MT_N / 2;      /* pairs wholly inside the block */
            for (int it = 0; it < (MT_N / 2 + 63) / 64; ++it) {
                const int k = it * 64 + lane;
                const uint64_t g = gb + 2 * (uint64_t) k + (uint64_t) par;
                const bool valid = k < npairs && g >= g_first && g + 1 < g_end;
                bool b = false;
                if (valid) b = nla_urand_from(0., 1., mt_temper(mt[2 * k + par]), mt_temper(mt[2 * k + par + 1])) < 0.45;      /* PF, isres.c:72 */
                const unsigned long long vm = __ballot(valid), bm = __ballot(b);
                if (lane == 0 && vm) {                              /* the valid lanes are one run of consecutive steps */
                    const int lo = __builtin_ctzll(vm);
                    const int64_t s = (int64_t) ((gb + 2 * (uint64_t) (it * 64 + lo) + (uint64_t) par - g_rank0) >> 1);
                    mt_scatter_bits(P, s, bm >> lo, __builtin_popcountll(vm), popm1, rowwords, bits);
                }
            }
            if (par) w_last = mt_temper(mt[MT_N - 1]);             /* first word of the step that straddles into the next block */
        }
        __syncthreads();
        /* regenerate in place (as mt_generate_kernel) */
        for (int k = lane; k < MT_N - MT_M; k += 64) { uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k + MT_M]); mt[k] = v; }
        __syncthreads();
        for (int k = MT_N - MT_M + lane; k < 2 * (MT_N - MT_M); k += 64) { uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]); mt[k] = v; }
        __syncthreads();
        for (int k = 2 * (MT_N - MT_M) + lane; k < MT_N - 1; k += 64) { uint32_t v = mt_twist(mt[k], mt[k + 1], mt[k - (MT_N - MT_M)]); mt[k] = v; }
        __syncthreads();
        if (lane == 0) mt[MT_N - 1] = mt_twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
        __syncthreads();
        if (par && touches && lane == 0) {
            const uint64_t g = gb + MT_N - 1;                       /* (word 623 of block r, word 0 of block r + 1) */
            if (g >= g_first && g + 1 < g_end) {
                const bool b = nla_urand_from(0., 1., w_last, mt_temper(mt[0])) < 0.45;
                mt_scatter_bits(P, (int64_t) ((g - g_rank0) >> 1), b ? 1ULL : 0ULL, 1, popm1, rowwords, bits);
            }
        }
    }
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

extern "C" int nla_k_mt_rankbits(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first,
                                 uint64_t count, int64_t popm1, int64_t rowwords, uint64_t *bits, void *stream)
{
    if (nseg <= 0 || count == 0 || popm1 <= 0) return 0;
    hipLaunchKernelGGL(mt_rankbits_kernel, dim3(nseg), dim3(64), 0, (hipStream_t) stream, seg_states, seg_first, g_rank0, g_first, count,
                       popm1, rowwords, (unsigned long long *) bits);
    NLA_LAUNCH_CHECK();
    return 0;
}
