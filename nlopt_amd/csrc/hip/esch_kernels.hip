/* esch_kernels.hip — NLOPT_GN_ESCH (src/algs/esch/esch.c, C. H. da Silva Santos' evolutionary strategy with Cauchy
 * mutation) on gfx950.  SURVEY.md §8f.1: the fourth stochastic population algorithm behind the same boundary
 * (optimize.c:946-949), reusing the MT19937 word stream, the device objectives and the evaluation kernel.
 *
 * Reference loops replaced:
 *   randcauchy (esch.c:28-50)        a rejection loop on nlopt_urand: u is redrawn until tan(pi (u - 1/2)) lies in
 *       [-5, 5].  In the initialisation every draw is a randcauchy, so the stream is a sequence of 2-word attempts
 *       that are accepted or not independently: the accepted values, in order, are what the rows consume —
 *       count / scan / write compaction (esch_cauchy_*), then esch_fill_rows scales them into the box.
 *   crossover (:192-203)             three iurand per offspring; one workgroup copies the two parent pieces.
 *   point mutations (:207-218)       (no n)/10 steps, each iurand(no), iurand(n), randcauchy.  A step's length in
 *       stream words (2 + 2 per attempt) depends on the words it meets, so step c+1 starts where step c ends — a
 *       chain — but "where does a step that starts at word p end" is a pure function next(p) of the stream.  The
 *       segment is cut into blocks; for every block and every possible entry offset a lane walks next() through
 *       the block (esch_mut_scan: exit offset + steps taken), one thread chains the blocks (esch_mut_chain), and
 *       the blocks then replay from their true entries, each step writing its element with "last step wins"
 *       (esch_mut_mark / esch_mut_apply), exactly as the serial loop's later writes overwrite earlier ones.
 *   selection (:243-251)             parents and offspring sorted together by fitness with a STABLE sort (glibc's
 *       qsort_r is a merge sort): hipcub's radix sort on order-preserving 64-bit keys.
 * Individuals are (slot -> physical row) like the reference's {pointer, fitness} structs: selection permutes the
 * slots, rows never move.
 */
#include "dev_common.h"
#include <hipcub/hipcub.hpp>
#include "../../../include/nlopt_amd.h"

#define ESCH_PI 3.14159265358979323846
#define ESCH_BAND 10.0
#define ESCH_BLOCK 4096                /* stream words per mutation block */
#define ESCH_ENTRIES 64                /* entry offsets tried per block (a step is shorter than this unless ~30 attempts in a row fail) */

/* one attempt of randcauchy from two stream words: accepted? and the value folded to [0, 1] (valor before scaling) */
__device__ __forceinline__ bool esch_attempt(uint32_t w0, uint32_t w1, double &v01)
{
    const double u = nla_urand_from(0., 1., w0, w1);
    const double c = 1.0 * tan((u - 0.5) * ESCH_PI) + 0.0;
    if ((c < 0.0 - (ESCH_BAND * 0.5)) || (c > 0.0 + (ESCH_BAND * 0.5))) return false;
    const double f = (c < 0) ? -c : c + (ESCH_BAND * 0.5);
    v01 = f / ESCH_BAND;
    return true;
}

/* ---- initialisation: accepted values of a run of attempts, in order ------------------------------------------------ */
#define ESCH_PER_WG 1024
__global__ __launch_bounds__(256) void esch_cauchy_count_kernel(const uint32_t *__restrict__ words, int64_t nattempts, int32_t *__restrict__ counts)
{
    __shared__ int s_w[4];
    int c = 0;
    const int64_t a0 = (int64_t) blockIdx.x * ESCH_PER_WG;
    for (int q = threadIdx.x; q < ESCH_PER_WG; q += 256) {
        const int64_t a = a0 + q;
        double v;
        if (a < nattempts && esch_attempt(words[2 * a], words[2 * a + 1], v)) ++c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

/* exclusive scan of counts[0..nwg) in place by one workgroup; *total += sum */
__global__ __launch_bounds__(1024) void esch_scan_kernel(int32_t *__restrict__ counts, int nwg, int64_t *__restrict__ total)
{
    __shared__ long long s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nwg + 1023) / 1024;
    const int b0 = tid * per, b1 = b0 + per < nwg ? b0 + per : nwg;
    long long sum = 0;
    for (int b = b0; b < b1; ++b) sum += counts[b];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long v = tid >= off ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    long long run = s_part[tid] - sum;
    for (int b = b0; b < b1; ++b) { const int c = counts[b]; counts[b] = (int32_t) run; run += c; }
    if (tid == 1023) *total += s_part[1023];
}

__global__ __launch_bounds__(256) void esch_cauchy_write_kernel(const uint32_t *__restrict__ words, int64_t nattempts, int64_t attempt_base,
                                                                const int32_t *__restrict__ offs, int64_t vbase, int64_t vcap,
                                                                double *__restrict__ v, int64_t *__restrict__ vatt)
{
    __shared__ int s_w[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t a0 = (int64_t) blockIdx.x * ESCH_PER_WG;
    int64_t base = vbase + offs[blockIdx.x];
    for (int q0 = 0; q0 < ESCH_PER_WG; q0 += 256) {          /* attempts in order: wave-major within a pass of 256 */
        const int64_t a = a0 + q0 + threadIdx.x;
        double val = 0;
        const bool ok = a < nattempts && esch_attempt(words[2 * a], words[2 * a + 1], val);
        const unsigned long long bal = __ballot(ok);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int64_t o = base + __popcll(bal & ((1ull << lane) - 1));
        for (int w = 0; w < wave; ++w) o += s_w[w];
        if (ok && o < vcap) { v[o] = val; vatt[o] = attempt_base + a; }
        base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
}

/* element e = (individual, item) of the initial populations := lb + (ub - lb) * valor  (esch.c:49, rows in draw order) */
__global__ __launch_bounds__(256) void esch_fill_rows_kernel(int n, int ld, const double *__restrict__ lb, const double *__restrict__ ub,
                                                             const double *__restrict__ v, int64_t e0, int64_t count, double *__restrict__ R)
{
    const int64_t e = e0 + (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (e >= e0 + count) return;
    const int64_t id = e / n;
    const int item = (int) (e - id * n);
    R[(size_t) id * ld + item] = lb[item] + (ub[item] - lb[item]) * v[e - e0];
}

/* ---- crossover ------------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(256) void esch_crossover_kernel(int n, int ld, int64_t np, const uint32_t *__restrict__ words,
                                                             const int32_t *__restrict__ slot, double *__restrict__ R)
{
    const int64_t id = blockIdx.x;
    const uint32_t *w = words + 3 * id;
    const int64_t p1 = (int64_t) (w[0] % (uint32_t) np), p2 = (int64_t) (w[1] % (uint32_t) np);
    const int cross = (int) (w[2] % (uint32_t) n);
    const double *a = R + (size_t) slot[p1] * ld, *b = R + (size_t) slot[p2] * ld;
    double *o = R + (size_t) slot[np + id] * ld;
    for (int j = threadIdx.x; j < n; j += 256) o[j] = j < cross ? a[j] : b[j];
}

/* ---- point mutations --------------------------------------------------------------------------------------------- */
/* a step starting at word p: [iurand(no)] [iurand(n)] then attempts at p+2, p+4, ... until one is accepted; returns the word
 * after the accepted attempt and the accepted value, or -1 if the generated segment ends before the step does */
__device__ __forceinline__ int64_t esch_step(const uint32_t *__restrict__ W, int64_t M, int64_t p, double &v01)
{
    int64_t q = p + 2;
    for (;;) {
        if (q + 1 >= M) return -1;
        if (esch_attempt(W[q], W[q + 1], v01)) return q + 2;
        q += 2;
    }
}

/* nx[p/2] = esch_step(p) for every even word position p of the segment (all step starts are even: steps have even length):
 * the functional graph of the chain, computed once in parallel — the walks below then cost one look-up per step instead of
 * the step's loads and tan() */
__global__ __launch_bounds__(256) void esch_mut_next_kernel(const uint32_t *__restrict__ W, int64_t M, int32_t *__restrict__ nx)
{
    const int64_t h = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (h >= M / 2 + 8) return;                               /* the table has 8 entries of slack past the segment: no step starts there */
    double v;
    nx[h] = 2 * h < M ? (int32_t) esch_step(W, M, 2 * h, v) : -1;
}

/* for block b and entry offset o (a chain entering the block at word b*BLOCK + o): exit offset into the next block
 * and the number of steps that START inside this block */
__global__ __launch_bounds__(ESCH_ENTRIES) void esch_mut_scan_kernel(const int32_t *__restrict__ nx, int64_t M, int32_t *__restrict__ exit_off,
                                                                     int32_t *__restrict__ nsteps)
{
    const int64_t b = blockIdx.x, end = (b + 1) * ESCH_BLOCK;
    int64_t p = b * ESCH_BLOCK + threadIdx.x;
    int steps = 0;
    bool cut = (threadIdx.x & 1) || p >= M;                    /* odd offsets are never entered */
    while (!cut && p < end) {
        const int64_t q = p < M ? nx[p >> 1] : -1;
        if (q < 0) { cut = true; break; }                     /* incomplete step: the segment is too short from here on */
        p = q; ++steps;
    }
    const int64_t eo = p - end;
    exit_off[b * ESCH_ENTRIES + threadIdx.x] = (cut || eo >= ESCH_ENTRIES) ? -1 : (int32_t) eo;
    nsteps[b * ESCH_ENTRIES + threadIdx.x] = steps;
}

/* chain the blocks: entry[b] = offset at which the chain enters block b (-1: not reached / not needed), first[b] = index of
 * the first step that starts in block b; out[0] = steps found (>= total when the segment was long enough), out[1] = word
 * position after step number `total` (filled by the mark kernel).  One workgroup: the per-block tables are brought into LDS
 * a tile of blocks at a time (coalesced), one thread then walks the tile — a look-up per block out of LDS instead of two
 * dependent global loads. */
#define ESCH_CHAIN_TILE 128
__global__ __launch_bounds__(256) void esch_mut_chain_kernel(int64_t nblocks, int64_t total, const int32_t *__restrict__ exit_off,
                                                             const int32_t *__restrict__ nsteps, int32_t *__restrict__ entry,
                                                             int64_t *__restrict__ first, int64_t *__restrict__ out)
{
    __shared__ int32_t s_exit[ESCH_CHAIN_TILE * ESCH_ENTRIES], s_steps[ESCH_CHAIN_TILE * ESCH_ENTRIES];
    __shared__ long long s_state[3];                           /* steps so far, entry offset, chain broken */
    const int tid = threadIdx.x;
    if (tid == 0) { s_state[0] = 0; s_state[1] = 0; s_state[2] = 0; }
    for (int64_t b0 = 0; b0 < nblocks; b0 += ESCH_CHAIN_TILE) {
        const int nb = (int) (nblocks - b0 < ESCH_CHAIN_TILE ? nblocks - b0 : ESCH_CHAIN_TILE);
        __syncthreads();
        for (int q = tid; q < nb * ESCH_ENTRIES; q += 256) { s_exit[q] = exit_off[b0 * ESCH_ENTRIES + q]; s_steps[q] = nsteps[b0 * ESCH_ENTRIES + q]; }
        __syncthreads();
        if (tid == 0) {
            long long steps = s_state[0];
            int o = (int) s_state[1];
            bool broken = s_state[2] != 0;
            for (int i = 0; i < nb; ++i) {
                const int64_t b = b0 + i;
                first[b] = steps;
                if (broken || steps >= total) { entry[b] = -1; continue; }
                entry[b] = o;
                const int32_t e = s_exit[i * ESCH_ENTRIES + o];
                steps += s_steps[i * ESCH_ENTRIES + o];
                if (e < 0) broken = true; else o = e;
            }
            s_state[0] = steps; s_state[1] = o; s_state[2] = broken;
        }
    }
    __syncthreads();
    if (tid == 0) out[0] = s_state[0];
}

/* replay block b from its true entry: step c (global index) mutates element (io, ip); last[io*n + ip] = max(c + 1).
 * The step with index total - 1 also records where the chain stands afterwards. */
__global__ __launch_bounds__(64) void esch_mut_mark_kernel(const uint32_t *__restrict__ W, const int32_t *__restrict__ nxt, int64_t M, int64_t total, int n,
                                                           int64_t no, const int32_t *__restrict__ entry, const int64_t *__restrict__ first,
                                                           int32_t *__restrict__ last, int64_t *__restrict__ out)
{
    if (threadIdx.x != 0) return;
    const int64_t b = blockIdx.x, end = (b + 1) * ESCH_BLOCK;
    if (entry[b] < 0) return;
    int64_t p = b * ESCH_BLOCK + entry[b], c = first[b];
    while (p < end && c < total) {
        const int64_t nx = p < M ? nxt[p >> 1] : -1;
        if (nx < 0) break;
        const int64_t io = (int64_t) (W[p] % (uint32_t) no);
        const int ip = (int) (W[p + 1] % (uint32_t) n);
        atomicMax(last + io * n + ip, (int32_t) (c + 1));
        p = nx;
        if (c == total - 1) out[1] = p;
        ++c;
    }
}

__global__ __launch_bounds__(64) void esch_mut_apply_kernel(const uint32_t *__restrict__ W, const int32_t *__restrict__ nxt, int64_t M, int64_t total, int n,
                                                            int ld, int64_t np, int64_t no, const int32_t *__restrict__ entry,
                                                            const int64_t *__restrict__ first,
                                                            const int32_t *__restrict__ last, const double *__restrict__ lb,
                                                            const double *__restrict__ ub, const int32_t *__restrict__ slot,
                                                            double *__restrict__ R)
{
    if (threadIdx.x != 0) return;
    const int64_t b = blockIdx.x, end = (b + 1) * ESCH_BLOCK;
    if (entry[b] < 0) return;
    int64_t p = b * ESCH_BLOCK + entry[b], c = first[b];
    while (p < end && c < total) {
        const int64_t nx = p < M ? nxt[p >> 1] : -1;
        if (nx < 0) break;
        const int64_t io = (int64_t) (W[p] % (uint32_t) no);
        const int ip = (int) (W[p + 1] % (uint32_t) n);
        if (last[io * n + ip] == (int32_t) (c + 1)) {        /* the serial loop's last write to this element */
            double v = 0;
            (void) esch_attempt(W[nx - 2], W[nx - 1], v);    /* the step's accepted attempt is the one that ends it */
            R[(size_t) slot[np + io] * ld + ip] = lb[ip] + (ub[ip] - lb[ip]) * v;
        }
        p = nx;
        ++c;
    }
}

/* ---- evaluation support / selection -------------------------------------------------------------------------------- */
/* G[i - i0] := row of individual i (contiguous copy for nla_k_eval or the host callback) */
__global__ __launch_bounds__(256) void esch_gather_rows_kernel(int n, int ld, const int32_t *__restrict__ slot, int64_t i0,
                                                               const double *__restrict__ R, double *__restrict__ G)
{
    const double *r = R + (size_t) slot[i0 + blockIdx.x] * ld;
    double *g = G + (size_t) blockIdx.x * ld;
    for (int j = threadIdx.x; j < n; j += 256) g[j] = r[j];
}

/* order-preserving key of a double (a < b  <=>  key(a) < key(b), -0 < +0 aside — fitness compares a < b / a > b only) */
__global__ __launch_bounds__(256) void esch_keys_kernel(int64_t count, const double *__restrict__ fit, unsigned long long *__restrict__ keys,
                                                        int32_t *__restrict__ idx)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double f = fit[i];
    if (f == 0.0) f = 0.0;                                    /* -0 and +0 compare equal in the reference's comparator */
    unsigned long long u = (unsigned long long) __double_as_longlong(f);
    u = (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
    keys[i] = u; idx[i] = (int32_t) i;
}
__global__ __launch_bounds__(256) void esch_permute_kernel(int64_t count, const int32_t *__restrict__ order, const int32_t *__restrict__ slot_in,
                                                           const double *__restrict__ fit_in, int32_t *__restrict__ slot_out,
                                                           double *__restrict__ fit_out)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    slot_out[i] = slot_in[order[i]]; fit_out[i] = fit_in[order[i]];
}

/* ---- launchers ------------------------------------------------------------------------------------------------------ */
extern "C" int nla_k_esch_cauchy(const uint32_t *words, int64_t nattempts, int64_t attempt_base, int32_t *counts, int64_t *vtotal,
                                 int64_t vbase, int64_t vcap, double *v, int64_t *vatt, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    if (nattempts <= 0) return 0;
    const int nwg = (int) ((nattempts + ESCH_PER_WG - 1) / ESCH_PER_WG);
    hipLaunchKernelGGL(esch_cauchy_count_kernel, dim3(nwg), dim3(256), 0, st, words, nattempts, counts);
    hipLaunchKernelGGL(esch_scan_kernel, dim3(1), dim3(1024), 0, st, counts, nwg, vtotal);
    hipLaunchKernelGGL(esch_cauchy_write_kernel, dim3(nwg), dim3(256), 0, st, words, nattempts, attempt_base, counts, vbase, vcap, v, vatt);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_esch_fill_rows(int n, int ld, const double *lb, const double *ub, const double *v, int64_t e0, int64_t count, double *R,
                                    void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(esch_fill_rows_kernel, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, (hipStream_t) stream, n, ld, lb, ub, v, e0, count, R);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_esch_crossover(int n, int ld, int64_t np, int64_t no, const uint32_t *words, const int32_t *slot, double *R, void *stream)
{
    if (no <= 0) return 0;
    hipLaunchKernelGGL(esch_crossover_kernel, dim3((unsigned) no), dim3(256), 0, (hipStream_t) stream, n, ld, np, words, slot, R);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t nla_esch_mut_scratch_bytes(int64_t M)
{
    const size_t nb = (size_t) ((M + ESCH_BLOCK - 1) / ESCH_BLOCK) + 1;
    return nb * ESCH_ENTRIES * 4 * 2 + nb * 4 + nb * 8 + ((size_t) M / 2 + 8) * 4 + 512;
}

/* the (no n)/10 point mutations of one generation from the M stream words W; last: no*n ints of scratch (zeroed here);
 * h-visible result through `out` (device, 2 x int64): out[0] = steps the segment holds (< total: M was too short, nothing
 * was applied), out[1] = words consumed by the `total` steps */
extern "C" int nla_k_esch_mutate(const uint32_t *W, int64_t M, int64_t total, int n, int ld, int64_t np, int64_t no, const double *lb,
                                 const double *ub, const int32_t *slot, double *R, int32_t *last, void *scratch, int64_t *out, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    const int64_t nb = (M + ESCH_BLOCK - 1) / ESCH_BLOCK;
    if (nb <= 0 || total <= 0) return (int) hipErrorInvalidValue;
    char *p = (char *) scratch;
    int32_t *exit_off = (int32_t *) p; p += (size_t) nb * ESCH_ENTRIES * 4;
    int32_t *nsteps = (int32_t *) p; p += (size_t) nb * ESCH_ENTRIES * 4;
    int32_t *entry = (int32_t *) p; p += (((size_t) nb * 4 + 7) & ~(size_t) 7);
    int64_t *first = (int64_t *) p; p += (size_t) nb * 8;
    int32_t *nx = (int32_t *) p;
    if (M >= (1LL << 31)) return (int) hipErrorInvalidValue;
    (void) hipMemsetAsync(last, 0, sizeof(int32_t) * (size_t) no * (size_t) n, st);
    (void) hipMemsetAsync(out, 0, 2 * sizeof(int64_t), st);
    hipLaunchKernelGGL(esch_mut_next_kernel, dim3((unsigned) ((M / 2 + 8 + 255) / 256)), dim3(256), 0, st, W, M, nx);
    hipLaunchKernelGGL(esch_mut_scan_kernel, dim3((unsigned) nb), dim3(ESCH_ENTRIES), 0, st, nx, M, exit_off, nsteps);
    hipLaunchKernelGGL(esch_mut_chain_kernel, dim3(1), dim3(256), 0, st, nb, total, exit_off, nsteps, entry, first, out);
    hipLaunchKernelGGL(esch_mut_mark_kernel, dim3((unsigned) nb), dim3(64), 0, st, W, nx, M, total, n, no, entry, first, last, out);
    hipLaunchKernelGGL(esch_mut_apply_kernel, dim3((unsigned) nb), dim3(64), 0, st, W, nx, M, total, n, ld, np, no, entry, first, last, lb, ub, slot, R);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_esch_gather_rows(int n, int ld, const int32_t *slot, int64_t i0, int64_t count, const double *R, double *G, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(esch_gather_rows_kernel, dim3((unsigned) count), dim3(256), 0, (hipStream_t) stream, n, ld, slot, i0, R, G);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t nla_esch_sort_scratch_bytes(int64_t count)
{
    size_t tmp = 0;
    (void) hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const unsigned long long *) nullptr, (unsigned long long *) nullptr,
                                              (const int32_t *) nullptr, (int32_t *) nullptr, (int) count);
    /* keys in | keys out | idx in | order | radix temp */
    return (size_t) count * (8 + 8 + 4 + 4) + tmp + 1024;
}

/* selection: (slot, fit) of the count individuals reordered by fitness, stable; fit_out / slot_out may not alias the inputs */
extern "C" int nla_k_esch_select(int64_t count, const int32_t *slot_in, const double *fit_in, int32_t *slot_out, double *fit_out,
                                 void *scratch, size_t scratch_bytes, void *stream)
{
    hipStream_t st = (hipStream_t) stream;
    if (count <= 0) return 0;
    char *p = (char *) scratch;
    unsigned long long *kin = (unsigned long long *) p; p += (size_t) count * 8;
    unsigned long long *kout = (unsigned long long *) p; p += (size_t) count * 8;
    int32_t *iin = (int32_t *) p; p += (size_t) count * 4;
    int32_t *order = (int32_t *) p; p += (size_t) count * 4;
    p = (char *) (((uintptr_t) p + 255) & ~(uintptr_t) 255);
    size_t tmp = scratch_bytes - (size_t) (p - (char *) scratch);
    const unsigned g = (unsigned) ((count + 255) / 256);
    hipLaunchKernelGGL(esch_keys_kernel, dim3(g), dim3(256), 0, st, count, fit_in, kin, iin);
    if (hipcub::DeviceRadixSort::SortPairs(p, tmp, kin, kout, iin, order, (int) count, 0, 64, st) != hipSuccess) return (int) hipErrorUnknown;
    hipLaunchKernelGGL(esch_permute_kernel, dim3(g), dim3(256), 0, st, count, order, slot_in, fit_in, slot_out, fit_out);
    NLA_LAUNCH_CHECK();
    return 0;
}
