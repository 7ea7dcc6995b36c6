/* mlsl_kernels.hip — Multi-Level Single-Linkage (src/algs/mlsl/mlsl.c) device work besides the
 * batched local searches (lbfgs_kernels.hip) and the sample generation + evaluation (the
 * row-from-stream kernel of crs_kernels.hip):
 *
 *   dist2        |a_i - b_j|^2 for all pairs of two point sets — the reference's distance2
 *                (mlsl.c:118-127) inside find_closest_pt / find_closest_lm / pts_update_newpt /
 *                pts_update_newlm (:131-194), N_new x |pts| x n flops per iteration, the sampling
 *                phase's hot spot.  LDS- and register-tiled (64x64 pairs per workgroup, 4x4 per thread);
 *                every pair is summed by one thread over k in ascending order without FMA, i.e. in the
 *                reference's order: distances are bit-identical to the CPU's.
 *   masked mins  closest_pt_d / closest_lm_d updates: min over the partner set restricted to
 *                partners with strictly smaller f (mlsl.c:133,147,164,184).
 */
#include "dev_common.h"
#include "../../../include/nlopt_amd.h"

/* A workgroup owns 64 x 64 pairs, a thread 4 x 4 of them (rows ty + 16 r, columns tx + 16 c): per coordinate 8 LDS reads feed 16
 * pairs.  (Rounds 1-3 ran one pair per thread on 16 x 16 tiles — two LDS reads per pair and coordinate, bound by LDS bandwidth:
 * 6.4 ms for 1000 x 3000 pairs at n = 4096; measured on the MI355X in round 4 (gpurun_out/r04_first): the sampling phase of
 * config 4 went 9.4 -> 7.6 ms per iteration, the direct bit-for-bit test and the MLSL files green, and the old kernel was deleted.)
 * Every pair is summed by ONE thread over k ascending, subtract / multiply / add unfused: bit-identical to distance2.
 * (Round 5 tried the rows of A through the SCALAR unit — a wavefront owning 8 rows, their coordinates scalar loads and scalar operands
 * of the subtraction, 2 LDS reads per 16 pairs instead of 8: slower, 2.54 against 2.17 ms for 1000 x 4000 pairs at n = 4096 and 1.54
 * against 1.21 ms for 1000 x 1800 (profiles/r05_mlsl_ahead_ab.txt) — scalar loads and LDS reads share one counter, every wait for
 * either is a wait for both.  Deleted.) */
#define DKR 32         /* coordinates per LDS tile */
/* T = pairs per thread and direction: the workgroup's tile is 16 T x 16 T pairs (T = 4: 64 x 64, the throughput shape; T = 2: 32 x 32 — a
 * quarter of the work per workgroup for the calls that have few pairs: a workgroup walks all n coordinates of its tile alone, 1.1 ms at
 * n = 4096 for a 64 x 64 tile whatever the number of tiles — the 1000 x 300 pairs of new samples against new minima, 80 tiles, took
 * those 1.1 ms (profiles/r05_mlsl_timeline_ahead.txt) */
template <int T>
__global__ __launch_bounds__(256) void mlsl_dist2_kernel(int n, int ld, const double *__restrict__ A, int na,
                                                         const double *__restrict__ B, int nb, double *__restrict__ D)
{
    constexpr int DR = 16 * T;
    __shared__ double sa[DR][DKR + 1], sb[DR][DKR + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * DR, j0 = blockIdx.x * DR;
    double d[T][T];
#pragma unroll
    for (int r = 0; r < T; ++r)
#pragma unroll
        for (int c = 0; c < T; ++c) d[r][c] = 0.;
    /* the NEXT coordinate tile travels from global memory into registers while this one is being summed out of LDS (a thread stages
     * 2 T + 2 T values per tile: element q * 256 + tid of the DR x 32 tile, row = e / 32 — a wavefront reads two 256-byte row pieces) */
    constexpr int PER = DR * DKR / 256;
    double pa[PER], pb[PER];
    auto fetch = [&](int k0) {
        const int kc = n - k0 < DKR ? n - k0 : DKR;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + (int) threadIdx.x, r = e / DKR, k = e - r * DKR;
            pa[q] = (i0 + r < na && k < kc) ? A[(size_t) (i0 + r) * ld + k0 + k] : 0.;
            pb[q] = (j0 + r < nb && k < kc) ? B[(size_t) (j0 + r) * ld + k0 + k] : 0.;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < n; k0 += DKR) {
        const int kc = n - k0 < DKR ? n - k0 : DKR;
        __syncthreads();                       /* the previous tile has been read by everyone */
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = q * 256 + (int) threadIdx.x, r = e / DKR, k = e - r * DKR;
            sa[r][k] = pa[q]; sb[r][k] = pb[q];
        }
        __syncthreads();
        if (k0 + DKR < n) fetch(k0 + DKR);
        for (int k = 0; k < kc; ++k) {
            double a[T], b[T];
#pragma unroll
            for (int r = 0; r < T; ++r) { a[r] = sa[ty + 16 * r][k]; b[r] = sb[tx + 16 * r][k]; }
#pragma unroll
            for (int r = 0; r < T; ++r)
#pragma unroll
                for (int c = 0; c < T; ++c) { const double dx = a[r] - b[c]; d[r][c] += dx * dx; }
        }
    }
#pragma unroll
    for (int r = 0; r < T; ++r)
#pragma unroll
        for (int c = 0; c < T; ++c) {
            const int i = i0 + ty + 16 * r, j = j0 + tx + 16 * c;
            if (i < na && j < nb) D[(size_t) i * nb + j] = d[r][c];
        }
}

/* out[i] = min(init[i], min_j { D[i][j] : FB[j] < FA[i] })        (one wavefront per row i) */
__global__ __launch_bounds__(256) void mlsl_rowmin_kernel(const double *__restrict__ D, int ldd, int na, int nb, const double *__restrict__ FA,
                                                          const double *__restrict__ FB, const double *__restrict__ init,
                                                          double *__restrict__ out)
{
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= na) return;
    const double fi = FA[i];
    double m = HUGE_VAL;
    for (int j = lane; j < nb; j += 64) if (FB[j] < fi) { const double d = D[(size_t) i * ldd + j]; m = d < m ? d : m; }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { const double o = __shfl_xor(m, s, 64); m = o < m ? o : m; }
    if (lane == 0) { const double b = init ? init[i] : HUGE_VAL; out[i] = m < b ? m : b; }
}

/* inout[j] = min(inout[j], min_i { D[i][j] : FA[i] < FB[j] }) for j with skip[j] == 0 (one thread per column j) */
__global__ __launch_bounds__(256) void mlsl_colmin_kernel(const double *__restrict__ D, int ldd, int na, int nb, const double *__restrict__ FA,
                                                          const double *__restrict__ FB, const int32_t *__restrict__ skip,
                                                          double *__restrict__ inout)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nb || (skip && skip[j])) return;
    const double fj = FB[j];
    double m = inout[j];
    for (int i = 0; i < na; ++i) if (FA[i] < fj) { const double d = D[(size_t) i * ldd + j]; m = d < m ? d : m; }
    inout[j] = m;
}

/* Sobol points by index (nlopt_sobol_next, sobolseq.c:236-242; see ../sobol.c for why a point is a pure function
 * of its index): row r of P := lb + (ub - lb) * (x_k / 2^32), k = index_first + r, x_k = XOR of V[c] over the set
 * bits c of gray(k).  One thread per (point, coordinate); V is 32 x n u32, coalesced over the coordinate. */
__global__ __launch_bounds__(256) void mlsl_sobol_rows_kernel(int n, int ld, const double *__restrict__ lb, const double *__restrict__ ub,
                                                              const uint32_t *__restrict__ V, uint32_t index_first, int count,
                                                              double *__restrict__ P)
{
    const int i = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (i >= n || r >= count) return;
    const uint32_t k = index_first + (uint32_t) r;
    uint32_t g = k ^ (k >> 1), acc = 0;
    while (g) { const int c = __builtin_ctz(g); acc ^= V[(size_t) c * n + i]; g &= g - 1; }
    const double u = (double) acc / 4294967296.0;
    P[(size_t) r * ld + i] = lb[i] + (ub[i] - lb[i]) * u;
}

extern "C" int nla_k_mlsl_sobol_rows(int n, int ld, const double *lb, const double *ub, const uint32_t *V, uint32_t index_first,
                                     int count, double *P, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(mlsl_sobol_rows_kernel, dim3((n + 255) / 256, count), dim3(256), 0, (hipStream_t) stream, n, ld, lb, ub, V,
                       index_first, count, P);
    NLA_LAUNCH_CHECK();
    return 0;
}

#ifndef NLA_DIST2_SMALL_BELOW
#define NLA_DIST2_SMALL_BELOW 400
#endif
extern "C" int nla_k_mlsl_dist2(int n, int ld, const double *A, int na, const double *B, int nb, double *D, void *stream)
{
    if (na <= 0 || nb <= 0) return 0;
    /* few 64 x 64 tiles: the small tile (measured alone, MI355X, n = 4096, profiles/r05_mlsl_ahead_ab.txt: 1000 x 300 pairs 0.38 against
     * 0.71 ms, 305 x 4000 0.91 against 1.09 — but 1000 x 1800, 464 tiles, 1.30 against 1.21 and 305 x 8000 1.63 against 1.58) */
    if ((long) ((na + 63) / 64) * (long) ((nb + 63) / 64) <= NLA_DIST2_SMALL_BELOW)
        hipLaunchKernelGGL(mlsl_dist2_kernel<2>, dim3((unsigned) ((nb + 31) / 32), (unsigned) ((na + 31) / 32)), dim3(256), 0, (hipStream_t) stream, n, ld, A, na, B, nb, D);
    else
        hipLaunchKernelGGL(mlsl_dist2_kernel<4>, dim3((unsigned) ((nb + 63) / 64), (unsigned) ((na + 63) / 64)), dim3(256), 0, (hipStream_t) stream, n, ld, A, na, B, nb, D);
    NLA_LAUNCH_CHECK();
    return 0;
}
extern "C" int nla_k_mlsl_rowmin(const double *D, int ldd, int na, int nb, const double *FA, const double *FB, const double *init, double *out, void *stream)
{
    if (na <= 0) return 0;
    hipLaunchKernelGGL(mlsl_rowmin_kernel, dim3((unsigned) ((na + 3) / 4)), dim3(256), 0, (hipStream_t) stream, D, ldd, na, nb, FA, FB, init, out);
    NLA_LAUNCH_CHECK();
    return 0;
}
extern "C" int nla_k_mlsl_colmin(const double *D, int ldd, int na, int nb, const double *FA, const double *FB, const int32_t *skip, double *inout, void *stream)
{
    if (nb <= 0 || na <= 0) return 0;
    hipLaunchKernelGGL(mlsl_colmin_kernel, dim3((unsigned) ((nb + 255) / 256)), dim3(256), 0, (hipStream_t) stream, D, ldd, na, nb, FA, FB, skip, inout);
    NLA_LAUNCH_CHECK();
    return 0;
}

/* out[a * nc + b] = D[rows[a] * ldd + cols[b]]: the distances between the minimisers of a batch's searches (rows of the batch's distance
 * matrix) and the batch's own start points (columns) — all the commit walk on the host needs of that matrix (mlsl_driver.c) */
__global__ __launch_bounds__(256) void mlsl_gather_pairs_kernel(const double *__restrict__ D, int ldd, const int64_t *__restrict__ rows, int nr,
                                                                 const int64_t *__restrict__ cols, int nc, double *__restrict__ out)
{
    const int b = blockIdx.x * 256 + threadIdx.x, a = blockIdx.y;
    if (b < nc && a < nr) out[(size_t) a * nc + b] = D[(size_t) rows[a] * (size_t) ldd + (size_t) cols[b]];
}
/* the same transposed, out[b * nr + a]: the commit walk asks, for start point b, about the minimisers a < b one after the other — with the
 * pairs in this order its inner loop reads consecutive memory instead of one cache line per question (0.45 -> 0.1 ms per iteration at
 * config 4's 300-search batches) */
__global__ __launch_bounds__(256) void mlsl_gather_pairs_t_kernel(const double *__restrict__ D, int ldd, const int64_t *__restrict__ rows, int nr,
                                                                   const int64_t *__restrict__ cols, int nc, double *__restrict__ out)
{
    const int b = blockIdx.x * 256 + threadIdx.x, a = blockIdx.y;
    if (b < nc && a < nr) out[(size_t) b * nr + a] = D[(size_t) rows[a] * (size_t) ldd + (size_t) cols[b]];
}
extern "C" int nla_k_mlsl_gather_pairs_t(const double *D, int ldd, const int64_t *rows, int nr, const int64_t *cols, int nc, double *out, void *stream)
{
    if (nr <= 0 || nc <= 0) return 0;
    hipLaunchKernelGGL(mlsl_gather_pairs_t_kernel, dim3((unsigned) ((nc + 255) / 256), (unsigned) nr), dim3(256), 0, (hipStream_t) stream, D, ldd, rows, nr, cols, nc, out);
    NLA_LAUNCH_CHECK();
    return 0;
}
extern "C" int nla_k_mlsl_gather_pairs(const double *D, int ldd, const int64_t *rows, int nr, const int64_t *cols, int nc, double *out, void *stream)
{
    if (nr <= 0 || nc <= 0) return 0;
    hipLaunchKernelGGL(mlsl_gather_pairs_kernel, dim3((unsigned) ((nc + 255) / 256), (unsigned) nr), dim3(256), 0, (hipStream_t) stream, D, ldd, rows, nr, cols, nc, out);
    NLA_LAUNCH_CHECK();
    return 0;
}

/* rows by index: dst row c := src row idx[c] (the start points of a batch of local searches, mlsl.c:399-404; the accepted minima
 * joining the set of local minima, mlsl.c:410-414) — one launch instead of one copy operation per row */
__global__ __launch_bounds__(256) void mlsl_gather_rows_kernel(int n, int ld, const double *__restrict__ src, const int64_t *__restrict__ idx,
                                                                double *__restrict__ dst)
{
    const double *r = src + (size_t) idx[blockIdx.x] * (size_t) ld;
    double *g = dst + (size_t) blockIdx.x * (size_t) ld;
    for (int j = threadIdx.x; j < n; j += 256) g[j] = r[j];
}

/* the bound test of is_potential_minimizer (mlsl.c:211-218) for `count` points at once: flags[c] = 1 if some coordinate of
 * row idx[c] lies within thr (= dbound R) of a bound whose box side is wider than thr.  Same comparisons as the reference,
 * on the same doubles. */
__global__ __launch_bounds__(256) void mlsl_near_bound_kernel(int n, int ld, const double *__restrict__ P, const int64_t *__restrict__ idx,
                                                               const double *__restrict__ lb, const double *__restrict__ ub, double thr,
                                                               int32_t *__restrict__ flags)
{
    __shared__ int s_hit;
    const double *x = P + (size_t) idx[blockIdx.x] * (size_t) ld;
    if (threadIdx.x == 0) s_hit = 0;
    __syncthreads();
    int hit = 0;
    for (int j = threadIdx.x; j < n; j += 256)
        if ((x[j] - lb[j] <= thr || ub[j] - x[j] <= thr) && ub[j] - lb[j] > thr) hit = 1;
    if (hit) s_hit = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[blockIdx.x] = s_hit;
}

extern "C" int nla_k_mlsl_gather_rows(int n, int ld, const double *src, const int64_t *idx, int count, double *dst, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(mlsl_gather_rows_kernel, dim3((unsigned) count), dim3(256), 0, (hipStream_t) stream, n, ld, src, idx, dst);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_mlsl_near_bound(int n, int ld, const double *P, const int64_t *idx, int count, const double *lb, const double *ub,
                                     double thr, int32_t *flags, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(mlsl_near_bound_kernel, dim3((unsigned) count), dim3(256), 0, (hipStream_t) stream, n, ld, P, idx, lb, ub, thr, flags);
    NLA_LAUNCH_CHECK();
    return 0;
}


/* F[i] := -F[i]: the reference's maximisation wrapper (f_max, optimize.c:970-980) applied to a batch of device-evaluated samples */
__global__ __launch_bounds__(256) void mlsl_negate_kernel(double *__restrict__ F, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) F[i] = -F[i];
}
extern "C" int nla_k_mlsl_negate(double *F, int count, void *stream)
{
    if (count <= 0) return 0;
    hipLaunchKernelGGL(mlsl_negate_kernel, dim3((unsigned) ((count + 255) / 256)), dim3(256), 0, (hipStream_t) stream, F, count);
    NLA_LAUNCH_CHECK();
    return 0;
}
