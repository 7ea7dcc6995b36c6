/* crs_kernels.hip — Controlled Random Search (CRS2_LM) on gfx950.
 *
 * Reference loops replaced: K1/K2 population init + evaluation
 * (src/algs/crs/crs.c:211-226), K3 Vitter method-A selection (crs.c:89-109), K4 the
 * centroid/reflection gather-sum (crs.c:101-120), K5 local mutation (crs.c:139-146), K6 the data
 * the in-order commit needs, and the row write-back (crs.c:153).
 *
 * Data layout in HBM: X is N x ld fp64, row-major, ld = n rounded up to even so every row starts
 * 16-byte aligned (the reference interleaves f as column 0 of an (n+1)-wide row, crs.c:43; f lives
 * in a separate host array here).  Trial points TX/TM are K x ld.  The MT word stream for a run of
 * 2n-word blocks is a flat uint32 array; the pre-digested selection of block b is pos[b*n ..].
 *
 * Roofline: crs_gather_kernel is the hot kernel — HBM-bound, 8*n*(n+1) algorithmic bytes per
 * trial (n random rows + the best row).  Parallelism: the n coordinates of one trial (one lane
 * per coordinate pair, rows accumulated in the reference's order so x is bit-identical) times K
 * speculative trials per launch; memory-level parallelism comes from U independent row loads in
 * flight per lane.  Everything else here is O(n) or O(N) per trial and off the roofline.
 */
#include "dev_common.h"
#include "../../../include/nlopt_amd.h"
#include <limits.h>

/* ------------------------------------------------------------------------------------------------
 * K1+K2: rows of the initial population from the MT stream, evaluated in the same pass.
 * One wavefront per row; lane l writes coordinates l, l+64, ... (coalesced 512 B stores).
 * ---------------------------------------------------------------------------------------------- */
template <int OBJ>
__global__ __launch_bounds__(256) void crs_init_rows_kernel(int n, int ld, const double *__restrict__ lb,
                                                             const double *__restrict__ ub,
                                                             const uint32_t *__restrict__ words, int64_t row_first,
                                                             int64_t nrows, double *__restrict__ X, double *__restrict__ F)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const uint32_t *w = words + (size_t) r * 2 * (size_t) n;
    double *xr = X + (size_t) (row_first + r) * (size_t) ld;
    auto gen = [&](int i) {                      /* k[1+j] = nlopt_urand(lb[j], ub[j]), crs.c:216-218 */
        const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
        return nla_urand_from(lb[i], ub[i], ww.x, ww.y);
    };
    for (int i = lane; i < n; i += 64) xr[i] = gen(i);
    if (OBJ >= 0) {
        double f = nla_wave_objective<(OBJ >= 0 ? OBJ : 0)>(n, gen);
        if (lane == 0) F[row_first + r] = f;
    }
}

/* generic batched evaluation: one wavefront per candidate */
template <int OBJ>
__global__ __launch_bounds__(256) void eval_kernel(int n, int ld, const double *__restrict__ P, int64_t count,
                                                    double *__restrict__ F)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= count) return;
    const double *x = P + (size_t) c * (size_t) ld;
    double f = nla_wave_objective<OBJ>(n, [&](int i) { return x[i]; });
    if (lane == 0) F[c] = f;
}

/* ------------------------------------------------------------------------------------------------
 * K3: Vitter method A, one lane per 2n-word stream block.
 * The reference's chain (crs.c:93-100) is, per visited row: q = (q*Nfree)/Nleft while q > v.  It
 * is inherently serial in fp64 (each q is a rounded mul + a rounded div of the previous one), so
 * the parallelism is across *future* blocks: every trial consumes exactly 2n words
 * [jn | v_0..v_{n-2} | last] regardless of history, hence block b's selection
 * is a pure function of its words and (n, N).  The skip loop and the pick step are fused into one
 * loop with exactly one IEEE division per visited row, so the lanes of a wavefront stay
 * converged (each runs ~N(n-1)/n iterations).  Positions are in "reduced" space (the best row
 * removed); the best's index is applied when the block is used.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void crs_vitter_kernel(int n, int64_t N, const uint32_t *__restrict__ words, int nblocks,
                                                         int32_t *__restrict__ jn_out, int32_t *__restrict__ pos,
                                                         int32_t *__restrict__ last)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t *w = words + (size_t) b * 2 * (size_t) n;
    int32_t *out = pos + (size_t) b * (size_t) n;
    jn_out[b] = (int32_t) (w[0] % (uint32_t) n);                 /* nlopt_iurand(n), crs.c:72 */
    int Nleft = (int) (N - 1), nleft = n, Nfree = Nleft - nleft; /* crs.c:90-91 */
    int r = 0, t = 0;
    if (n > 1) {
        double q = ((double) Nfree) / Nleft;
        double v = nla_res53(w[1], w[2]);
        double vn = (n > 2) ? nla_res53(w[3], w[4]) : 0.0;       /* prefetched uniform of the next pick */
        for (;;) {
            const bool take = !(q > v);
            if (take) {
                out[t] = r;
                ++t; --nleft;
                if (nleft == 1) { ++r; --Nleft; break; }
                v = vn;
                if (t + 1 < n - 1) vn = nla_res53(w[1 + 2 * (t + 1)], w[2 + 2 * (t + 1)]);
            }
            const double num = take ? (double) Nfree : q * (double) (Nfree - 1);
            Nfree -= take ? 0 : 1;
            --Nleft; ++r;
            q = num / (double) Nleft;
        }
    }
    out[n - 1] = r;
    last[b] = (int32_t) (w[2 * n - 1] % (uint32_t) Nleft);       /* nlopt_iurand(Nleft), crs.c:109 */
}

/* ------------------------------------------------------------------------------------------------
 * K4: the gather-sum.  grid = K slots x wps wavefronts; wavefront `chunk` of a slot owns
 * coordinates [chunk*64*VEC, (chunk+1)*64*VEC).  Row indices are wave-uniform (scalar loads);
 * U row segments are requested before the first is accumulated.
 * ---------------------------------------------------------------------------------------------- */
template <int VEC> struct VecT;
template <> struct VecT<1> { typedef double T; };
template <> struct VecT<2> { typedef double2 T; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T ldv(const double *p);
template <> __device__ __forceinline__ double ldv<1>(const double *p) { return *p; }
template <> __device__ __forceinline__ double2 ldv<2>(const double *p) { return *reinterpret_cast<const double2 *>(p); }

__device__ __forceinline__ void acc_row(double &a, double v, double m) { a = a + v * m; }
__device__ __forceinline__ void acc_row(double2 &a, double2 v, double m) { a.x = a.x + v.x * m; a.y = a.y + v.y * m; }

template <int VEC, int U>
__global__ __launch_bounds__(256) void crs_gather_kernel(int n, int ld, const double *__restrict__ X, int64_t i0,
                                                          const int32_t *__restrict__ jn_arr, const int32_t *__restrict__ pos,
                                                          const int32_t *__restrict__ last, int K, int wps,
                                                          const double *__restrict__ lb, const double *__restrict__ ub,
                                                          double *__restrict__ TX)
{
    typedef typename VecT<VEC>::T V;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int slot = wave / wps;
    if (slot >= K) return;
    const int chunk = wave - slot * wps;
    const int col = (chunk * 64 + lane) * VEC;
    const bool active = col < n;
    const size_t colc = active ? (size_t) col : 0;
    const int32_t *p = pos + (size_t) slot * (size_t) n;
    const int jn = jn_arr[slot];
    const double hneg = -(0.5 * n);      /* x -= xi*(0.5*n)  ==  x += xi*(-(0.5*n)), exactly */
    const double *Xc = X + colc;

    V acc = ldv<VEC>(Xc + (size_t) i0 * (size_t) ld);      /* x := best (crs.c:69) */

    const int nmain = n - 1;             /* picks 0..n-2 come from pos[]; pick n-1 is the jump */
    int t0 = 0;
    for (; t0 + U <= nmain; t0 += U) {
        V v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = p[t0 + u];
            const int64_t a = r + (r >= i0 ? 1 : 0);        /* i += i == i0 skipping, crs.c:92,97,106 */
            v[u] = ldv<VEC>(Xc + (size_t) a * (size_t) ld);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc_row(acc, v[u], (t0 + u == jn) ? hneg : 1.0);
    }
    for (; t0 < nmain; ++t0) {
        const int64_t r = p[t0];
        const int64_t a = r + (r >= i0 ? 1 : 0);
        acc_row(acc, ldv<VEC>(Xc + (size_t) a * (size_t) ld), (t0 == jn) ? hneg : 1.0);
    }
    {   /* last pick: i += iurand(Nleft); i += i == i0  (crs.c:109) */
        const int64_t rb = p[n - 1];
        int64_t a = rb + (rb >= i0 ? 1 : 0) + (int64_t) last[slot];
        a += (a == i0) ? 1 : 0;
        acc_row(acc, ldv<VEC>(Xc + (size_t) a * (size_t) ld), (n - 1 == jn) ? hneg : 1.0);
    }
    if (active) {
        const double s = 2.0 / n;        /* x[k] *= 2.0 / n, then clamp (crs.c:116-120) */
        double *o = TX + (size_t) slot * (size_t) ld + col;
        if constexpr (VEC == 1) {
            double a0 = *reinterpret_cast<double *>(&acc);
            o[0] = nla_clamp_box(a0 * s, lb[col], ub[col]);
        } else {
            double2 a2 = *reinterpret_cast<double2 *>(&acc);
            double2 r2;
            r2.x = nla_clamp_box(a2.x * s, lb[col], ub[col]);
            r2.y = nla_clamp_box(a2.y * s, lb[col + 1], ub[col + 1]);
            *reinterpret_cast<double2 *>(o) = r2;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * post kernel: 3K single-wavefront tasks — [0,K) evaluate the trial, [K,2K) local mutation of the
 * trial (as if it will be rejected) + its evaluation, [2K,3K) hazard rank of the slot.
 * ---------------------------------------------------------------------------------------------- */
template <int OBJ>
__global__ __launch_bounds__(64) void crs_post_kernel(int n, int ld, const double *__restrict__ X, int64_t i0,
                                                       const double *__restrict__ TX, double *__restrict__ TM,
                                                       const uint32_t *__restrict__ words_next, int K,
                                                       const int64_t *__restrict__ W, int nW,
                                                       const int32_t *__restrict__ pos, const int32_t *__restrict__ last,
                                                       const double *__restrict__ lb, const double *__restrict__ ub,
                                                       double *__restrict__ fT, double *__restrict__ fM,
                                                       int32_t *__restrict__ minhz)
{
    const int lane = threadIdx.x;
    const int task = blockIdx.x / K, s = blockIdx.x - task * K;
    if (task == 0) {
        if (OBJ < 0) return;
        const double *x = TX + (size_t) s * (size_t) ld;
        double f = nla_wave_objective<(OBJ >= 0 ? OBJ : 0)>(n, [&](int i) { return x[i]; });
        if (lane == 0) fT[s] = f;
    } else if (task == 1) {
        if (OBJ < 0) return;
        const double *x = TX + (size_t) s * (size_t) ld;
        const double *xb = X + (size_t) i0 * (size_t) ld;
        const uint32_t *w = words_next + (size_t) s * 2 * (size_t) n;
        double *m = TM + (size_t) s * (size_t) ld;
        auto mut = [&](int i) {            /* p_i = best_i (1+w) - w p_i, clamp (crs.c:140-145) */
            const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
            const double wv = nla_urand_from(0., 1., ww.x, ww.y);
            return nla_clamp_box(xb[i] * (1 + wv) - wv * x[i], lb[i], ub[i]);
        };
        for (int i = lane; i < n; i += 64) m[i] = mut(i);
        double f = nla_wave_objective<(OBJ >= 0 ? OBJ : 0)>(n, mut);
        if (lane == 0) fM[s] = f;
    } else {
        /* which of the rows that can be overwritten this round (W, worst first) did slot s read? */
        const int32_t *p = pos + (size_t) s * (size_t) n;
        const int64_t rb = p[n - 1];
        int64_t al = rb + (rb >= i0 ? 1 : 0) + (int64_t) last[s];
        al += (al == i0) ? 1 : 0;
        int best = INT_MAX;
        for (int r = lane; r < nW; r += 64) {
            const int64_t a = W[r];
            if (a == i0) continue;                       /* the best row is never sampled */
            bool hit = (a == al);
            if (!hit && n > 1) {
                const int32_t rho = (int32_t) (a - (a > i0 ? 1 : 0));
                int lo = 0, hi = n - 2;                  /* binary search in the ascending picks */
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int32_t pv = p[mid];
                    if (pv == rho) { hit = true; break; }
                    if (pv < rho) lo = mid + 1; else hi = mid - 1;
                }
            }
            if (hit && r < best) best = r;
        }
        best = nla_wave_min_i32(best);
        if (lane == 0) minhz[s] = best;
    }
}

/* write accepted candidates back into the population (crs.c:153) */
__global__ __launch_bounds__(256) void crs_commit_kernel(int n, int ld, double *__restrict__ X, const double *__restrict__ TX,
                                                          const double *__restrict__ TM, const int32_t *__restrict__ slot,
                                                          const int32_t *__restrict__ kind, const int64_t *__restrict__ row)
{
    const int c = blockIdx.x;
    const double *src = (kind[c] == 1 ? TX : TM) + (size_t) slot[c] * (size_t) ld;
    double *dst = X + (size_t) row[c] * (size_t) ld;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

/* one local mutation in place (host-callback mode): p := clamp(best(1+w) - w p) */
__global__ __launch_bounds__(256) void crs_mutate_kernel(int n, const double *__restrict__ best, double *__restrict__ p,
                                                          const uint32_t *__restrict__ w, const double *__restrict__ lb,
                                                          const double *__restrict__ ub)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double wv = nla_urand_from(0., 1., w[2 * i], w[2 * i + 1]);
    p[i] = nla_clamp_box(best[i] * (1 + wv) - wv * p[i], lb[i], ub[i]);
}

/* ------------------------------------------------------------------------------------------------
 * launchers
 * ---------------------------------------------------------------------------------------------- */
extern "C" int nla_k_crs_init_rows(int obj, int n, int ld, const double *lb, const double *ub, const uint32_t *words,
                                   int64_t row_first, int64_t nrows, double *X, double *F, void *stream)
{
    if (nrows <= 0) return 0;
    const dim3 grid((unsigned) ((nrows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t) stream;
    if (obj < 0) {
        hipLaunchKernelGGL((crs_init_rows_kernel<-1>), grid, block, 0, st, n, ld, lb, ub, words, row_first, nrows, X, F);
    } else {
#define CALL(O) hipLaunchKernelGGL((crs_init_rows_kernel<O>), grid, block, 0, st, n, ld, lb, ub, words, row_first, nrows, X, F)
        NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    }
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_eval(int obj, int n, int ld, const double *P, int64_t count, double *F, void *stream)
{
    if (count <= 0) return 0;
    const dim3 grid((unsigned) ((count + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t) stream;
#define CALL(O) hipLaunchKernelGGL((eval_kernel<O>), grid, block, 0, st, n, ld, P, count, F)
    NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_vitter(int n, int64_t N, const uint32_t *words, int nblocks,
                                int32_t *jn, int32_t *pos, int32_t *last, void *stream)
{
    if (nblocks <= 0) return 0;
    hipLaunchKernelGGL(crs_vitter_kernel, dim3((unsigned) ((nblocks + 63) / 64)), dim3(64), 0, (hipStream_t) stream,
                       n, N, words, nblocks, jn, pos, last);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_gather(int n, int ld, const double *X, int64_t i0, const int32_t *jn, const int32_t *pos,
                                const int32_t *last, int K, const double *lb, const double *ub, double *TX, void *stream)
{
    if (K <= 0) return 0;
    hipStream_t st = (hipStream_t) stream;
    const bool vec2 = (n % 2 == 0) && (ld % 2 == 0) && n >= 128;
    const int cpw = vec2 ? 128 : 64;                        /* coordinates per wavefront */
    const int wps = (n + cpw - 1) / cpw;
    const long waves = (long) wps * K;
    const dim3 grid((unsigned) ((waves + 3) / 4)), block(256);
    /* few wavefronts in flight => deeper per-lane load pipelines */
    if (vec2) {
        if (waves <= 2048)
            hipLaunchKernelGGL((crs_gather_kernel<2, 32>), grid, block, 0, st, n, ld, X, i0, jn, pos, last, K, wps, lb, ub, TX);
        else
            hipLaunchKernelGGL((crs_gather_kernel<2, 16>), grid, block, 0, st, n, ld, X, i0, jn, pos, last, K, wps, lb, ub, TX);
    } else {
        hipLaunchKernelGGL((crs_gather_kernel<1, 16>), grid, block, 0, st, n, ld, X, i0, jn, pos, last, K, wps, lb, ub, TX);
    }
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_post(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                              const uint32_t *words_next, int K, const int64_t *W, int nW,
                              const int32_t *pos, const int32_t *last, const double *lb, const double *ub,
                              double *fT, double *fM, int32_t *minhz, void *stream)
{
    if (K <= 0) return 0;
    const dim3 grid((unsigned) (3 * K)), block(64);
    hipStream_t st = (hipStream_t) stream;
    if (obj < 0) {
        hipLaunchKernelGGL((crs_post_kernel<-1>), grid, block, 0, st, n, ld, X, i0, TX, TM, words_next, K, W, nW, pos, last, lb, ub, fT, fM, minhz);
    } else {
#define CALL(O) hipLaunchKernelGGL((crs_post_kernel<O>), grid, block, 0, st, n, ld, X, i0, TX, TM, words_next, K, W, nW, pos, last, lb, ub, fT, fM, minhz)
        NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    }
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_commit(int n, int ld, double *X, const double *TX, const double *TM, int ncommit,
                                const int32_t *slot, const int32_t *kind, const int64_t *row, void *stream)
{
    if (ncommit <= 0) return 0;
    hipLaunchKernelGGL(crs_commit_kernel, dim3((unsigned) ncommit), dim3(256), 0, (hipStream_t) stream,
                       n, ld, X, TX, TM, slot, kind, row);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_mutate(int n, const double *best, double *p, const uint32_t *words,
                                const double *lb, const double *ub, void *stream)
{
    hipLaunchKernelGGL(crs_mutate_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       n, best, p, words, lb, ub);
    NLA_LAUNCH_CHECK();
    return 0;
}
