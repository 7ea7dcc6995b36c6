/* local_common.h — what the batched local optimisers (lbfgs_kernels.hip, mma_kernels.hip) share: one workgroup of
 * LB_T threads per local search, workgroup reductions with a fixed tree (thread-strided partials, xor-butterfly per
 * wavefront, wavefronts in order) and the objective + gradient evaluation by the workgroup. */
#ifndef NLA_LOCAL_COMMON_H
#define NLA_LOCAL_COMMON_H
#include "dev_common.h"

#define LB_T 256
#define LB_W (LB_T / 64)

struct lb_shared {
    double red[2 * LB_W];
    int ired[2 * LB_W];
};

__device__ __forceinline__ double lb_wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}
__device__ __forceinline__ double lb_block_sum(double v, lb_shared &S)
{
    v = lb_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) S.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = S.red[0];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) t += S.red[w];
    return t;
}
__device__ __forceinline__ double lb_block_max(double v, lb_shared &S)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const double o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) S.red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = S.red[0];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) t = S.red[w] > t ? S.red[w] : t;
    return t;
}
__device__ __forceinline__ double lb_block_min(double v, lb_shared &S) { return -lb_block_max(-v, S); }
__device__ __forceinline__ int lb_block_isum(int v, lb_shared &S)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) S.ired[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = S.ired[0];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) t += S.ired[w];
    return t;
}

/* masked dot product (mxudot, job > 0): coordinates on an active bound (ix < 0) are skipped */
__device__ __forceinline__ double lb_mdot(int n, const double *x, const double *y, const int *ix, lb_shared &S)
{
    double t = 0;
    for (int i = threadIdx.x; i < n; i += LB_T) if (ix[i] >= 0) t += x[i] * y[i];
    return lb_block_sum(t, S);
}

/* objective and gradient of one point by the workgroup (same per-element formulas as the host
 * callbacks in ../objfuncs.h nla_obj_eval_seq) */
template <int OBJ>
__device__ __forceinline__ double lb_objgrad(int n, const double *x, double *g, lb_shared &S, double *scratch)
{
    const int tid = threadIdx.x;
    nla_obj_part t = nla_obj_wave_reduce<OBJ>(nla_obj_partial<OBJ>(n, tid, LB_T, [&](int i) { return x[i]; }));
    __syncthreads();
    if ((tid & 63) == 0) { scratch[2 * (tid >> 6)] = t.a; scratch[2 * (tid >> 6) + 1] = t.b; }
    __syncthreads();
    t.a = scratch[0]; t.b = scratch[1];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) { nla_obj_part o; o.a = scratch[2 * w]; o.b = scratch[2 * w + 1]; t = nla_obj_combine<OBJ>(t, o); }
    const double f = nla_obj_finish<OBJ>(n, t, [&](int i) { return x[i]; });
    if (OBJ == NLA_OBJ_RASTRIGIN) {
        for (int i = tid; i < n; i += LB_T) g[i] = 2 * x[i] + 10.0 * NLA_PI2 * sin(NLA_PI2 * x[i]);
    } else if (OBJ == NLA_OBJ_ACKLEY) {
        const double r = sqrt(t.a / (unsigned) n), e1 = exp(-0.2 * r), e2 = exp(t.b / (unsigned) n);
        for (int i = tid; i < n; i += LB_T) {
            double gi = e2 * NLA_PI2 * sin(NLA_PI2 * x[i]) / (unsigned) n;
            if (r > 0) gi += 4.0 * e1 * x[i] / ((unsigned) n * r);
            g[i] = gi;
        }
    } else if (OBJ == NLA_OBJ_GRIEWANK) {
        for (int i = tid; i < n; i += LB_T) {
            const double sq = sqrt(i + 1.);
            g[i] = x[i] * 0.0005 + t.b * tan(x[i] / sq) / sq;
        }
    } else if (OBJ == NLA_OBJ_ROSENBROCK) {
        for (int i = tid; i < n; i += LB_T) {
            double gi = 0;
            if (i > 0) { const double a = x[i] - x[i - 1] * x[i - 1]; gi = 200 * a; }
            if (i + 1 < n) { const double a = x[i + 1] - x[i] * x[i], b = 1 - x[i]; gi += -400 * a * x[i] - 2 * b; }
            g[i] = gi;
        }
    } else if (OBJ == NLA_OBJ_LEVY) {
        for (int i = tid; i < n; i += LB_T) {
            double gi = 0;
            if (i == 0) gi = 2 * NLA_PI3 * sin(NLA_PI3 * x[0]) * cos(NLA_PI3 * x[0]);
            if (i == n - 1) {
                const double a = x[n - 1] - 1, b = 1 + nla_sqr(sin(NLA_PI2 * x[n - 1]));
                gi += b + a * 2 * NLA_PI2 * sin(NLA_PI2 * x[n - 1]) * cos(NLA_PI2 * x[n - 1]);
            }
            if (i + 1 < n) { const double a = x[i] - 1, b = 1 + nla_sqr(sin(NLA_PI3 * x[i + 1])); gi += 2 * a * b; }
            if (i > 0) { const double a = x[i - 1] - 1; gi += 2 * NLA_PI3 * nla_sqr(a) * sin(NLA_PI3 * x[i]) * cos(NLA_PI3 * x[i]); }
            g[i] = gi;
        }
    } else {
        for (int i = tid; i < n; i += LB_T) g[i] = 2 * x[i];
    }
    __syncthreads();
    return f;
}

#endif
