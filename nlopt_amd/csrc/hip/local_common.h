/* local_common.h — what the batched local optimisers (lbfgs_kernels.hip, mma_kernels.hip) share: one workgroup of
 * LB_T threads per local search, workgroup reductions with a fixed tree (thread-strided partials, xor-butterfly per
 * wavefront, wavefronts in order) and the objective + gradient evaluation by the workgroup. */
#ifndef NLA_LOCAL_COMMON_H
#define NLA_LOCAL_COMMON_H
#include "dev_common.h"

#ifndef LB_T
#define LB_T 256                  /* (tools/simt_emu builds may shrink the workgroup: -DLB_T=128) */
#endif
#define LB_W (LB_T / 64)

struct lb_shared {
    double red[2 * LB_W];
    int ired[2 * LB_W];
};

__device__ __forceinline__ double lb_wave_sum(double v)
{
#define LB_S_(M) v += nla_xor_lane<M>(v)
    NLA_BUTTERFLY(LB_S_);
#undef LB_S_
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
#define LB_S_(M) { const double o = nla_xor_lane<M>(v); v = o > v ? o : v; }
    NLA_BUTTERFLY(LB_S_);
#undef LB_S_
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
#define LB_S_(M) v += nla_xor_lane<M>(v)
    NLA_BUTTERFLY(LB_S_);
#undef LB_S_
    __syncthreads();
    if ((threadIdx.x & 63) == 0) S.ired[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = S.ired[0];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) t += S.ired[w];
    return t;
}

/* ---- exact-order mode ("amd_exact_dot", nla_lbfgs_params.exact / nla_mma_params.exact) --------------------------------
 * The reference accumulates every dot product, norm and objective sum in ONE accumulator over i = 0 .. n-1
 * (mssubs.c:601-641 mxudot, stop.c:37-57 vector_norm, mma.c:88-124 dual_func, the zoo's loops).  In exact mode the terms
 * are formed in parallel into LDS and then added in that order by every thread redundantly (LDS broadcast reads), so all
 * threads hold the bit-identical sum the sequential host loop produces.  A verification mode: n serial additions per
 * reduction instead of log2(256) + n/256. */
#ifndef LB_XCH
#define LB_XCH 1024                 /* (cobyla_kernels.hip: 64 — its searches have n <= 64 and want the LDS for their own state) */
#endif
struct lb_exact_buf { double a[LB_XCH]; double b[LB_XCH]; };

template <class Term>
__device__ __forceinline__ double lb_seq_sum(int n, double init, Term term, double *buf)
{
    double acc = init;
    for (int base = 0; base < n; base += LB_XCH) {
        const int m = n - base < LB_XCH ? n - base : LB_XCH;
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += LB_T) buf[i] = term(base + i);
        __syncthreads();
        for (int i = 0; i < m; ++i) acc += buf[i];
    }
    return acc;
}
/* two sums over the same index range at once (terms written by `fill(i, &a, &b)`), second one optionally a product */
template <bool PROD, class Fill>
__device__ __forceinline__ void lb_seq_sum2(int first, int n, double *acc_a, double *acc_b, Fill fill, lb_exact_buf &B)
{
    double a = *acc_a, b = *acc_b;
    for (int base = first; base < n; base += LB_XCH) {
        const int m = n - base < LB_XCH ? n - base : LB_XCH;
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += LB_T) fill(base + i, &B.a[i], &B.b[i]);
        __syncthreads();
        for (int i = 0; i < m; ++i) { a += B.a[i]; if (PROD) b *= B.b[i]; else b += B.b[i]; }
    }
    *acc_a = a; *acc_b = b;
}

/* masked dot product (mxudot, job > 0): coordinates on an active bound (ix < 0) are skipped */
__device__ __forceinline__ double lb_mdot(int n, const double *x, const double *y, const int *ix, lb_shared &S)
{
    double t = 0;
    for (int i = threadIdx.x; i < n; i += LB_T) if (ix[i] >= 0) t += x[i] * y[i];
    return lb_block_sum(t, S);
}
/* the same in the reference's summation order (mssubs.c:601-641); a skipped coordinate contributes +0.0 */
__device__ __forceinline__ double lb_mdot_exact(int n, const double *x, const double *y, const int *ix, lb_exact_buf &B)
{
    return lb_seq_sum(n, 0., [&](int i) { return ix[i] >= 0 ? x[i] * y[i] : 0.; }, B.a);
}

/* f in the host callback's summation order (../objfuncs.h nla_obj_eval_seq): one accumulator over i ascending, the
 * accumulator's start value as there.  *t receives the sums the gradient formulas need. */
template <int OBJ>
__device__ __forceinline__ double lb_obj_exact(int n, const double *x, nla_obj_part *t, lb_exact_buf &B)
{
    double a = 0, b = 0;
    if (OBJ == NLA_OBJ_RASTRIGIN) {
        a = lb_seq_sum(n, 10.0 * n, [&](int i) { return nla_rastrigin_term(x[i]); }, B.a);
        t->a = a; t->b = 0;
        return a;
    } else if (OBJ == NLA_OBJ_ACKLEY) {
        lb_seq_sum2<false>(0, n, &a, &b, [&](int i, double *pa, double *pb) { *pa = nla_sqr(x[i]); *pb = nla_ackley_cos_term(x[i]); }, B);
        t->a = a; t->b = b;
        return nla_ackley_finish(a, b, (unsigned) n);
    } else if (OBJ == NLA_OBJ_GRIEWANK) {
        a = 1; b = 1;
        lb_seq_sum2<true>(0, n, &a, &b, [&](int i, double *pa, double *pb) { *pa = nla_griewank_sum_term(x[i]); *pb = nla_griewank_prod_term(x[i], (unsigned) i); }, B);
        t->a = a; t->b = b;
        return a - b;
    } else if (OBJ == NLA_OBJ_ROSENBROCK) {
        a = lb_seq_sum(n - 1, 0., [&](int i) { return nla_rosenbrock_term(x[i], x[i + 1]); }, B.a);
    } else if (OBJ == NLA_OBJ_LEVY) {
        a = lb_seq_sum(n - 1, nla_levy_head(x[0], x[n - 1]), [&](int i) { return nla_levy_term(x[i], x[i + 1]); }, B.a);
    } else {
        a = lb_seq_sum(n, 0., [&](int i) { return nla_sqr(x[i]); }, B.a);
    }
    t->a = a; t->b = 0;
    return a;
}

/* objective and gradient of one point by the workgroup (same per-element formulas as the host
 * callbacks in ../objfuncs.h nla_obj_eval_seq); exact != 0: f summed in the host's order; sign = -1: the maximisation
 * wrapper of the reference (f_max, optimize.c:970-980: -f and -gradient) */
template <int OBJ>
__device__ __forceinline__ double lb_objgrad(int n, const double *x, double *g, lb_shared &S, double *scratch, int exact,
                                             lb_exact_buf &XB, double sign)
{
    const int tid = threadIdx.x;
    nla_obj_part t;
    double f;
    if (exact) f = lb_obj_exact<OBJ>(n, x, &t, XB);
    else {
        t = nla_obj_wave_reduce<OBJ>(nla_obj_partial<OBJ>(n, tid, LB_T, [&](int i) { return x[i]; }));
        __syncthreads();
        if ((tid & 63) == 0) { scratch[2 * (tid >> 6)] = t.a; scratch[2 * (tid >> 6) + 1] = t.b; }
        __syncthreads();
        t.a = scratch[0]; t.b = scratch[1];
#pragma unroll
        for (int w = 1; w < LB_W; ++w) { nla_obj_part o; o.a = scratch[2 * w]; o.b = scratch[2 * w + 1]; t = nla_obj_combine<OBJ>(t, o); }
        f = nla_obj_finish<OBJ>(n, t, [&](int i) { return x[i]; });
    }
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
    if (sign < 0) {
        for (int i = tid; i < n; i += LB_T) g[i] = -g[i];
        __syncthreads();
        f = -f;
    }
    return f;
}


/* the host's abort flag (pinned memory: 0, 100 = time limit, -999 = forced stop), read ONCE per poll for the whole workgroup: the
 * flag changes while the kernel runs, and threads that read it separately could see different values, leave a loop at
 * different iterations and hang at the next barrier */
__device__ __forceinline__ int lb_poll_abort(const int32_t *flag)
{
    __shared__ int s_abort_seen;
    __syncthreads();
    if (threadIdx.x == 0) s_abort_seen = *(const volatile int32_t *) flag;
    __syncthreads();
    return s_abort_seen;
}

#endif
