/* dev_common.h — device-side building blocks shared by the gfx950 kernels.
 *
 * Wavefront = 64 lanes (CDNA4).  One wavefront evaluates one candidate: lane l owns coordinates
 * l, l+64, l+128, ... (coalesced 512-byte row segments), accumulates its partial in coordinate
 * order and the 64 partials are combined with a xor-butterfly (DPP/ds_swizzle shuffles, no LDS).
 * The per-element terms are the *same source* as the host callbacks (../objfuncs.h); only the
 * association order of the final reduction differs from the sequential host loop, which moves f
 * by O(1e-16) relative — inside the 1e-10 tolerance of the parity contract.
 *
 * All arithmetic that feeds *x* (population rows, trial points) must be bit-identical to the
 * reference, which is built with -ffp-contract=off (CMakeLists.txt:280-284): this directory is
 * compiled with -ffp-contract=off and every file also carries the pragma below.
 */
#ifndef NLA_DEV_COMMON_H
#define NLA_DEV_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../objfuncs.h"

#pragma clang fp contract(off)

#define NLA_WAVE 64

__device__ __forceinline__ double nla_wave_sum(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, NLA_WAVE);
    return v;
}
__device__ __forceinline__ double nla_wave_prod(double v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v *= __shfl_xor(v, m, NLA_WAVE);
    return v;
}
__device__ __forceinline__ int nla_wave_min_i32(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m, NLA_WAVE); v = o < v ? o : v; }
    return v;
}

/* genrand_res53 from two consecutive tempered words, first word = high 27 bits
 * (src/util/mt19937ar.c:194-198) */
__device__ __forceinline__ double nla_res53(uint32_t w0, uint32_t w1)
{
    uint32_t a = w0 >> 5, b = w1 >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* nlopt_urand(a,b) = a + (b-a)*res53, src/util/mt19937ar.c:203-206 (three roundings, no FMA) */
__device__ __forceinline__ double nla_urand_from(double a, double b, uint32_t w0, uint32_t w1)
{
    return a + (b - a) * nla_res53(w0, w1);
}

__device__ __forceinline__ double nla_clamp_box(double v, double lo, double hi)
{   /* if (x > ub) x = ub; else if (x < lb) x = lb;   (crs.c:118-119) */
    if (v > hi) v = hi;
    else if (v < lo) v = lo;
    return v;
}

/* Objective of one candidate by one wavefront.  `get(i)` returns coordinate i (a load, or a value
 * recomputed on the fly); every lane returns the full f. */
template <int OBJ, class Get>
__device__ __forceinline__ double nla_wave_objective(int n, Get get)
{
    const int lane = threadIdx.x & (NLA_WAVE - 1);
    if (OBJ == NLA_OBJ_RASTRIGIN) {
        double s = 0;
        for (int i = lane; i < n; i += NLA_WAVE) s += nla_rastrigin_term(get(i));
        return 10.0 * n + nla_wave_sum(s);
    } else if (OBJ == NLA_OBJ_ACKLEY) {
        double s = 0, c = 0;
        for (int i = lane; i < n; i += NLA_WAVE) { double x = get(i); s += nla_sqr(x); c += nla_ackley_cos_term(x); }
        return nla_ackley_finish(nla_wave_sum(s), nla_wave_sum(c), (unsigned) n);
    } else if (OBJ == NLA_OBJ_GRIEWANK) {
        double s = 0, p = 1;
        for (int i = lane; i < n; i += NLA_WAVE) {
            double x = get(i);
            s += nla_griewank_sum_term(x);
            p *= nla_griewank_prod_term(x, (unsigned) i);
        }
        return (1.0 + nla_wave_sum(s)) - nla_wave_prod(p);
    } else if (OBJ == NLA_OBJ_ROSENBROCK) {
        double s = 0;
        for (int i = lane; i + 1 < n; i += NLA_WAVE) s += nla_rosenbrock_term(get(i), get(i + 1));
        return nla_wave_sum(s);
    } else if (OBJ == NLA_OBJ_LEVY) {
        double s = 0;
        for (int i = lane; i + 1 < n; i += NLA_WAVE) s += nla_levy_term(get(i), get(i + 1));
        double head = nla_levy_head(get(0), get(n - 1));
        return head + nla_wave_sum(s);
    } else { /* NLA_OBJ_SPHERE */
        double s = 0;
        for (int i = lane; i < n; i += NLA_WAVE) s += nla_sqr(get(i));
        return nla_wave_sum(s);
    }
}

#define NLA_OBJ_DISPATCH(obj, CALL)                                   \
    switch (obj) {                                                    \
    case NLA_OBJ_RASTRIGIN:  { CALL(NLA_OBJ_RASTRIGIN);  } break;     \
    case NLA_OBJ_ACKLEY:     { CALL(NLA_OBJ_ACKLEY);     } break;     \
    case NLA_OBJ_GRIEWANK:   { CALL(NLA_OBJ_GRIEWANK);   } break;     \
    case NLA_OBJ_ROSENBROCK: { CALL(NLA_OBJ_ROSENBROCK); } break;     \
    case NLA_OBJ_LEVY:       { CALL(NLA_OBJ_LEVY);       } break;     \
    case NLA_OBJ_SPHERE:     { CALL(NLA_OBJ_SPHERE);     } break;     \
    default: return (int) hipErrorInvalidValue;                       \
    }

#define NLA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int) e_; } while (0)

#endif
