/* dev_common.h — device-side building blocks shared by the gfx950 kernels.
 *
 * Wavefront = 64 lanes (CDNA4).  One wavefront evaluates one candidate: lane l owns coordinates
 * l, l+64, l+128, ... (coalesced 512-byte row segments), accumulates its partial in coordinate
 * order and the 64 partials are combined with a xor-butterfly (nla_xor_lane below: DPP moves and lane swaps, no LDS).
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

/* ---- the value lane (l ^ M) holds, M = 1, 2, 4, 8, 16, 32: one step of every xor-butterfly in this library.  __shfl_xor compiles to
 * ds_bpermute_b32 — an LDS-crossbar round trip per 32-bit half and step, ~12 of them in a row with a dependent fp64 add between two
 * (round 6: 1992 ds_bpermute in lbfgs_resident.hip's code object; the Strang loops do one such reduction per history column).  The same
 * lanes meet here through the VALU:   1, 2  quad permutes;  4  two row shifts under complementary bank masks;  8  a row rotation;
 * 16 / 32  gfx950's v_permlane16_swap / v_permlane32_swap (rows / halves of two copies trade places, each lane keeps the copy that
 * received its partner).  Same partner, same operands, same sum: bit for bit what the shuffle gave (tests/test_gpu_kernels.py
 * compares the two on every lane; every kernel's parity tests sit on top). ---- */
#ifdef NLA_SIMT_EMU
template <int M> __device__ __forceinline__ double nla_xor_lane(double v) { return __shfl_xor(v, M, NLA_WAVE); }
template <int M> __device__ __forceinline__ int nla_xor_lane(int v) { return __shfl_xor(v, M, NLA_WAVE); }
#else
typedef unsigned nla_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned nla_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
template <int M> __device__ __forceinline__ int nla_xor_lane(int v)
{
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "a lane distance of the butterfly");
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xF, 0xF, false);
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xF, 0xF, false);
    else if constexpr (M == 4) {
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104 /* row_shl:4: lane l <- l + 4 */, 0xF, 0x5 /* quads 0, 2 of a row */, false);
        return __builtin_amdgcn_update_dpp(t, v, 0x114 /* row_shr:4: lane l <- l - 4 */, 0xF, 0xA /* quads 1, 3 */, false);
    } else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128 /* row_ror:8 */, 0xF, 0xF, false);
    else if constexpr (M == 16) {
        const nla_u2 r = __builtin_amdgcn_permlane16_swap((unsigned) v, (unsigned) v, false, false);      /* x: rows 1, 3 now hold rows 0, 2;  y: rows 0, 2 hold rows 1, 3 */
        return (int) ((nla_lane_id() & 16u) ? r.x : r.y);
    } else {
        const nla_u2 r = __builtin_amdgcn_permlane32_swap((unsigned) v, (unsigned) v, false, false);      /* x: the upper half holds the lower;  y: the lower holds the upper */
        return (int) ((nla_lane_id() & 32u) ? r.x : r.y);
    }
}
template <int M> __device__ __forceinline__ double nla_xor_lane(double v)
{
    return __hiloint2double(nla_xor_lane<M>(__double2hiint(v)), nla_xor_lane<M>(__double2loint(v)));
}
#endif
/* a whole butterfly: STEP(M) for M = 32, 16, 8, 4, 2, 1 — the order every reduction of the library has always used */
#define NLA_BUTTERFLY(STEP) do { STEP(32); STEP(16); STEP(8); STEP(4); STEP(2); STEP(1); } while (0)

__device__ __forceinline__ double nla_wave_sum(double v)
{
#define NLA_S_(M) v += nla_xor_lane<M>(v)
    NLA_BUTTERFLY(NLA_S_);
#undef NLA_S_
    return v;
}
__device__ __forceinline__ double nla_wave_prod(double v)
{
#define NLA_S_(M) v *= nla_xor_lane<M>(v)
    NLA_BUTTERFLY(NLA_S_);
#undef NLA_S_
    return v;
}
__device__ __forceinline__ int nla_wave_min_i32(int v)
{
#define NLA_S_(M) { const int o = nla_xor_lane<M>(v); v = o < v ? o : v; }
    NLA_BUTTERFLY(NLA_S_);
#undef NLA_S_
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

/* Objective of one candidate, data-parallel over its coordinates.  Every objective of the zoo is
 * (sum a, second sum or product b) over per-coordinate terms plus a scalar finish, so a partial
 * result is the pair (a, b); partials combine by (+, + or *).  Thread `first` of `stride`
 * accumulates coordinates first, first+stride, ... in that order; the 64 partials of a wavefront
 * combine with a xor-butterfly, the wavefronts of a workgroup in wavefront order through LDS. */
struct nla_obj_part { double a, b; };

template <int OBJ>
__device__ __forceinline__ nla_obj_part nla_obj_combine(nla_obj_part x, nla_obj_part y)
{
    nla_obj_part r;
    r.a = x.a + y.a;
    r.b = (OBJ == NLA_OBJ_GRIEWANK) ? x.b * y.b : x.b + y.b;
    return r;
}

template <int OBJ, class Get>
__device__ __forceinline__ nla_obj_part nla_obj_partial(int n, int first, int stride, Get get)
{
    nla_obj_part r;
    r.a = 0;
    r.b = (OBJ == NLA_OBJ_GRIEWANK) ? 1 : 0;
    if (OBJ == NLA_OBJ_RASTRIGIN) {
        for (int i = first; i < n; i += stride) r.a += nla_rastrigin_term(get(i));
    } else if (OBJ == NLA_OBJ_ACKLEY) {
        for (int i = first; i < n; i += stride) { double x = get(i); r.a += nla_sqr(x); r.b += nla_ackley_cos_term(x); }
    } else if (OBJ == NLA_OBJ_GRIEWANK) {
        for (int i = first; i < n; i += stride) {
            double x = get(i);
            r.a += nla_griewank_sum_term(x);
            r.b *= nla_griewank_prod_term(x, (unsigned) i);
        }
    } else if (OBJ == NLA_OBJ_ROSENBROCK) {
        for (int i = first; i + 1 < n; i += stride) r.a += nla_rosenbrock_term(get(i), get(i + 1));
    } else if (OBJ == NLA_OBJ_LEVY) {
        for (int i = first; i + 1 < n; i += stride) r.a += nla_levy_term(get(i), get(i + 1));
    } else { /* NLA_OBJ_SPHERE */
        for (int i = first; i < n; i += stride) r.a += nla_sqr(get(i));
    }
    return r;
}

template <int OBJ, class Get>
__device__ __forceinline__ double nla_obj_finish(int n, nla_obj_part t, Get get)
{
    if (OBJ == NLA_OBJ_RASTRIGIN) return 10.0 * n + t.a;
    if (OBJ == NLA_OBJ_ACKLEY) return nla_ackley_finish(t.a, t.b, (unsigned) n);
    if (OBJ == NLA_OBJ_GRIEWANK) return (1.0 + t.a) - t.b;
    if (OBJ == NLA_OBJ_LEVY) return nla_levy_head(get(0), get(n - 1)) + t.a;
    return t.a;
}

template <int OBJ>
__device__ __forceinline__ nla_obj_part nla_obj_wave_reduce(nla_obj_part t)
{
#define NLA_S_(M) { nla_obj_part o; o.a = nla_xor_lane<M>(t.a); o.b = nla_xor_lane<M>(t.b); t = nla_obj_combine<OBJ>(t, o); }
    NLA_BUTTERFLY(NLA_S_);
#undef NLA_S_
    return t;
}

/* one wavefront per candidate; `get(i)` returns coordinate i (a load, or a value recomputed on the
 * fly); every lane returns the full f */
template <int OBJ, class Get>
__device__ __forceinline__ double nla_wave_objective(int n, Get get)
{
    const int lane = threadIdx.x & (NLA_WAVE - 1);
    return nla_obj_finish<OBJ>(n, nla_obj_wave_reduce<OBJ>(nla_obj_partial<OBJ>(n, lane, NLA_WAVE, get)), get);
}

/* one workgroup of WAVES wavefronts per candidate (blockDim.x == 64*WAVES, all threads call);
 * `scratch` = 2*WAVES doubles of LDS; every thread returns the full f */
template <int OBJ, int WAVES, class Get>
__device__ __forceinline__ double nla_block_objective(int n, Get get, double *scratch)
{
    const int lane = threadIdx.x & (NLA_WAVE - 1), wave = threadIdx.x >> 6;
    nla_obj_part t = nla_obj_wave_reduce<OBJ>(nla_obj_partial<OBJ>(n, (int) threadIdx.x, NLA_WAVE * WAVES, get));
    __syncthreads();                      /* scratch may still be read from a previous call */
    if (lane == 0) { scratch[2 * wave] = t.a; scratch[2 * wave + 1] = t.b; }
    __syncthreads();
    t.a = scratch[0]; t.b = scratch[1];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) {
        nla_obj_part o;
        o.a = scratch[2 * w]; o.b = scratch[2 * w + 1];
        t = nla_obj_combine<OBJ>(t, o);
    }
    return nla_obj_finish<OBJ>(n, t, get);
}

/* the same f, bit for bit, as nla_block_objective<OBJ, VW> gives in a workgroup of VW wavefronts — computed by a workgroup of WAVES <= VW
 * wavefronts: each wavefront plays the virtual wavefronts wave, wave + WAVES, ... (same elements per lane, same butterfly), the partial
 * results are combined in the same order.  `scratch` = 2*VW doubles.  (The CRS2_LM chain kernel runs 1 - 8 wavefronts per workgroup
 * depending on n; the finish kernel of the conservative passes and the column-sharded job always 8: one reduction for all of them.) */
template <int OBJ, int WAVES, int VW, class Get>
__device__ __forceinline__ double nla_block_objective_as(int n, Get get, double *scratch)
{
    static_assert(WAVES <= VW, "more wavefronts than virtual ones");
    const int lane = threadIdx.x & (NLA_WAVE - 1), wave = threadIdx.x >> 6;
    __syncthreads();                      /* scratch may still be read from a previous call */
    for (int vw = wave; vw < VW; vw += WAVES) {
        const nla_obj_part t = nla_obj_wave_reduce<OBJ>(nla_obj_partial<OBJ>(n, vw * NLA_WAVE + lane, NLA_WAVE * VW, get));
        if (lane == 0) { scratch[2 * vw] = t.a; scratch[2 * vw + 1] = t.b; }
    }
    __syncthreads();
    nla_obj_part t;
    t.a = scratch[0]; t.b = scratch[1];
#pragma unroll
    for (int w = 1; w < VW; ++w) {
        nla_obj_part o;
        o.a = scratch[2 * w]; o.b = scratch[2 * w + 1];
        t = nla_obj_combine<OBJ>(t, o);
    }
    return nla_obj_finish<OBJ>(n, t, get);
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

#ifndef NLA_OBJ_NEGATE
#define NLA_OBJ_NEGATE 0x100            /* (include/nlopt_amd.h) */
#endif
/* launcher side of NLA_OBJ_NEGATE: strips the flag from obj, yields the factor the kernel multiplies f by */
static inline double nla_obj_sign(int *obj)
{
    if (*obj >= 0 && (*obj & NLA_OBJ_NEGATE)) { *obj &= ~NLA_OBJ_NEGATE; return -1.; }
    return 1.;
}

#define NLA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int) e_; } while (0)

#endif
