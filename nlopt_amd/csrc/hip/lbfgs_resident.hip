/* lbfgs_resident.hip — the batched NLOPT_LD_LBFGS search (Luksan's PLIS, src/algs/luksan/plis.c:106-417) for n <= 4096 with a
 * compiled-in device objective, tree-reduction sums: the kernel config 4 runs (G_MLSL + LD_LBFGS, Ackley n = 4096).
 *
 * Why a second kernel (round 4).  The phase profile of lbfgs_batch_kernel (lbfgs_kernels.hip) on the MI355X
 * (profiles/r04_lbfgs_phase_profile_streaming.txt) showed that the Strang recurrences — the part the roofline model prices — were 20 % of a
 * search; the rest were the short vector loops around them (project, pytrcs, pytrcd, the line search's x update ...), each 10-50 us
 * although it moves 100 KB: the vectors lived in global memory, a thread walked its 16 coordinates with a load -> store round trip
 * per coordinate (the compiler may not move a load above a store that might alias it), i.e. 16 dependent HBM/L2 latencies per loop;
 * and the scalar state of the line search (some 60 doubles, uniform but held in VGPRs) pushed the kernel to 62-107 spilled registers.
 * Here:
 *   x and the gradient live in LDS (2 x 32 KB per workgroup, two workgroups per CU), the bound type ix as bytes in LDS;
 *   the search direction s lives in REGISTERS for the whole search (thread t owns coordinates t, t + 256, ...: 16 values);
 *   what stays in global memory (the history columns; xl / xu, read-only after the set-up) is loaded 16 independent loads at a time;
 *   the scalar state (line search PS1L01, termination PYFUT1, counters) is ONE copy in LDS that thread 0 advances between two
 *   barriers — the other threads read the few values they need (the step r, the next phase) after the barrier;
 *   two sums that are needed at the same point share one reduction (|g|^2 with x1.g1, |s|^2 with g.s, |x|_1 with |dx|_1).
 * Every sum keeps the summation tree of lbfgs_batch_kernel's default mode (thread-strided partials in coordinate order,
 * xor-butterfly per wavefront, wavefronts in order), every per-element formula and the order of the scalar logic are the same:
 * the two kernels produce bit-identical searches (tests/test_gpu_lbfgs.py::test_resident_kernel_is_the_streaming_kernel), so the
 * parity statement of the streaming kernel carries over — sums differ from the reference's sequential ones by rounding only;
 * "amd_exact_dot" = 1 (reference order, bit for bit) runs on lbfgs_batch_kernel.
 *
 * Roofline: the history stream, 32 k n bytes per iteration (k columns, two matrices, a dot and an axpy pass each) — the only
 * HBM traffic left besides xl / xu; bound by the two dependent workgroup reductions per history column.
 */
#include "local_common.h"
#include <limits.h>
#include <type_traits>
#include "../lbfgs_scalar.h"
#include "../../../include/nlopt_amd.h"

/* The file is compiled twice: as itself with 16 coordinates per thread (n <= 4096, two workgroups per compute unit — config 4's kernel), and
 * through lbfgs_resident32.hip with 32 (LR_WIDE: 4096 < n <= 8192, x and the gradient 2 x 64 KB of LDS, ONE workgroup per compute unit, the
 * direction 64 VGPRs) — the same source, the same sums, so the same search bit for bit as the streaming kernel there too. */
#ifdef LR_WIDE
#define LR_E 32
#define LR_PER_EU 1
#define LR_KERNEL lbfgs_resident32_kernel
#define LR_SUPPORTED nla_lbfgs_resident32_supported
#define LR_BATCH nla_k_lbfgs_batch_resident32
#else
#define LR_E 16                        /* coordinates per thread */
#define LR_PER_EU 2
#define LR_KERNEL lbfgs_resident_kernel
#define LR_SUPPORTED nla_lbfgs_resident_supported
#define LR_BATCH nla_k_lbfgs_batch_resident
#endif
#define LR_NMAX (LB_T * LR_E)          /* 4096 (8192) */

namespace {                            /* (the two builds of this file define the same names with different LR_E: internal linkage) */

/* the one copy of a search's scalar state (LDS) */
struct lr_ctl {
    lb_ls_state lss; lb_ls_io q; lb_counters c; lb_stop ls;
    double gmax, umax, fval, fo, p, po, gnorm, snorm, rmax, rmin, b, xtol_rel, tolg;
    int kd, nred, maxst, xstop, nevals, k, cols, head, forced, tmo, go;
};
struct lr_red { double v[2][LB_W][2]; int iv[2][LB_W]; };
enum { LR_EXIT = 1, LR_RELEASE, LR_CONTINUE, LR_STRANG, LR_STEEPEST, LR_AGAIN, LR_LS_EVAL, LR_RESTORE, LR_PYTRCD, LR_NO_STEP, LR_XTOL_ABS };

#if defined(NLA_LB_PROF) && !defined(LR_WIDE)     /* tools/lbfgs_prof.py: per-phase device time of every search (10 ns ticks); the shipped library has none of this */
#define LB_PROF_PHASES 12
#define LB_PROF_CAP 4096
__device__ unsigned long long nla_lb_prof[LB_PROF_CAP][LB_PROF_PHASES + 4];
#define PROF_DECL unsigned long long pf_acc[LB_PROF_PHASES] = {}, pf_last = wall_clock64(), pf_iters = 0
#define PROF(i) do { const unsigned long long pf_t = wall_clock64(); pf_acc[i] += pf_t - pf_last; pf_last = pf_t; } while (0)
#define PROF_ITER ++pf_iters
#define PROF_STORE do { if (tid == 0 && inst < LB_PROF_CAP) { for (int pf_i = 0; pf_i < LB_PROF_PHASES; ++pf_i) nla_lb_prof[inst][pf_i] = pf_acc[pf_i]; \
        nla_lb_prof[inst][LB_PROF_PHASES] = pf_iters; nla_lb_prof[inst][LB_PROF_PHASES + 1] = C.nevals; nla_lb_prof[inst][LB_PROF_PHASES + 2] = C.cols; \
        nla_lb_prof[inst][LB_PROF_PHASES + 3] = 0; } } while (0)
}
extern "C" int nla_lbfgs_prof_read(unsigned long long *out, int count)
{
    if (count > LB_PROF_CAP) count = LB_PROF_CAP;
    return (int) hipMemcpyFromSymbol(out, HIP_SYMBOL(nla_lb_prof), sizeof(unsigned long long) * (LB_PROF_PHASES + 4) * (size_t) count);
}
namespace {
#else
#define PROF_DECL
#define PROF(i)
#define PROF_ITER
#define PROF_STORE
#endif

/* two reductions at once, ONE barrier: the partials alternate between two LDS slots (`par`, uniform), so a wavefront may already
 * write the next reduction's partial while a slower one still reads this one's — the next write to the SAME slot is two
 * reductions later, behind the barrier in between, which every thread passes only after it has read this one's.  The sums are
 * lb_block_sum's (local_common.h): xor-butterfly per wavefront, wavefronts in order; the maxima lb_block_max's. */
template <bool MAX>
__device__ __forceinline__ void lr_reduce2(double &a, double &b, lr_red &R, int &par)
{
#define LR_S_(M) { const double oa = nla_xor_lane<M>(a), ob = nla_xor_lane<M>(b); if (MAX) { a = oa > a ? oa : a; b = ob > b ? ob : b; } else { a += oa; b += ob; } }
    NLA_BUTTERFLY(LR_S_);
#undef LR_S_
    if ((threadIdx.x & 63) == 0) { R.v[par][threadIdx.x >> 6][0] = a; R.v[par][threadIdx.x >> 6][1] = b; }
    __syncthreads();
    a = R.v[par][0][0]; b = R.v[par][0][1];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) {
        const double oa = R.v[par][w][0], ob = R.v[par][w][1];
        if (MAX) { a = oa > a ? oa : a; b = ob > b ? ob : b; }
        else { a += oa; b += ob; }
    }
    par ^= 1;
}
template <bool MAX>
__device__ __forceinline__ double lr_reduce1(double a, lr_red &R, int &par)
{
#define LR_S_(M) { const double oa = nla_xor_lane<M>(a); if (MAX) a = oa > a ? oa : a; else a += oa; }
    NLA_BUTTERFLY(LR_S_);
#undef LR_S_
    if ((threadIdx.x & 63) == 0) R.v[par][threadIdx.x >> 6][0] = a;
    __syncthreads();
    a = R.v[par][0][0];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) { const double oa = R.v[par][w][0]; if (MAX) a = oa > a ? oa : a; else a += oa; }
    par ^= 1;
    return a;
}
__device__ __forceinline__ int lr_reduce_isum(int a, lr_red &R, int &par)
{
#define LR_S_(M) a += nla_xor_lane<M>(a)
    NLA_BUTTERFLY(LR_S_);
#undef LR_S_
    if ((threadIdx.x & 63) == 0) R.iv[par][threadIdx.x >> 6] = a;
    __syncthreads();
    a = R.iv[par][0];
#pragma unroll
    for (int w = 1; w < LB_W; ++w) a += R.iv[par][w];
    par ^= 1;
    return a;
}

/* ---- sums in the REFERENCE'S order ("amd_exact_dot" = 1): one accumulator over i = 0 .. n-1 (mssubs.c:601-641 mxudot, stop.c:37-57,
 * the zoo's loops).  The coordinates come in blocks of LB_T consecutive ones (thread t holds t + LB_T e): a block's terms are
 * staged in LDS, every wavefront then adds them up in order for itself — 64 terms per LDS read, handed to the accumulator one by
 * one with v_readlane (the value travels through scalar registers: no LDS traffic in the dependent chain) — so all threads hold the
 * bit-identical sum the sequential host loop produces.  Two staging sets used in turn: ONE barrier per block.  Two sums (or a sum
 * and a product: Griewank) run as two independent chains in one pass. */
struct lr_xbuf { double a[2][LB_T], b[2][LB_T]; };
struct lr_nobuf { };
__device__ __forceinline__ double lr_lane(double v, int l)
{
#ifdef NLA_SIMT_EMU
    (void) l; return v;
#else
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
#endif
}
/* Sixteen terms of the chain in sixteen instructions (round 6; the readlane form above it is three per term — two v_readlane_b32 and the add —
 * and measured 2.5 ns per term, which is what a search costs in this mode): v_fmac_f64 is the one double-precision VOP2 instruction, so it
 * takes a DPP operand, and row_newbcast:j hands every lane lane j OF ITS ROW of 16 — with the same sixteen terms in each of the wavefront's
 * four rows every lane accumulates a + t_0 + t_1 + ... in order.  fma(t, 1.0, a) rounds t + a once: the add, bit for bit.
 * (s_nop: a VGPR written by a vector instruction may not be read through DPP by the next two.) */
#ifndef NLA_SIMT_EMU
#define LR_FD_(acc, v, J) "v_fmac_f64_dpp " acc ", " v ", %[one] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void lr_chain16(double &a, double va)
{
    const double one = 1.0;
#define LR_C_(J) LR_FD_("%[a]", "%[va]", J)
    asm volatile("s_nop 1\n\t" LR_C_(0) LR_C_(1) LR_C_(2) LR_C_(3) LR_C_(4) LR_C_(5) LR_C_(6) LR_C_(7) LR_C_(8) LR_C_(9) LR_C_(10) LR_C_(11) LR_C_(12) LR_C_(13) LR_C_(14) LR_C_(15)
                 : [a] "+v"(a) : [va] "v"(va), [one] "v"(one));
#undef LR_C_
}
/* two chains side by side (each one's next step waits for its last: the other fills the wait) */
__device__ __forceinline__ void lr_chain16x2(double &a, double va, double &b, double vb)
{
    const double one = 1.0;
#define LR_C_(J) LR_FD_("%[a]", "%[va]", J) LR_FD_("%[b]", "%[vb]", J)
    asm volatile("s_nop 1\n\t" LR_C_(0) LR_C_(1) LR_C_(2) LR_C_(3) LR_C_(4) LR_C_(5) LR_C_(6) LR_C_(7) LR_C_(8) LR_C_(9) LR_C_(10) LR_C_(11) LR_C_(12) LR_C_(13) LR_C_(14) LR_C_(15)
                 : [a] "+v"(a), [b] "+v"(b) : [va] "v"(va), [vb] "v"(vb), [one] "v"(one));
#undef LR_C_
}
#endif
template <bool PROD>
__device__ __forceinline__ void lr_seq2(int nterms, double &ra, double &rb, const double (&ta)[LR_E], const double (&tb)[LR_E], lr_xbuf &XB, int &xpar)
{
    const unsigned tid = threadIdx.x;
    double a = ra, b = rb;
#pragma unroll
    for (int e = 0; e < LR_E; ++e) {
        const int m = nterms - e * LB_T;                 /* terms in this block (uniform) */
        if (m > 0) {
            const int mm = m < LB_T ? m : LB_T;
            XB.a[xpar][tid] = ta[e]; XB.b[xpar][tid] = tb[e];
            __syncthreads();
#ifdef NLA_SIMT_EMU
            for (int i = 0; i < mm; ++i) { a += XB.a[xpar][i]; if (PROD) b *= XB.b[xpar][i]; else b += XB.b[xpar][i]; }
#else
            if (!PROD && mm == LB_T) {
                /* a whole block, two sums: sixteen LDS reads (the block's terms, sixteen at a time, the same in all four rows of the wavefront) in flight, then the chains */
                double va[LB_T / 16], vb[LB_T / 16];
#pragma unroll
                for (int c = 0; c < LB_T / 16; ++c) { va[c] = XB.a[xpar][16 * c + (tid & 15)]; vb[c] = XB.b[xpar][16 * c + (tid & 15)]; }
#pragma unroll
                for (int c = 0; c < LB_T / 16; ++c) lr_chain16x2(a, va[c], b, vb[c]);
            } else
            for (int c = 0; c < mm; c += 64) {
                const double va = XB.a[xpar][c + (tid & 63)], vb = XB.b[xpar][c + (tid & 63)];
                if (mm - c >= 64) {
#pragma unroll
                    for (int l = 0; l < 64; ++l) { a += lr_lane(va, l); if (PROD) b *= lr_lane(vb, l); else b += lr_lane(vb, l); }
                } else for (int l = 0; l < mm - c; ++l) { a += lr_lane(va, l); if (PROD) b *= lr_lane(vb, l); else b += lr_lane(vb, l); }
            }
#endif
            xpar ^= 1;
        }
    }
    ra = a; rb = b;
}
template <bool PROD> __device__ __forceinline__ void lr_seq2(int, double &, double &, const double (&)[LR_E], const double (&)[LR_E], lr_nobuf &, int &) { }
/* one sum (the recurrences' dot products: the chain's three instructions per term are what a column costs in this mode) */
__device__ __forceinline__ double lr_seq1(int nterms, const double (&ta)[LR_E], lr_xbuf &XB, int &xpar)
{
    const unsigned tid = threadIdx.x;
    double a = 0.;
#pragma unroll
    for (int e = 0; e < LR_E; ++e) {
        const int m = nterms - e * LB_T;
        if (m > 0) {
            const int mm = m < LB_T ? m : LB_T;
            XB.a[xpar][tid] = ta[e];
            __syncthreads();
#ifdef NLA_SIMT_EMU
            for (int i = 0; i < mm; ++i) a += XB.a[xpar][i];
#else
            if (mm == LB_T) {
                double va[LB_T / 16];
#pragma unroll
                for (int c = 0; c < LB_T / 16; ++c) va[c] = XB.a[xpar][16 * c + (tid & 15)];
#pragma unroll
                for (int c = 0; c < LB_T / 16; ++c) lr_chain16(a, va[c]);
            } else
            for (int c = 0; c < mm; c += 64) {
                const double va = XB.a[xpar][c + (tid & 63)];
                if (mm - c >= 64) {
#pragma unroll
                    for (int l = 0; l < 64; ++l) a += lr_lane(va, l);
                } else for (int l = 0; l < mm - c; ++l) a += lr_lane(va, l);
            }
#endif
            xpar ^= 1;
        }
    }
    return a;
}
__device__ __forceinline__ double lr_seq1(int, const double (&)[LR_E], lr_nobuf &, int &) { return 0.; }

/* sin and cos of one argument at once: Ackley and Rastrigin need cos(2 pi x) for f and sin(2 pi x) for the gradient of the SAME
 * coordinates — one argument reduction and one pair of polynomials instead of two (the device library's sincos returns exactly
 * the two values its sin and cos return: tests/test_gpu_lbfgs.py::test_device_sincos_is_sin_and_cos; glibc's does NOT
 * (../objfuncs.h), so the CPU emulation calls them separately) */
__device__ __forceinline__ void lr_sincos(double a, double *sn, double *cs)
{
#ifdef NLA_SIMT_EMU
    *sn = nla_libm_sin(a); *cs = nla_libm_cos(a);      /* (behind the wrappers gcc cannot merge the two into glibc's sincos) */
#else
    sincos(a, sn, cs);
#endif
}

/* objective and gradient of the point in LDS (x -> g), the formulas and the summation tree of lb_objgrad (local_common.h) */
template <int OBJ, bool EXACT, class XBuf>
__device__ __forceinline__ double lr_objgrad(int n, const double *x, double *g, lr_red &R, int &par, double sign, XBuf &XB, int &xpar)
{
    const int tid = threadIdx.x;
    constexpr bool FUSED = OBJ == NLA_OBJ_RASTRIGIN || OBJ == NLA_OBJ_ACKLEY;
    double sn[FUSED ? LR_E : 1];
    nla_obj_part t;
    if constexpr (EXACT) {
        /* f in the host callback's summation order (../objfuncs.h nla_obj_eval_seq; local_common.h lb_obj_exact): one accumulator over i
         * ascending with the accumulator's start value as there; t = the sums the gradient formulas need */
        constexpr bool PROD = OBJ == NLA_OBJ_GRIEWANK;
        double ta[LR_E], tb[LR_E], ia = 0., ib = PROD ? 1. : 0.;
        int nt = n;
        if (OBJ == NLA_OBJ_RASTRIGIN) ia = 10.0 * n;
        if (OBJ == NLA_OBJ_GRIEWANK) ia = 1.;
        if (OBJ == NLA_OBJ_ROSENBROCK || OBJ == NLA_OBJ_LEVY) nt = n - 1;
        if (OBJ == NLA_OBJ_LEVY) ia = nla_levy_head(x[0], x[n - 1]);
#pragma unroll
        for (int e = 0; e < LR_E; ++e) {
            const int i = tid + e * LB_T;
            ta[e] = 0.; tb[e] = PROD ? 1. : 0.; sn[FUSED ? e : 0] = 0.;
            if (i < nt) {
                const double xv = x[i];
                if (FUSED) {
                    double cs;
                    lr_sincos(NLA_PI2 * xv, &sn[FUSED ? e : 0], &cs);
                    if (OBJ == NLA_OBJ_RASTRIGIN) ta[e] = xv * xv - 10.0 * cs;
                    else { ta[e] = nla_sqr(xv); tb[e] = cs; }
                } else if (OBJ == NLA_OBJ_GRIEWANK) { ta[e] = nla_griewank_sum_term(xv); tb[e] = nla_griewank_prod_term(xv, (unsigned) i); }
                else if (OBJ == NLA_OBJ_ROSENBROCK) ta[e] = nla_rosenbrock_term(xv, x[i + 1]);
                else if (OBJ == NLA_OBJ_LEVY) ta[e] = nla_levy_term(xv, x[i + 1]);
                else ta[e] = nla_sqr(xv);
            }
        }
        lr_seq2<PROD>(nt, ia, ib, ta, tb, XB, xpar);
        t.a = ia; t.b = (OBJ == NLA_OBJ_ACKLEY || PROD) ? ib : 0.;
    } else if (FUSED) {                   /* nla_obj_partial's sums (thread-strided, coordinate order) with the sines kept for the gradient */
        t.a = 0; t.b = 0;
#pragma unroll
        for (int e = 0; e < LR_E; ++e) {
            const int i = tid + e * LB_T;
            sn[FUSED ? e : 0] = 0.;
            if (i < n) {
                const double xv = x[i];
                double cs;
                lr_sincos(NLA_PI2 * xv, &sn[FUSED ? e : 0], &cs);
                if (OBJ == NLA_OBJ_RASTRIGIN) t.a += xv * xv - 10.0 * cs;
                else { t.a += nla_sqr(xv); t.b += cs; }
            }
        }
    } else t = nla_obj_partial<OBJ>(n, tid, LB_T, [&](int i) { return x[i]; });
    double f;
    if constexpr (EXACT) {
        f = OBJ == NLA_OBJ_ACKLEY ? nla_ackley_finish(t.a, t.b, (unsigned) n) : (OBJ == NLA_OBJ_GRIEWANK ? t.a - t.b : t.a);
    } else {
        t = nla_obj_wave_reduce<OBJ>(t);
        if ((tid & 63) == 0) { R.v[par][tid >> 6][0] = t.a; R.v[par][tid >> 6][1] = t.b; }
        __syncthreads();
        t.a = R.v[par][0][0]; t.b = R.v[par][0][1];
#pragma unroll
        for (int w = 1; w < LB_W; ++w) { nla_obj_part o; o.a = R.v[par][w][0]; o.b = R.v[par][w][1]; t = nla_obj_combine<OBJ>(t, o); }
        par ^= 1;
        f = nla_obj_finish<OBJ>(n, t, [&](int i) { return x[i]; });
    }
    if (OBJ == NLA_OBJ_RASTRIGIN) {
#pragma unroll
        for (int e = 0; e < LR_E; ++e) { const int i = tid + e * LB_T; if (i < n) g[i] = 2 * x[i] + 10.0 * NLA_PI2 * sn[FUSED ? e : 0]; }
    } else if (OBJ == NLA_OBJ_ACKLEY) {
        const double r = sqrt(t.a / (unsigned) n), e1 = exp(-0.2 * r), e2 = exp(t.b / (unsigned) n);
#pragma unroll
        for (int e = 0; e < LR_E; ++e) {
            const int i = tid + e * LB_T;
            if (i < n) {
                double gi = e2 * NLA_PI2 * sn[FUSED ? e : 0] / (unsigned) n;
                if (r > 0) gi += 4.0 * e1 * x[i] / ((unsigned) n * r);
                g[i] = gi;
            }
        }
    } else if (OBJ == NLA_OBJ_GRIEWANK) {
        _Pragma("unroll 1") for (int i = tid; i < n; i += LB_T) {
            const double sq = sqrt(i + 1.);
            g[i] = x[i] * 0.0005 + t.b * tan(x[i] / sq) / sq;
        }
    } else if (OBJ == NLA_OBJ_ROSENBROCK) {
        _Pragma("unroll 1") for (int i = tid; i < n; i += LB_T) {
            double gi = 0;
            if (i > 0) { const double a = x[i] - x[i - 1] * x[i - 1]; gi = 200 * a; }
            if (i + 1 < n) { const double a = x[i + 1] - x[i] * x[i], b = 1 - x[i]; gi += -400 * a * x[i] - 2 * b; }
            g[i] = gi;
        }
    } else if (OBJ == NLA_OBJ_LEVY) {
        _Pragma("unroll 1") for (int i = tid; i < n; i += LB_T) {
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
        _Pragma("unroll 1") for (int i = tid; i < n; i += LB_T) g[i] = 2 * x[i];
    }
    if (sign < 0) {
        _Pragma("unroll 1") for (int i = tid; i < n; i += LB_T) g[i] = -g[i];
        f = -f;
    }
    __syncthreads();            /* every thread has read its neighbours' x: the point may change again */
    return f;
}

}
#ifndef LR_WIDE
/* development / test aid: sin, cos and sincos of the device library for n arguments (tests/test_gpu_lbfgs.py) */
__global__ void lr_debug_sincos_kernel(int n, const double *a, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s2, c2;
    sincos(a[i], &s2, &c2);
    out[4 * (size_t) i] = sin(a[i]); out[4 * (size_t) i + 1] = cos(a[i]); out[4 * (size_t) i + 2] = s2; out[4 * (size_t) i + 3] = c2;
}
/* development / test aid: for every lane of 4 wavefronts and every distance of the butterfly, the partner's value by __shfl_xor and by
 * nla_xor_lane (dev_common.h) — out[(2 s + which) * 256 + thread], s = 0 .. 5 for M = 32 .. 1; the int version behind them (12 * 256 on) */
__global__ void lr_debug_xor_lane_kernel(const double *in, double *out)
{
    const int t = threadIdx.x;
    const double v = in[t];
    const int iv = (int) (long long) (in[t] * 1e6);
    int s = 0;
#define LR_D_(M) { out[(2 * s) * 256 + t] = __shfl_xor(v, M, 64); out[(2 * s + 1) * 256 + t] = nla_xor_lane<M>(v); \
                   out[(12 + 2 * s) * 256 + t] = (double) __shfl_xor(iv, M, 64); out[(12 + 2 * s + 1) * 256 + t] = (double) nla_xor_lane<M>(iv); ++s; }
    NLA_BUTTERFLY(LR_D_);
#undef LR_D_
}
extern "C" int nla_k_debug_xor_lane(const double *in, double *out, void *stream)
{
    hipLaunchKernelGGL(lr_debug_xor_lane_kernel, dim3(1), dim3(256), 0, (hipStream_t) stream, in, out);
    NLA_LAUNCH_CHECK();
    return 0;
}
extern "C" int nla_k_debug_sincos(int n, const double *a, double *out, void *stream)
{
    if (n <= 0) return 0;
    hipLaunchKernelGGL(lr_debug_sincos_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream, n, a, out);
    NLA_LAUNCH_CHECK();
    return 0;
}
#endif
namespace {

/* thread 0, when PS1L01 reports the line search finished (q.isys == 0): take its results over (plis.c:395-403) */
__device__ __forceinline__ int lr_line_search_finished(lr_ctl &C)
{
    C.fval = C.q.f; C.p = C.q.p; C.kd = C.q.kd; C.nred = C.q.nred; C.maxst = C.q.maxst; C.c.iters = C.q.iters;
    if (C.c.iters <= 0) { C.fval = C.fo; C.p = C.po; C.c.irest = LB_MAX(C.c.irest, 1); return LR_RESTORE; }      /* zero step: restore and restart */
    return LR_PYTRCD;
}

#define LR_FOR(e) _Pragma("unroll") for (int e = 0; e < LR_E; ++e)
#define LR_I(e) (tid + (unsigned) (e) * LB_T)          /* coordinate e of this thread */

/* Global memory goes through BUFFER instructions: one descriptor (4 SGPRs: base, size) per vector, the thread's byte offset
 * tid * 8 in ONE VGPR for every access of the kernel, the coordinate's 2 KB stride in the scalar offset.  The plain-pointer form
 * made the compiler keep a 64-bit address per (vector, coordinate) alive across the whole search loop — ~130 VGPRs of hoisted
 * addresses, which it then spilled (round 4, first version of this kernel: 256 VGPRs + 500 B of scratch per lane).  The descriptor's
 * size is the vector's n * 8 bytes: a coordinate >= n loads 0.0 and its store is dropped by the hardware's bounds check, so
 * the loops need no `i < n` branches (coordinates >= n carry x = g = 0 and the bound type "fixed" in LDS: they add +0.0). */
#ifdef NLA_SIMT_EMU
struct lr_buf { char *base; unsigned bytes; };
static inline lr_buf lr_make_buf(const void *p, unsigned bytes) { lr_buf b = { (char *) p, bytes }; return b; }
static inline double lr_bload(lr_buf b, unsigned voff, unsigned soff) { double v = 0.; if (voff + soff + 8 <= b.bytes) memcpy(&v, b.base + voff + soff, 8); return v; }
static inline void lr_bstore(double v, lr_buf b, unsigned voff, unsigned soff) { if (voff + soff + 8 <= b.bytes) memcpy(b.base + voff + soff, &v, 8); }
#define LR_UNIFORM(x) (x)
#define LR_SCHED_FENCE() do { } while (0)
#define LR_WAVE_ALL(p) simt_wave_all(p)
#else
typedef __amdgpu_buffer_rsrc_t lr_buf;
typedef unsigned lr_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lr_buf lr_make_buf(const void *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *) p, 0, bytes, 0x00020000); }
__device__ __forceinline__ double lr_bload(lr_buf b, unsigned voff, unsigned soff) { return __builtin_bit_cast(double, (lr_v2u) __builtin_amdgcn_raw_buffer_load_b64(b, voff, soff, 0)); }
__device__ __forceinline__ void lr_bstore(double v, lr_buf b, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lr_v2u, v), b, voff, soff, 0); }
/* the instruction scheduler may not move anything across this point: a refill of a column array must stay BEHIND the last use of
 * the values it replaces, or the compiler needs a second set of registers for the array (and copies it every iteration) */
#define LR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LR_WAVE_ALL(p) (__all(p) != 0)
#define LR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)          /* a workgroup-uniform value read from LDS: tell the compiler (scalar branches, scalar address arithmetic) */
#endif
#define LR_LD(buf, e) lr_bload(buf, voff, (unsigned) (e) * (LB_T * 8u))
#define LR_ST(v, buf, e) lr_bstore(v, buf, voff, (unsigned) (e) * (LB_T * 8u))

template <int OBJ, bool EXACT>
__global__ __launch_bounds__(LB_T) __attribute__((amdgpu_waves_per_eu(LR_PER_EU, LR_PER_EU))) void LR_KERNEL(
    int n, int ld, int mf, int count, const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ X,
    double *__restrict__ work, double *__restrict__ hist, nla_lbfgs_params P, nla_lbfgs_result *__restrict__ out)
{
    /* one LDS block with the scalar state FIRST: its fields then sit at small constant addresses (ds instructions carry a 16-bit
     * offset); behind the 64 KB of vectors every field needed an address register of its own, which the compiler kept alive
     * across the whole search and spilled */
    __shared__ struct { lr_ctl C; lr_red R; typename std::conditional<EXACT, lr_xbuf, lr_nobuf>::type XB; signed char six[LR_NMAX]; double sx[LR_NMAX], sg[LR_NMAX]; } L;
    lr_ctl &C = L.C;
    lr_red &R = L.R;
    signed char *const six = L.six;
    double *const sx = L.sx, *const sg = L.sg;
    const int inst = blockIdx.x;
    const unsigned tid = threadIdx.x, voff = tid * 8u, nbytes = (unsigned) n * 8u;
    if (inst >= count) return;
    double *hx = hist + (size_t) inst * 2 * (size_t) mf * ld, *hg = hx + (size_t) mf * ld;
    double *ucol = work + (size_t) count * 4 * ld + (size_t) inst * 2 * mf, *vcol = ucol + mf;
    /* PLIS's own copies of the bounds (xl / xu, plis.c:232-241,463-469) differ from lb / ub only on coordinates of type 5 (fixed:
     * lb == ub, or an empty interval), and no formula below reads a type-5 coordinate's bounds — so the box is read from lb / ub
     * themselves: ONE pair of vectors for all searches of the batch (cache-resident) instead of a copy per search */
    const lr_buf bxl = lr_make_buf(lb, nbytes), bxu = lr_make_buf(ub, nbytes);
    const double eta9 = 1e120, eps8 = 1., eps9 = 1e-8, alf1 = 1e-10, alf2 = 1e10, told = 1e-4, xmax = 1e16, maxf = 1e20, minf_est = -HUGE_VAL;
    int par = 0, xpar = 0, go;
    (void) xpar;
    double sr[LR_E];                    /* the search direction, this thread's coordinates */
    PROF_DECL;
    /* column "i-th newest" of the ring (the reference shifts all columns every iteration, mxdrsu, mssubs.c:503-524); head < mf, i <= mf */
#define COLIDX(h, i) ((h) + (i) - 1 < mf ? (h) + (i) - 1 : (h) + (i) - 1 - mf)
#define COLX(h, i) lr_make_buf(hx + (size_t) COLIDX(h, i) * ld, nbytes)
#define COLG(h, i) lr_make_buf(hg + (size_t) COLIDX(h, i) * ld, nbytes)
#define COLU(h, i) (ucol[COLIDX(h, i)])
    /* pcbs04 (DO_PROJECT != 0) then pyadc0 (DO_PROJECT < 2) on this thread's coordinates (both touch one coordinate at a time: no
     * barrier between them) */
#define LR_PROJECT_AND_ACTIVATE(DO_PROJECT) do {                                                                             \
        double l_[LR_E], u_[LR_E];                                                                                          \
        LR_FOR(e) { l_[e] = LR_LD(bxl, e); u_[e] = LR_LD(bxu, e); }                                                         \
        LR_FOR(e) {                                                                                                         \
            const unsigned i = LR_I(e);                                                                                     \
            int ii = six[i], t = ii < 0 ? -ii : ii;                                                                         \
            double v = sx[i];                                                                                               \
            if (DO_PROJECT) {                                                                                               \
                if ((t == 1 || t == 3 || t == 4) && v <= l_[e] + eps9 * LB_MAX(fabs(l_[e]), 1.)) v = l_[e];                \
                if ((t == 2 || t == 3 || t == 4) && v >= u_[e] - eps9 * LB_MAX(fabs(u_[e]), 1.)) v = u_[e];                \
            }                                                                                                               \
            if (DO_PROJECT < 2) {                                                                                           \
                if (t >= 5) ii = -t;                                                                                        \
                else if ((t == 1 || t == 3 || t == 4) && v <= l_[e]) { v = l_[e]; ii = (t == 4) ? -3 : -t; }                \
                else if ((t == 2 || t == 3 || t == 4) && v >= u_[e]) { v = u_[e]; ii = (t == 3) ? -4 : -t; }                \
                six[i] = (signed char) ii;                                                                                  \
            }                                                                                                               \
            sx[i] = v;                                                                                                      \
        }                                                                                                                   \
    } while (0)

    /* a sum (two sums) over this thread's coordinates e where MASK holds: the workgroup tree, or the reference's order (EXACT) */
#define LR_SUM2(ra, rb, MASK, TA, TB) do {                                                                                  \
        ra = 0.; rb = 0.;                                                                                                   \
        if constexpr (EXACT) {                                                                                              \
            double ta_[LR_E], tb_[LR_E];                                                                                    \
            LR_FOR(e) { ta_[e] = 0.; tb_[e] = 0.; if (MASK) { ta_[e] = (TA); tb_[e] = (TB); } }                             \
            lr_seq2<false>(n, ra, rb, ta_, tb_, L.XB, xpar);                                                                \
        } else { LR_FOR(e) if (MASK) { ra += (TA); rb += (TB); } lr_reduce2<false>(ra, rb, R, par); }                        \
    } while (0)
#define LR_SUM1(ra, MASK, TA) do {                                                                                          \
        ra = 0.;                                                                                                            \
        if constexpr (EXACT) {                                                                                              \
            double ta_[LR_E];                                                                                               \
            LR_FOR(e) { ta_[e] = 0.; if (MASK) ta_[e] = (TA); }                                                             \
            ra = lr_seq1(n, ta_, L.XB, xpar);                                                                               \
        } else { LR_FOR(e) if (MASK) ra += (TA); ra = lr_reduce1<false>(ra, R, par); }                                       \
    } while (0)

    {                                                                        /* plis.c:463-469, 232-241 */
        const lr_buf bX = lr_make_buf(X + (size_t) inst * ld, nbytes);
        const lr_buf bx1 = lr_make_buf(hx, nbytes), bg1 = lr_make_buf(hg, nbytes);
        double l0[LR_E], u0[LR_E], x0[LR_E];
        LR_FOR(e) { l0[e] = LR_LD(bxl, e); u0[e] = LR_LD(bxu, e); x0[e] = LR_LD(bX, e); }
        LR_FOR(e) {
            const unsigned i = LR_I(e);
            const int lbu = l0[e] <= -0.99 * HUGE_VAL, ubu = u0[e] >= 0.99 * HUGE_VAL;
            int t = lbu ? (ubu ? 0 : 2) : (ubu ? 1 : (l0[e] == u0[e] ? 5 : 3));
            if ((t == 3 || t == 4) && u0[e] <= l0[e]) t = 5;               /* (xl = xu = lb there; type 5: never read) */
            if (i >= (unsigned) n) { t = -5; x0[e] = 0.; }                  /* no such coordinate: x = g = 0, "fixed" — every loop below passes over it */
            six[i] = (signed char) t; sx[i] = x0[e]; sg[i] = 0.; sr[e] = 0.;
            LR_ST(0., bx1, e); LR_ST(0., bg1, e);      /* the reference zero-fills xo (plis.c:475); column 1 is read before it is written */
        }
    }
    if (tid == 0) {
        memset(&C, 0, sizeof C);
        C.xtol_rel = P.xtol_rel <= 0. ? 1e-16 : P.xtol_rel;                  /* plis.c:202-214 */
        C.tolg = P.tolg <= 0. ? 1e-8 : P.tolg;
        C.ls.minf_max = P.minf_max; C.ls.ftol_rel = P.ftol_rel <= 0. ? 1e-14 : P.ftol_rel; C.ls.ftol_abs = P.ftol_abs; C.ls.maxeval = P.maxeval;
        C.fo = minf_est; C.rmax = eta9; C.kd = 1;
        C.c.ites = 1; C.c.mtesx = 2; C.c.mtesf = 2; C.c.iters = 2; C.c.ires1 = 999; C.c.ires2 = 0; C.c.kd = 1;
        C.c.mit = INT_MAX; C.c.mfg = P.maxeval > 0 ? P.maxeval : INT_MAX;
        C.c.kit = -(C.c.ires1 * n + C.c.ires2);
    }
    LR_PROJECT_AND_ACTIVATE(1);
    __syncthreads();
    PROF(0);
    {
        const double f = lr_objgrad<OBJ, EXACT>(n, sx, sg, R, par, P.sign, L.XB, xpar);
        if (tid == 0) {
            C.fval = f;
            if (P.ftrace && C.nevals < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + C.nevals] = f;
            ++C.nevals; ++C.c.nfg;
            if (P.abort && *(const volatile int32_t *) P.abort == 100) { C.tmo = 1; C.c.iterm = 100; }     /* plis.c:263 */
        }
    }
    PROF(6);

    for (;;) {
        PROF_ITER;
        /* pytrcg: largest free gradient component, largest wrong-signed multiplier on an active bound */
        {
            double gm = 0, um = 0;
            LR_FOR(e) {
                const double t = sg[LR_I(e)];
                const int ii = six[LR_I(e)];
                if (ii >= 0) gm = LB_MAX(gm, fabs(t));
                else if (ii <= -5) { }
                else if (ii == -1 || ii == -3) { if (-t > um) um = -t; }
                else if (ii == -2 || ii == -4) { if (t > um) um = t; }
            }
            lr_reduce2<true>(gm, um, R, par);
            if (tid == 0) {
                int g_ = LR_CONTINUE;
                if (C.c.iterm == 100) g_ = LR_EXIT;                          /* the time limit hit during the first evaluation */
                else {
                    C.gmax = gm; C.umax = um; C.c.kd = C.kd;
                    /* the host's abort flag (pinned host memory: a PCIe round trip that the whole workgroup waits for) is looked at on
                     * every 4th iteration: a time limit or a forced stop is honoured at most 3 iterations (~0.2 ms) later than the
                     * reference's per-iteration test (plis.c:263,273,371, pssubs.c:914) would — it is an asynchronous event either way */
                    if (P.abort && (C.c.nit & 3) == 0) { const int ab = *(const volatile int32_t *) P.abort; C.forced = ab == -999; C.tmo = ab == 100; }
                    lb_pyfut1(n, C.fval, &C.fo, um, gm, C.xstop, &C.ls, C.forced, C.nevals, C.tolg, &C.c);
                    if (C.c.iterm != 0) g_ = LR_EXIT;
                    else if (C.tmo) { C.c.iterm = 100; g_ = LR_EXIT; }       /* plis.c:273 */
                    else if (C.rmax > 0. && um > eps8 * gm) g_ = LR_RELEASE;
                }
                C.go = g_;
            }
            __syncthreads();
            go = LR_UNIFORM(C.go);
        }
        if (go == LR_EXIT) break;
        if (go == LR_RELEASE) {                                              /* pyrmc0: release wrong-signed active bounds */
            int rel = 0;
            LR_FOR(e) {
                const int t = six[LR_I(e)];
                if (t >= 0 || t <= -5) continue;
                if ((t == -1 || t == -3) && -sg[LR_I(e)] <= 0.) continue;
                if ((t == -2 || t == -4) && sg[LR_I(e)] <= 0.) continue;
                ++rel;
                six[LR_I(e)] = (signed char) LB_MIN(-t, 3);
            }
            rel = lr_reduce_isum(rel, R, par);
            if (tid == 0 && rel > 1) C.c.irest = LB_MAX(C.c.irest, 1);       /* read again by thread 0 only, below */
        }
        PROF(1);
    direction:
        {
            /* |g|^2 and x1.g1 (column 1 = the newest pair) over the free coordinates, one reduction */
            const int head = LR_UNIFORM(C.head);
            double cx[LR_E], cg[LR_E], gg = 0, bb = 0;
            {
                const lr_buf px = COLX(head, 1), pg = COLG(head, 1);
                LR_FOR(e) { cx[e] = LR_LD(px, e); cg[e] = LR_LD(pg, e); }
            }
            LR_SUM2(gg, bb, six[LR_I(e)] >= 0, sg[LR_I(e)] * sg[LR_I(e)], cx[e] * cg[e]);
            if (tid == 0) {
                int g_ = LR_STEEPEST;
                C.gnorm = sqrt(gg);
                if (C.c.irest == 0) {
                    const int k = LB_MIN(C.c.nit - C.c.kit, mf);
                    if (k <= 0) C.c.irest = LB_MAX(C.c.irest, 1);
                    else if (bb <= 0.) C.c.irest = LB_MAX(C.c.irest, 1);
                    else { COLU(head, 1) = 1. / bb; C.cols += k; C.k = k; C.b = bb; g_ = LR_STRANG; }
                }
                if (g_ == LR_STEEPEST) {                                      /* steepest descent */
                    C.snorm = C.gnorm;
                    if (C.c.kit < C.c.nit) C.c.kit = C.c.nit;
                    else { C.c.iterm = -10; if (C.c.iters < 0) C.c.iterm = C.c.iters - 5; }
                }
                C.go = g_;
            }
            __syncthreads();
            go = LR_UNIFORM(C.go);
        }
        PROF(2);
        {
            double ssq = 0, pp = 0;
            unsigned live = 0, nvalid = 0;               /* bit e: coordinate e of this thread is free / exists */
            LR_FOR(e) { sr[e] = 0.; if (LR_I(e) < (unsigned) n) nvalid |= 1u << e; if (six[LR_I(e)] >= 0) { live |= 1u << e; sr[e] = -sg[LR_I(e)]; } }       /* mxuneg */
            if (go == LR_STRANG) {
                /* the two Strang loops (mxdrcb / mxdrcf, mssubs.c:353-441): the next column is on its way while this one's dot
                 * product is reduced; one barrier per column.  (A column's values on coordinates that are not free are loaded but
                 * never used: `live` masks every sum and update, as mxudot / mxudir's ix test does.) */
                const int head = LR_UNIFORM(C.head), k = LR_UNIFORM(C.k);
                const bool all_free = LR_WAVE_ALL(live == nvalid);          /* wave-uniform: no coordinate of this wavefront sits on a bound */
#define LR_LOAD(arr, buf_) do { const lr_buf q_ = buf_; LR_FOR(e) arr[e] = LR_LD(q_, e); } while (0)
                if (!EXACT && all_free) {
                /* The usual case, without masks: a coordinate that does not exist (>= n) has sr = 0 and its column entries load as
                 * 0.0 (bounds-checked buffer), so it adds +0.0 and stays 0 — the sums are the masked ones bit for bit.
                 * Software pipeline, two columns deep.  A column's x-part is dead once its dot product has been formed, its g-part
                 * once the axpy is done: each array is refilled at that point with the column TWO steps ahead, which then has two
                 * whole steps (two reductions, four vector passes) to arrive — the history comes from HBM / the memory-side cache
                 * (~2 us away with 300 searches streaming; one step is ~0.6 us of work).  Two sets of arrays (a0/b0 for odd steps,
                 * a1/b1 for even ones), the loop unrolled by two so that no array is ever copied.  The refills are UNCONDITIONAL
                 * (past the end they fetch the last column once more): a conditional refill makes "old or new" a second register
                 * array with a copy per step; the scheduling fences keep a refill behind the last use of what it replaces. */
                double a0[LR_E], b0[LR_E], a1[LR_E], b1[LR_E], u0, u1, un;
#define LR_BACK_STEP(j_, A, B, U) do {                              /* mxdrcb, column j_ */                                  \
        double t_ = 0;                                                                                                      \
        LR_FOR(e) t_ += sr[e] * A[e];                                                                                       \
        const int jn_ = (j_) + 2 <= k ? (j_) + 2 : k;                                                                       \
        LR_SCHED_FENCE();                                                                                                   \
        LR_LOAD(A, COLX(head, jn_)); un = COLU(head, jn_);                                                                  \
        const double v_ = U * lr_reduce1<false>(t_, R, par);                                                                \
        if (tid == 0) vcol[(j_) - 1] = v_;                                                                                  \
        LR_FOR(e) sr[e] = sr[e] + (-v_) * B[e];                                                                             \
        LR_SCHED_FENCE();                                                                                                   \
        LR_LOAD(B, COLG(head, jn_));                                                                                        \
        U = un;                                                                                                             \
    } while (0)
#define LR_FWD_STEP(j_, A, B, U, V) do {                            /* mxdrcf, column j_: A = its g-part, B = its x-part */  \
        double t_ = 0;                                                                                                      \
        LR_FOR(e) t_ += sr[e] * A[e];                                                                                       \
        const int jp_ = (j_) - 2 >= 1 ? (j_) - 2 : 1;                                                                       \
        LR_SCHED_FENCE();                                                                                                   \
        LR_LOAD(A, COLG(head, jp_)); un = COLU(head, jp_); const double vn_ = vcol[jp_ - 1];                                \
        const double tt_ = U * lr_reduce1<false>(t_, R, par);                                                               \
        const double w_ = V - tt_;                                                                                          \
        LR_FOR(e) sr[e] = sr[e] + w_ * B[e];                                                                                \
        LR_SCHED_FENCE();                                                                                                   \
        LR_LOAD(B, COLX(head, jp_));                                                                                        \
        U = un; V = vn_;                                                                                                    \
    } while (0)
                {
                    const int j2 = k >= 2 ? 2 : 1;
                    LR_LOAD(a0, COLX(head, 1)); LR_LOAD(b0, COLG(head, 1)); u0 = COLU(head, 1);
                    LR_LOAD(a1, COLX(head, j2)); LR_LOAD(b1, COLG(head, j2)); u1 = COLU(head, j2);
                }
                for (int j = 1; j <= k; j += 2) {
                    LR_BACK_STEP(j, a0, b0, u0);
                    if (j + 1 <= k) LR_BACK_STEP(j + 1, a1, b1, u1);
                }
                {
                    /* g of column 1 for the scaling; meanwhile the forward loop's first two columns (k, k - 1) are on their way */
                    const int k2 = k >= 2 ? k - 1 : 1;
                    LR_SCHED_FENCE();
                    LR_LOAD(b1, COLG(head, 1));
                    LR_LOAD(a0, COLG(head, k)); LR_LOAD(b0, COLX(head, k)); u0 = COLU(head, k);
                    LR_LOAD(a1, COLG(head, k2)); u1 = COLU(head, k2);
                    double t = 0;
                    LR_FOR(e) t += b1[e] * b1[e];
                    LR_SCHED_FENCE();
                    LR_LOAD(b1, COLX(head, k2));
                    const double a = lr_reduce1<false>(t, R, par);
                    if (a > 0.) { const double sc = C.b / a; LR_FOR(e) sr[e] = sr[e] * sc; }
                    double v0 = vcol[k - 1], v1 = vcol[k2 - 1];
                    for (int j = k; j >= 1; j -= 2) {
                        LR_FWD_STEP(j, a0, b0, u0, v0);
                        if (j - 1 >= 1) LR_FWD_STEP(j - 1, a1, b1, u1, v1);
                    }
                }
                } else {
                /* some coordinate of this wavefront is on a bound: every sum and update masked by `live`, as mxudot / mxudir's ix test
                 * does (a column's values on such coordinates are loaded but never used); the pipeline one column deep — the
                 * sixteen mask tests take the registers the second pair of arrays would need */
                double ca[LR_E], cb[LR_E], u1, un;
                LR_LOAD(ca, COLX(head, 1));
                LR_LOAD(cb, COLG(head, 1));
                u1 = COLU(head, 1);
                for (int j = 1; j <= k; ++j) {                       /* mxdrcb */
                    double t;
                    const int jn = j < k ? j + 1 : j;
                    if constexpr (EXACT) LR_SUM1(t, live & (1u << e), sr[e] * ca[e]);
                    else { t = 0; LR_FOR(e) if (live & (1u << e)) t += sr[e] * ca[e]; }
                    LR_SCHED_FENCE();
                    LR_LOAD(ca, COLX(head, jn)); un = COLU(head, jn);
                    if constexpr (!EXACT) t = lr_reduce1<false>(t, R, par);
                    const double v = u1 * t;
                    if (tid == 0) vcol[j - 1] = v;
                    LR_FOR(e) if (live & (1u << e)) sr[e] = sr[e] + (-v) * cb[e];
                    LR_SCHED_FENCE();
                    LR_LOAD(cb, COLG(head, jn));
                    u1 = un;
                }
                LR_LOAD(cb, COLG(head, 1));
                LR_LOAD(ca, COLG(head, k));
                {
                    double t;
                    if constexpr (EXACT) LR_SUM1(t, live & (1u << e), cb[e] * cb[e]);
                    else { t = 0; LR_FOR(e) if (live & (1u << e)) t += cb[e] * cb[e]; }
                    LR_SCHED_FENCE();
                    LR_LOAD(cb, COLX(head, k));
                    if constexpr (!EXACT) t = lr_reduce1<false>(t, R, par);
                    const double a = t;
                    if (a > 0.) { const double sc = C.b / a; LR_FOR(e) sr[e] = sr[e] * sc; }
                }
                u1 = COLU(head, k);
                double v1 = vcol[k - 1], vn;
                for (int j = k; j >= 1; --j) {                       /* mxdrcf */
                    double t;
                    const int jp = j > 1 ? j - 1 : j;
                    if constexpr (EXACT) LR_SUM1(t, live & (1u << e), sr[e] * ca[e]);
                    else { t = 0; LR_FOR(e) if (live & (1u << e)) t += sr[e] * ca[e]; }
                    LR_SCHED_FENCE();
                    LR_LOAD(ca, COLG(head, jp)); un = COLU(head, jp); vn = vcol[jp - 1];
                    if constexpr (!EXACT) t = lr_reduce1<false>(t, R, par);
                    const double tt = u1 * t;
                    const double w = v1 - tt;
                    LR_FOR(e) if (live & (1u << e)) sr[e] = sr[e] + w * cb[e];
                    LR_SCHED_FENCE();
                    LR_LOAD(cb, COLX(head, jp));
                    u1 = un; v1 = vn;
                }
                }
            }
            PROF(3);
            LR_SUM2(ssq, pp, live & (1u << e), sr[e] * sr[e], sg[LR_I(e)] * sr[e]);      /* (|s|^2 is only looked at after the recurrences) */
            if (tid == 0) {
                int g_ = LR_CONTINUE;
                if (go == LR_STRANG) { C.snorm = sqrt(ssq); C.head = C.head > 0 ? C.head - 1 : mf - 1; }    /* mxdrsu: every column one older */
                if (C.kd > 0) C.p = pp;
                if (C.snorm <= 0.) C.c.irest = LB_MAX(C.c.irest, 1);
                else if (C.p + told * C.gnorm * C.snorm <= 0.) C.c.irest = 0;
                else C.c.irest = LB_MAX(C.c.irest, 1);
                if (C.c.irest == 0) {
                    C.nred = 0;
                    C.rmin = alf1 * C.gnorm / C.snorm;
                    C.rmax = LB_MIN(alf2 * C.gnorm / C.snorm, xmax / C.snorm);
                }
                if (C.c.iterm != 0) g_ = LR_EXIT;
                else if (C.tmo) { C.c.iterm = 100; g_ = LR_EXIT; }                               /* plis.c:371 */
                else if (C.c.irest != 0) g_ = LR_AGAIN;
                else { C.q.fp = C.fo; C.fo = C.fval; C.po = C.p; }
                C.go = g_;
            }
            __syncthreads();
            go = LR_UNIFORM(C.go);
        }
        if (go == LR_EXIT) break;
        if (go == LR_AGAIN) goto direction;
        /* pytrcs: save x, g in column 1; zero s on active bounds; largest step inside the box */
        {
            const int head = LR_UNIFORM(C.head);
            const lr_buf cx = COLX(head, 1), cg = COLG(head, 1);
            double rm = C.rmax, l_[LR_E], u_[LR_E];
            LR_FOR(e) { l_[e] = LR_LD(bxl, e); u_[e] = LR_LD(bxu, e); }
            LR_FOR(e) {
                const unsigned i = LR_I(e);
                const int ii = six[i];
                const double xv = sx[i];
                LR_ST(xv, cx, e); LR_ST(sg[i], cg, e);
                if (ii < 0) sr[e] = 0.;
                else {
                    if ((ii == 1 || ii >= 3) && sr[e] < -1. / eta9) rm = LB_MIN(rm, (l_[e] - xv) / sr[e]);
                    if ((ii == 2 || ii >= 3) && sr[e] > 1. / eta9) rm = LB_MIN(rm, (u_[e] - xv) / sr[e]);
                }
                if ((e & 3) == 3) LR_SCHED_FENCE();      /* four coordinates' divisions in flight, not all 32 (registers) */
            }
            rm = -lr_reduce1<true>(-rm, R, par);
            if (tid == 0) {
                int g_ = LR_NO_STEP;
                C.rmax = rm;
                if (rm != 0.) {
                    lb_ls_io *q = &C.q;
                    q->f = C.fval; q->fo = C.fo; q->p = C.p; q->po = C.po; q->minf_est = minf_est; q->maxf = maxf; q->rmin = C.rmin; q->rmax = rm;
                    q->tols = 1e-4; q->tolp = .8; q->kd = C.kd; q->ld = -1; q->nit = C.c.nit; q->kit = C.c.kit; q->nred = C.nred; q->mred = 10;
                    q->maxst = C.maxst; q->iest = 0; q->inits = 2; q->iters = C.c.iters; q->kters = 3; q->mes = 4; q->isys = 0;
                    lb_ps1l01(q, &C.lss);
                    g_ = q->isys == 0 ? lr_line_search_finished(C) : LR_LS_EVAL;
                }
                C.go = g_;
            }
            __syncthreads();
            go = LR_UNIFORM(C.go);
        }
        PROF(4);
        while (go == LR_LS_EVAL) {
            {
                const int head = LR_UNIFORM(C.head);
                const double r = C.q.r;
                const lr_buf xs = COLX(head, 1);
                double xo[LR_E];
                LR_FOR(e) xo[e] = LR_LD(xs, e);
                LR_FOR(e) if (six[LR_I(e)] >= 0) sx[LR_I(e)] = xo[e] + r * sr[e];
            }
            LR_PROJECT_AND_ACTIVATE(2);
            __syncthreads();
            PROF(5);
            const double f = lr_objgrad<OBJ, EXACT>(n, sx, sg, R, par, P.sign, L.XB, xpar);
            PROF(6);
            double pp;
            LR_SUM1(pp, six[LR_I(e)] >= 0, sg[LR_I(e)] * sr[e]);
            if (tid == 0) {
                int g_ = LR_LS_EVAL;
                C.q.f = f;
                if (P.ftrace && C.nevals < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + C.nevals] = f;
                ++C.nevals; ++C.c.nfg;
                C.q.p = pp;
                lb_ps1l01(&C.q, &C.lss);
                if (C.q.isys == 0) g_ = lr_line_search_finished(C);
                C.go = g_;
            }
            __syncthreads();
            go = LR_UNIFORM(C.go);
            PROF(7);
        }
        if (go == LR_RESTORE) {
            const int head = LR_UNIFORM(C.head);
            const lr_buf cx = COLX(head, 1), cg = COLG(head, 1);
            double a_[LR_E], b_[LR_E];
            LR_FOR(e) { a_[e] = LR_LD(cx, e); b_[e] = LR_LD(cg, e); }
            LR_FOR(e) { sx[LR_I(e)] = a_[e]; sg[LR_I(e)] = b_[e]; }
            __syncthreads();
            goto direction;
        }
        if (go == LR_PYTRCD) {
            /* pytrcd: column 1 := differences (zero on active coordinates); nlopt_stop_dx(x, dx) */
            const int head = LR_UNIFORM(C.head);
            const lr_buf dx = COLX(head, 1), dg = COLG(head, 1);
            double nx = 0, ndx = 0, a_[LR_E], b_[LR_E];
            LR_FOR(e) { a_[e] = LR_LD(dx, e); b_[e] = LR_LD(dg, e); }
            LR_FOR(e) {
                const unsigned i = LR_I(e);
                double ddx = sx[i] - a_[e], ddg = sg[i] - b_[e];
                if (six[i] < 0) { ddx = 0.; ddg = 0.; }
                LR_ST(ddx, dx, e); LR_ST(ddg, dg, e);
                a_[e] = ddx;
            }
            if (P.x_weights) {                                                /* nlopt_stop_dx's weighted norms (stop.c:37-57) */
                const lr_buf bw = lr_make_buf(P.x_weights, nbytes);
                double w_[LR_E];
                LR_FOR(e) w_[e] = LR_LD(bw, e);
                LR_SUM2(nx, ndx, true, w_[e] * fabs(sx[LR_I(e)]), w_[e] * fabs(a_[e]));
            } else LR_SUM2(nx, ndx, true, fabs(sx[LR_I(e)]), fabs(a_[e]));
            if (tid == 0) {
                C.po = C.q.r * C.po; C.p = C.q.r * C.p;
                C.xstop = ndx < C.xtol_rel * nx;                              /* nlopt_stop_dx, stop.c:110-120 */
                C.go = (!C.xstop && P.xtol_abs) ? LR_XTOL_ABS : LR_CONTINUE;
            }
            __syncthreads();
            if (LR_UNIFORM(C.go) == LR_XTOL_ABS) {
                const lr_buf bt = lr_make_buf(P.xtol_abs, nbytes);
                int viol = 0;
                LR_FOR(e) { const double ta = LR_LD(bt, e); if (LR_I(e) < (unsigned) n) viol += fabs(a_[e]) >= ta; }
                viol = lr_reduce_isum(viol, R, par);
                if (tid == 0) C.xstop = viol == 0;                            /* read by thread 0 only (pyfut1) */
            }
        }
        PROF(8);
        LR_FOR(e) if (six[LR_I(e)] < 0) six[LR_I(e)] = (signed char) -six[LR_I(e)];       /* mxvine */
        LR_PROJECT_AND_ACTIVATE(0);
        PROF(9);
    }
    {
        const lr_buf bX = lr_make_buf(X + (size_t) inst * ld, nbytes);
        LR_FOR(e) LR_ST(sx[LR_I(e)], bX, e);
    }
    PROF_STORE;
    if (tid == 0) {
        out[inst].f = C.fval; out[inst].ret = lb_result_of_iterm(C.c.iterm); out[inst].nevals = C.nevals; out[inst].iterm = C.c.iterm; out[inst].cols = C.cols;
        if (P.done) __hip_atomic_fetch_add(P.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#undef COLX
#undef COLG
#undef COLU
#undef COLIDX
}

}   /* namespace */

extern "C" int LR_SUPPORTED(int obj, int n, const nla_lbfgs_params *params)
{
    return obj >= 0 && n <= LR_NMAX && (params->exact == 0 || params->exact == 1);       /* exact 2 / 3: the streaming kernel asked for by name */
}

extern "C" int LR_BATCH(int obj, int n, int ld, int mf, int count, const double *lb, const double *ub, double *X,
                        double *work, double *hist, const nla_lbfgs_params *params, nla_lbfgs_result *out, void *stream)
{
    if (count <= 0) return 0;
    hipStream_t st = (hipStream_t) stream;
    nla_lbfgs_params P = *params;
    if (P.sign == 0.) P.sign = 1.;
    if (!LR_SUPPORTED(obj, n, &P)) return (int) hipErrorInvalidValue;
#define CALL(O) do { if (P.exact) hipLaunchKernelGGL((LR_KERNEL<O, true>), dim3(count), dim3(LB_T), 0, st, n, ld, mf, count, lb, ub, X, work, hist, P, out); \
                     else hipLaunchKernelGGL((LR_KERNEL<O, false>), dim3(count), dim3(LB_T), 0, st, n, ld, mf, count, lb, ub, X, work, hist, P, out); } while (0)
    NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}
