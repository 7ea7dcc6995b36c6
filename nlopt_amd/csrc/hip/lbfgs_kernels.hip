/* lbfgs_kernels.hip — batched NLOPT_LD_LBFGS (Luksan's PLIS, src/algs/luksan/plis.c:106-417) on gfx950:
 * one workgroup per local search, the whole optimisation loop on the device.
 *
 * Reference loops replaced: the Strang recurrences mxdrcb / mxdrcf over the k <= mf history pairs
 * (mssubs.c:353-441: k sequential dot + axpy pairs each), the masked vector kernels mxudot /
 * mxudir / mxuneg (mssubs.c:601-790), the bound handling pcbs04 / pyadc0 / pyrmc0 / pytrcg /
 * pytrcs / pytrcd (pssubs.c), and the objective + gradient evaluation.  The scalar control flow —
 * line search PS1L01 / PNINT1 and the termination test PYFUT1 — is the shared source
 * ../lbfgs_scalar.h, executed redundantly by every thread on workgroup-uniform values, so a local
 * search needs no host round trip at all.
 *
 * Roofline: HBM/L2 traffic of the history, 32 k n bytes per iteration (k columns, two matrices, a
 * dot and an axpy pass each); one workgroup streams its own 2 x mf x n history (21 MB at n = 4096),
 * the batch keeps every CU busy.  History layout: per instance mf columns of ld doubles for the
 * x-differences, then mf for the g-differences; the reference shifts all columns every iteration
 * (mxdrsu, mssubs.c:503-524) — here column "i-th newest" is ring-indexed, same numbers.
 *
 * Numerics: per-element formulas and the order of the scalar logic are the reference's; dot
 * products are workgroup reductions (fixed tree: thread-strided partials, xor-butterfly per
 * wavefront, wavefronts in order), so sums differ from the reference's sequential ones by
 * rounding only.  With params.exact != 0 ("amd_exact_dot") every sum is accumulated in the
 * reference's order instead (local_common.h) and the iterates are the reference's bit for bit
 * (up to the device libm inside a device objective).
 *
 * External evaluation (OBJ == NLA_OBJ_EXTERNAL): the objective is not on the device — a host
 * callback (the reference's nlopt_func contract: called on the caller's thread, one x at a time)
 * or a user-supplied device module.  The kernel is then a coroutine: at each of its two evaluation
 * points (plis.c:260 and :390) it writes the point into EX, saves its scalar state and returns;
 * the host delivers f / gradient into EF / EG and launches it again with ext.resume = 1.
 */
#include "local_common.h"
#include <limits.h>
#include "../lbfgs_scalar.h"
#include "../../../include/nlopt_amd.h"

/* scalar state of one search across an external evaluation (everything else lives in the instance's vectors) */
struct lb_saved {
    lb_ls_state lss; lb_ls_io q; lb_counters c;
    double gmax, umax, fval, fo, p, po, gnorm, snorm, rmax, rmin;
    int kd, nred, maxst, xstop, nevals, cols, head, point;
};

#define LB_EPT 16                  /* coordinates per thread held in registers by the direction loops (n <= 4096) */

__device__ __forceinline__ void lb_project(int n, double *x, const int *ix, const double *xl, const double *xu, double eps9)   /* pcbs04 */
{
    for (int i = threadIdx.x; i < n; i += LB_T) {
        const int t = ix[i] < 0 ? -ix[i] : ix[i];
        double v = x[i];
        if ((t == 1 || t == 3 || t == 4) && v <= xl[i] + eps9 * LB_MAX(fabs(xl[i]), 1.)) v = xl[i];
        if ((t == 2 || t == 3 || t == 4) && v >= xu[i] - eps9 * LB_MAX(fabs(xu[i]), 1.)) v = xu[i];
        x[i] = v;
    }
}
__device__ __forceinline__ void lb_add_active(int n, double *x, int *ix, const double *xl, const double *xu)                 /* pyadc0 */
{
    for (int i = threadIdx.x; i < n; i += LB_T) {
        const int ii = ix[i], t = ii < 0 ? -ii : ii;
        if (t >= 5) ix[i] = -t;
        else if ((t == 1 || t == 3 || t == 4) && x[i] <= xl[i]) { x[i] = xl[i]; ix[i] = (t == 4) ? -3 : -t; }
        else if ((t == 2 || t == 3 || t == 4) && x[i] >= xu[i]) { x[i] = xu[i]; ix[i] = (t == 3) ? -4 : -t; }
    }
}

/* The two Strang loops (mxdrcb / mxdrcf, mssubs.c:353-441) with the direction held in REGISTERS (thread t owns coordinates t, t+256, ...:
 * the same partial-sum order as lb_mdot) and the next history column prefetched while the current dot product is being reduced — one
 * barrier pair per column, no traffic for s.  A function of its own, not inlined: its five register arrays (160 VGPRs) are then
 * the callee's whole budget, and what the caller keeps alive (the line search's state, the counters) is saved ONCE around the call
 * instead of being spilled and reloaded inside the per-column loop — the kernel fits two workgroups per CU that way
 * (round 2: 443-490 VGPRs, one workgroup per CU, 320 searches taking turns on 256 CUs).  Returns |s|. */
__device__ __attribute__((noinline)) double lb_strang_in_registers(int n, int k, int mf, int head, int ld, const int *__restrict__ ix,
                                                                    const double *__restrict__ gf, double *__restrict__ s,
                                                                    const double *__restrict__ hx, const double *__restrict__ hg,
                                                                    const double *__restrict__ ucol, double *__restrict__ vcol, double b)
{
    /* the per-column reductions alternate between two LDS slots: a wavefront may already write the NEXT column's partial while a
     * slower one still reads this column's — ONE barrier per reduction instead of lb_block_sum's two (same tree, same sums) */
    __shared__ double red[2][LB_W];
    const int tid = threadIdx.x;
    int par = 0;
    auto block_sum = [&](double v) {
        v = lb_wave_sum(v);
        if ((tid & 63) == 0) red[par][tid >> 6] = v;
        __syncthreads();
        double t = red[par][0];
#pragma unroll
        for (int w = 1; w < LB_W; ++w) t += red[par][w];
        par ^= 1;
        return t;
    };
    double snorm, a;
#define COLX(i) (hx + (size_t) ((head + (i) - 1) % mf) * ld)
#define COLG(i) (hg + (size_t) ((head + (i) - 1) % mf) * ld)
#define COLU(i) (ucol[(head + (i) - 1) % mf])
    double sr[LB_EPT], c1[LB_EPT], c2[LB_EPT], n1[LB_EPT], n2[LB_EPT];
    unsigned live = 0;
    const int ept = (n + LB_T - 1) / LB_T;
#pragma unroll
    for (int e = 0; e < LB_EPT; ++e) {
        const int i = tid + e * LB_T;
        sr[e] = 0.; c1[e] = 0.; c2[e] = 0.; n1[e] = 0.; n2[e] = 0.;
        if (e < ept && i < n && ix[i] >= 0) { live |= 1u << e; sr[e] = -gf[i]; }     /* mxuneg */
    }
    auto load_col = [&](int j, double *px, double *pg) {
        const double *cx = COLX(j), *cg = COLG(j);
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) if (live & (1u << e)) { px[e] = cx[tid + e * LB_T]; pg[e] = cg[tid + e * LB_T]; }
    };
    load_col(1, c1, c2);
    for (int j = 1; j <= k; ++j) {                       /* mxdrcb */
        if (j < k) load_col(j + 1, n1, n2);
        double t = 0;
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) if (live & (1u << e)) t += sr[e] * c1[e];
        const double v = COLU(j) * block_sum(t);
        if (tid == 0) vcol[j - 1] = v;
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) if (live & (1u << e)) { sr[e] = sr[e] + (-v) * c2[e]; c1[e] = n1[e]; c2[e] = n2[e]; }
    }
    {
        double t = 0;
        const double *cg = COLG(1);
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) if (live & (1u << e)) { const double gv = cg[tid + e * LB_T]; t += gv * gv; }
        a = block_sum(t);
        if (a > 0.) {
            const double sc = b / a;
#pragma unroll
            for (int e = 0; e < LB_EPT; ++e) sr[e] = sr[e] * sc;
        }
    }
    load_col(k, c1, c2);
    for (int j = k; j >= 1; --j) {                       /* mxdrcf */
        if (j > 1) load_col(j - 1, n1, n2);
        double t = 0;
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) if (live & (1u << e)) t += sr[e] * c2[e];
        const double tt = COLU(j) * block_sum(t);
        const double w = vcol[j - 1] - tt;
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) if (live & (1u << e)) { sr[e] = sr[e] + w * c1[e]; c1[e] = n1[e]; c2[e] = n2[e]; }
    }
    {
        double t = 0;
#pragma unroll
        for (int e = 0; e < LB_EPT; ++e) {
            const int i = tid + e * LB_T;
            if (e < ept && i < n) s[i] = sr[e];
            if (live & (1u << e)) t += sr[e] * sr[e];
        }
        snorm = sqrt(block_sum(t));
    }
    __syncthreads();
#undef COLX
#undef COLG
#undef COLU
    return snorm;
}

/* two workgroups per CU (<= 256 registers per lane): what does not fit is spilled in the scalar outer logic, once per iteration —
 * the per-column loops live in lb_strang_in_registers and stay spill-free (checked in the ISA: only its prologue / epilogue touch scratch) */
#ifndef LB_WAVES_PER_EU
#define LB_WAVES_PER_EU 2
#endif
template <int OBJ>
__global__ __launch_bounds__(LB_T) __attribute__((amdgpu_waves_per_eu(LB_WAVES_PER_EU, LB_WAVES_PER_EU))) void lbfgs_batch_kernel(int n, int ld, int mf, int count, const double *__restrict__ lb,
                                                            const double *__restrict__ ub, double *__restrict__ X,
                                                            double *__restrict__ work, int *__restrict__ iwork,
                                                            double *__restrict__ hist, nla_lbfgs_params P,
                                                            nla_lbfgs_result *__restrict__ out, nla_local_ext E)
{
    constexpr bool EXT = OBJ == NLA_OBJ_EXTERNAL;
    __shared__ lb_shared S;
    __shared__ double oscratch[2 * LB_W];
    __shared__ lb_exact_buf XB;
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (inst >= count) return;
    if (EXT && E.resume && E.req[inst].state != 1) return;          /* finished earlier (or never asked) */
    double *x = X + (size_t) inst * ld;
    double *gf = work + (size_t) inst * 4 * ld, *s = gf + ld, *xl = s + ld, *xu = xl + ld;
    int *ix = iwork + (size_t) inst * ld;
    double *hx = hist + (size_t) inst * 2 * (size_t) mf * ld, *hg = hx + (size_t) mf * ld;
    /* per-column scalars u, v live behind the instance's vectors */
    double *ucol = work + (size_t) count * 4 * ld + (size_t) inst * 2 * mf, *vcol = ucol + mf;
    int head = 0;
#define COLX(i) (hx + (size_t) ((head + (i) - 1) % mf) * ld)
#define COLG(i) (hg + (size_t) ((head + (i) - 1) % mf) * ld)
#define COLU(i) (ucol[(head + (i) - 1) % mf])

    lb_ls_state lss;
    lb_ls_io q;
    lb_counters c;
    lb_stop ls;
    double gmax = 0, umax = 0, fval, fo, p = 0, po = 0, a, b, gnorm, snorm = 0, rmax, rmin = 0;
    const double eta9 = 1e120, eps8 = 1., eps9 = 1e-8, alf1 = 1e-10, alf2 = 1e10, told = 1e-4, xmax = 1e16, maxf = 1e20,
                 minf_est = -HUGE_VAL;
    int kd = 1, ld_ = -1, nred = 0, maxst = 0, xstop = 0, nevals = 0, k, cols = 0, forced = 0, tmo = 0;
    double xtol_rel = P.xtol_rel, tolg = P.tolg;
    lb_saved *sv = EXT ? (lb_saved *) E.save + inst : nullptr;
    (void) ld_;
#define MDOT(u, v) (P.exact ? lb_mdot_exact(n, u, v, ix, XB) : lb_mdot(n, u, v, ix, S))
    /* an evaluation point: device objective -> evaluate here; external -> publish the point, save the state, leave */
#define LB_EVAL(POINT, LABEL, FOUT)                                                                                     \
    if (EXT) {                                                                                                          \
        for (int i = tid; i < n; i += LB_T) E.EX[(size_t) inst * ld + i] = x[i];                                        \
        if (tid == 0) {                                                                                                 \
            sv->lss = lss; sv->q = q; sv->c = c; sv->gmax = gmax; sv->umax = umax; sv->fval = fval; sv->fo = fo;        \
            sv->p = p; sv->po = po; sv->gnorm = gnorm; sv->snorm = snorm; sv->rmax = rmax; sv->rmin = rmin; sv->kd = kd; \
            sv->nred = nred; sv->maxst = maxst; sv->xstop = xstop; sv->nevals = nevals; sv->cols = cols;                \
            sv->head = head; sv->point = POINT;                                                                         \
            E.req[inst].state = 1; E.req[inst].want_grad = 1;                                                           \
        }                                                                                                               \
        return;                                                                                                         \
    LABEL:                                                                                                              \
        for (int i = tid; i < n; i += LB_T) gf[i] = E.EG[(size_t) inst * ld + i];                                       \
        __syncthreads();                                                                                                \
        FOUT = E.EF[inst];                                                                                              \
    } else FOUT = lb_objgrad<EXT ? 0 : OBJ>(n, x, gf, S, oscratch, P.exact, XB, P.sign)

    if (xtol_rel <= 0.) xtol_rel = 1e-16;                                    /* plis.c:202-214 */
    ls.minf_max = P.minf_max; ls.ftol_rel = P.ftol_rel <= 0. ? 1e-14 : P.ftol_rel; ls.ftol_abs = P.ftol_abs; ls.maxeval = P.maxeval;
    if (tolg <= 0.) tolg = 1e-8;
    memset(&c, 0, sizeof c);
    memset(&lss, 0, sizeof lss);
    memset(&q, 0, sizeof q);
    fval = 0; fo = minf_est; gnorm = 0; rmax = eta9;
    if (EXT) { forced = E.forced; tmo = E.timeout; }
    if (EXT && E.resume) {                                                   /* continue where the search left */
        lss = sv->lss; q = sv->q; c = sv->c; gmax = sv->gmax; umax = sv->umax; fval = sv->fval; fo = sv->fo; p = sv->p;
        po = sv->po; gnorm = sv->gnorm; snorm = sv->snorm; rmax = sv->rmax; rmin = sv->rmin; kd = sv->kd; nred = sv->nred;
        maxst = sv->maxst; xstop = sv->xstop; nevals = sv->nevals; cols = sv->cols; head = sv->head;
        __syncthreads();
        if (tid == 0) E.req[inst].state = 0;
        if (sv->point == 0) goto resume_first; else goto resume_linesearch;
    }

    for (int i = tid; i < n; i += LB_T) {                                    /* plis.c:463-469 */
        const int lbu = lb[i] <= -0.99 * HUGE_VAL, ubu = ub[i] >= 0.99 * HUGE_VAL;
        int t = lbu ? (ubu ? 0 : 2) : (ubu ? 1 : (lb[i] == ub[i] ? 5 : 3));
        double l = lb[i], u = ub[i];
        if ((t == 3 || t == 4) && u <= l) { u = l; t = 5; }                  /* plis.c:232-241 */
        else if (t == 5 || t == 6) { l = x[i]; u = x[i]; t = 5; }
        ix[i] = t; xl[i] = l; xu[i] = u;
        hx[i] = 0.; hg[i] = 0.;               /* the reference zero-fills xo (plis.c:475); column 1 is read before it is written */
    }
    c.ites = 1; c.mtesx = 2; c.mtesf = 2; c.iters = 2; c.ires1 = 999; c.ires2 = 0; c.kd = 1;
    c.mit = INT_MAX; c.mfg = P.maxeval > 0 ? P.maxeval : INT_MAX;
    c.kit = -(c.ires1 * n + c.ires2);
    __syncthreads();
    lb_project(n, x, ix, xl, xu, eps9);
    __syncthreads();
    lb_add_active(n, x, ix, xl, xu);
    __syncthreads();
    LB_EVAL(0, resume_first, fval);
    if (P.ftrace && tid == 0 && nevals < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + nevals] = fval;
    ++nevals; ++c.nfg;
    if (!EXT && P.abort) tmo = lb_poll_abort(P.abort) == 100;
    if (tmo) c.iterm = 100;                                                  /* plis.c:263 */

    while (c.iterm != 100) {
        /* pytrcg: largest free gradient component, largest wrong-signed multiplier on an active bound */
        {
            double gm = 0, um = 0;
            for (int i = tid; i < n; i += LB_T) {
                const double t = gf[i];
                const int ii = ix[i];
                if (ii >= 0) gm = LB_MAX(gm, fabs(t));
                else if (ii <= -5) { }
                else if (ii == -1 || ii == -3) { if (-t > um) um = -t; }
                else if (ii == -2 || ii == -4) { if (t > um) um = t; }
            }
            gmax = lb_block_max(gm, S);
            umax = lb_block_max(um, S);
        }
        c.kd = kd;
        if (!EXT && P.abort) { const int ab = lb_poll_abort(P.abort); forced = ab == -999; tmo = ab == 100; }
        lb_pyfut1(n, fval, &fo, umax, gmax, xstop, &ls, forced, nevals, tolg, &c);
        if (c.iterm != 0) break;
        if (tmo) { c.iterm = 100; break; }                                   /* plis.c:273 */
        if (rmax > 0. && umax > eps8 * gmax) {                               /* pyrmc0: release wrong-signed active bounds */
            int rel = 0;
            for (int i = tid; i < n; i += LB_T) {
                const int t = ix[i];
                if (t >= 0 || t <= -5) continue;
                if ((t == -1 || t == -3) && -gf[i] <= 0.) continue;
                if ((t == -2 || t == -4) && gf[i] <= 0.) continue;
                ++rel;
                ix[i] = LB_MIN(-t, 3);
            }
            if (lb_block_isum(rel, S) > 1) c.irest = LB_MAX(c.irest, 1);
        }
        __syncthreads();
    direction:
        gnorm = sqrt(MDOT(gf, gf));
        if (c.irest == 0) {
            k = LB_MIN(c.nit - c.kit, mf);
            if (k <= 0) c.irest = LB_MAX(c.irest, 1);
            else {
                b = MDOT(COLX(1), COLG(1));
                if (b <= 0.) c.irest = LB_MAX(c.irest, 1);
                else {
                    if (tid == 0) COLU(1) = 1. / b;
                    cols += k;
                    if (n <= LB_T * LB_EPT && !P.exact) {
                        __syncthreads();                       /* COLU(1) visible */
                        snorm = lb_strang_in_registers(n, k, mf, head, ld, ix, gf, s, hx, hg, ucol, vcol, b);
                    } else {
                        for (int i = tid; i < n; i += LB_T) s[i] = ix[i] >= 0 ? -gf[i] : 0.;      /* mxuneg */
                        __syncthreads();
                        for (int j = 1; j <= k; ++j) {                           /* mxdrcb */
                            const double *cx = COLX(j), *cg = COLG(j);
                            const double v = COLU(j) * MDOT(s, cx);
                            if (tid == 0) vcol[j - 1] = v;
                            for (int i = tid; i < n; i += LB_T) if (ix[i] >= 0) s[i] = s[i] + (-v) * cg[i];
                            __syncthreads();
                        }
                        a = MDOT(COLG(1), COLG(1));
                        if (a > 0.) { const double sc = b / a; for (int i = tid; i < n; i += LB_T) s[i] = s[i] * sc; __syncthreads(); }
                        for (int j = k; j >= 1; --j) {                           /* mxdrcf */
                            const double *cx = COLX(j), *cg = COLG(j);
                            const double t = COLU(j) * MDOT(s, cg);
                            const double w = vcol[j - 1] - t;
                            for (int i = tid; i < n; i += LB_T) if (ix[i] >= 0) s[i] = s[i] + w * cx[i];
                            __syncthreads();
                        }
                        snorm = sqrt(MDOT(s, s));
                    }
                    head = (head + mf - 1) % mf;                             /* mxdrsu: every column one older */
                }
            }
        }
        if (c.irest != 0) {                                                  /* steepest descent */
            for (int i = tid; i < n; i += LB_T) s[i] = ix[i] >= 0 ? -gf[i] : 0.;
            __syncthreads();
            snorm = gnorm;
            if (c.kit < c.nit) c.kit = c.nit;
            else { c.iterm = -10; if (c.iters < 0) c.iterm = c.iters - 5; }
        }
        if (kd > 0) p = MDOT(gf, s);
        if (snorm <= 0.) c.irest = LB_MAX(c.irest, 1);
        else if (p + told * gnorm * snorm <= 0.) c.irest = 0;
        else c.irest = LB_MAX(c.irest, 1);
        if (c.irest == 0) {
            nred = 0;
            rmin = alf1 * gnorm / snorm;
            rmax = LB_MIN(alf2 * gnorm / snorm, xmax / snorm);
        }
        if (c.iterm != 0) break;
        if (tmo) { c.iterm = 100; break; }                                   /* plis.c:371 */
        if (c.irest != 0) goto direction;
        /* pytrcs: save x, g in column 1; zero s on active bounds; largest step inside the box */
        q.fp = fo; fo = fval; po = p;
        {
            double *cx = COLX(1), *cg = COLG(1);
            double rm = rmax;
            for (int i = tid; i < n; i += LB_T) {
                cx[i] = x[i]; cg[i] = gf[i];
                if (ix[i] < 0) s[i] = 0.;
                else {
                    if ((ix[i] == 1 || ix[i] >= 3) && s[i] < -1. / eta9) rm = LB_MIN(rm, (xl[i] - x[i]) / s[i]);
                    if ((ix[i] == 2 || ix[i] >= 3) && s[i] > 1. / eta9) rm = LB_MIN(rm, (xu[i] - x[i]) / s[i]);
                }
            }
            rmax = lb_block_min(rm, S);
        }
        if (rmax != 0.) {
            q.f = fval; q.fo = fo; q.p = p; q.po = po; q.minf_est = minf_est; q.maxf = maxf; q.rmin = rmin; q.rmax = rmax;
            q.tols = 1e-4; q.tolp = .8; q.kd = kd; q.ld = -1; q.nit = c.nit; q.kit = c.kit; q.nred = nred; q.mred = 10;
            q.maxst = maxst; q.iest = 0; q.inits = 2; q.iters = c.iters; q.kters = 3; q.mes = 4; q.isys = 0;
            for (;;) {
                lb_ps1l01(&q, &lss);
                if (q.isys == 0) break;
                {
                    const double *xs = COLX(1);
                    for (int i = tid; i < n; i += LB_T) if (ix[i] >= 0) x[i] = xs[i] + q.r * s[i];
                }
                __syncthreads();
                lb_project(n, x, ix, xl, xu, eps9);
                __syncthreads();
                LB_EVAL(1, resume_linesearch, q.f);
                if (P.ftrace && tid == 0 && nevals < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + nevals] = q.f;
                ++nevals; ++c.nfg;
                q.p = MDOT(gf, s);
            }
            fval = q.f; p = q.p; kd = q.kd; nred = q.nred; maxst = q.maxst; c.iters = q.iters;
            if (c.iters <= 0) {                                              /* zero step: restore and restart */
                fval = fo; p = po;
                const double *cx = COLX(1), *cg = COLG(1);
                for (int i = tid; i < n; i += LB_T) { x[i] = cx[i]; gf[i] = cg[i]; }
                __syncthreads();
                c.irest = LB_MAX(c.irest, 1);
                goto direction;
            }
            /* pytrcd: column 1 := differences (zero on active coordinates); nlopt_stop_dx(x, dx) */
            {
                double *dx = COLX(1), *dg = COLG(1);
                double nx = 0, ndx = 0;
                for (int i = tid; i < n; i += LB_T) {
                    double ddx = x[i] - dx[i], ddg = gf[i] - dg[i];
                    if (ix[i] < 0) { ddx = 0.; ddg = 0.; }
                    dx[i] = ddx; dg[i] = ddg;
                    if (P.x_weights) { nx += P.x_weights[i] * fabs(x[i]); ndx += P.x_weights[i] * fabs(ddx); }
                    else { nx += fabs(x[i]); ndx += fabs(ddx); }
                }
                po = q.r * po; p = q.r * p;
                if (P.exact) {                                                /* stop.c:37-57 vector_norm, sequential */
                    __syncthreads();
                    const double *w = P.x_weights;
                    nx = lb_seq_sum(n, 0., [&](int i) { return w ? w[i] * fabs(x[i]) : fabs(x[i]); }, XB.a);
                    ndx = lb_seq_sum(n, 0., [&](int i) { return w ? w[i] * fabs(dx[i]) : fabs(dx[i]); }, XB.a);
                } else {
                    nx = lb_block_sum(nx, S);
                    ndx = lb_block_sum(ndx, S);
                }
                xstop = ndx < xtol_rel * nx;                                  /* nlopt_stop_dx, stop.c:110-120 */
                if (!xstop && P.xtol_abs) {
                    int viol = 0;
                    __syncthreads();
                    for (int i = tid; i < n; i += LB_T) viol += fabs(dx[i]) >= P.xtol_abs[i];
                    xstop = lb_block_isum(viol, S) == 0;
                }
            }
        }
        for (int i = tid; i < n; i += LB_T) if (ix[i] < 0) ix[i] = -ix[i];   /* mxvine */
        __syncthreads();
        lb_add_active(n, x, ix, xl, xu);
        __syncthreads();
    }
    if (tid == 0) {
        out[inst].f = fval; out[inst].ret = lb_result_of_iterm(c.iterm); out[inst].nevals = nevals; out[inst].iterm = c.iterm; out[inst].cols = cols;
        if (P.done) __hip_atomic_fetch_add(P.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (EXT) E.req[inst].state = 2;
    }
#undef MDOT
#undef LB_EVAL
#undef COLX
#undef COLG
#undef COLU
}

extern "C" size_t nla_lbfgs_work_doubles(int ld, int mf, int count) { return (size_t) count * (4 * (size_t) ld + 2 * (size_t) mf); }
extern "C" size_t nla_lbfgs_hist_doubles(int ld, int mf, int count) { return (size_t) count * 2 * (size_t) mf * (size_t) ld; }

extern "C" size_t nla_lbfgs_save_bytes(void) { return sizeof(lb_saved); }

extern "C" int nla_lbfgs_resident_supported(int obj, int n, const nla_lbfgs_params *params);
extern "C" int nla_k_lbfgs_batch_resident(int obj, int n, int ld, int mf, int count, const double *lb, const double *ub, double *X, double *work,
                                          double *hist, const nla_lbfgs_params *params, nla_lbfgs_result *out, void *stream);
extern "C" int nla_lbfgs_resident32_supported(int obj, int n, const nla_lbfgs_params *params);                      /* 4096 < n <= 8192 (lbfgs_resident32.hip) */
extern "C" int nla_k_lbfgs_batch_resident32(int obj, int n, int ld, int mf, int count, const double *lb, const double *ub, double *X, double *work,
                                            double *hist, const nla_lbfgs_params *params, nla_lbfgs_result *out, void *stream);

extern "C" int nla_k_lbfgs_batch(int obj, int n, int ld, int mf, int count, const double *lb, const double *ub, double *X,
                                 double *work, int *iwork, double *hist, const nla_lbfgs_params *params, nla_lbfgs_result *out,
                                 const nla_local_ext *ext, void *stream)
{
    if (count <= 0) return 0;
    hipStream_t st = (hipStream_t) stream;
    nla_lbfgs_params P = *params;
    nla_local_ext E = {};
    if (P.sign == 0.) P.sign = 1.;
    /* a device objective, n <= 4096, tree sums: the resident kernel (lbfgs_resident.hip) — the same search bit for bit;
     * exact == 2 / 3 ("amd_lbfgs_streaming"): tree sums / the reference's order on THIS kernel, for the tests that compare the two */
    if (nla_lbfgs_resident_supported(obj, n, &P)) return nla_k_lbfgs_batch_resident(obj, n, ld, mf, count, lb, ub, X, work, hist, &P, out, stream);
    if (nla_lbfgs_resident32_supported(obj, n, &P)) return nla_k_lbfgs_batch_resident32(obj, n, ld, mf, count, lb, ub, X, work, hist, &P, out, stream);
    if (P.exact == 2) P.exact = 0;
    if (P.exact == 3) P.exact = 1;                 /* the reference's summation order on THIS kernel */
    if (obj == NLA_OBJ_EXTERNAL) {
        if (!ext || !ext->req || !ext->EX || !ext->EG || !ext->EF || !ext->save) return (int) hipErrorInvalidValue;
        E = *ext;
    }
#define CALL(O) hipLaunchKernelGGL((lbfgs_batch_kernel<O>), dim3(count), dim3(LB_T), 0, st, n, ld, mf, count, lb, ub, X, work, iwork, hist, P, out, E)
    if (obj == NLA_OBJ_EXTERNAL) { CALL(NLA_OBJ_EXTERNAL); }
    else NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}
