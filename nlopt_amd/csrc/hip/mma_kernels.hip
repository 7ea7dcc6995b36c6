/* mma_kernels.hip — batched NLOPT_LD_MMA without nonlinear constraints (src/algs/mma/mma.c:146-449, m = 0) on
 * gfx950: one workgroup per local search, the whole outer/inner iteration on the device.  This is the default local
 * optimiser of NLOPT_GD_MLSL(_LDS) (src/api/optimize.c:763-768), so those run end to end on the device.
 *
 * With m = 0 the dual problem has no variables (the reference "solves" it by evaluating dual_func, mma.c:58-137,
 * twice: optimize.c:533-536 and mma.c:298): the inner step is the closed-form minimiser of the separable moving-
 * asymptote approximation, one coordinate per lane, fused here with the approximation's value gval and the
 * conservativeness weight wval (two workgroup sums), then the objective + gradient at the new point.
 *
 * Numerics: per-coordinate formulas are the reference's; gval/wval and the norms of the x-tolerance test are
 * workgroup reductions (fixed tree) instead of sequential sums ⇒ they differ by rounding only; the decisions
 * gval >= fcur, fcur < minf are taken on those values.
 *
 * params.exact != 0 ("amd_exact_dot"): gval / wval / the norms are accumulated in the reference's sequential order
 * (gval starts from fval, mma.c:74,119-121) — the iterates are then the reference's bit for bit.
 *
 * External evaluation (OBJ == NLA_OBJ_EXTERNAL, include/nlopt_amd.h): the kernel is a coroutine around its three
 * evaluation points — the start (mma.c:219), the inner iteration (mma.c:297, gradient only if inner_gradients) and the
 * repeated call with a gradient when a step is accepted and inner_gradients = 0 (mma.c:337).
 *
 * Roofline: per evaluation a search reads x, sigma, dfdx, lb, ub, writes xcur, re-reads it for the objective and
 * writes the gradient (64 n bytes, L2-resident: 7 vectors of n doubles per search); the time goes into the
 * objective's transcendentals — fp64 VALU, not HBM.
 */
#include "local_common.h"
#include "../../../include/nlopt_amd.h"

#define MMA_RHOMIN 1e-5                                                       /* mma.c:40 */

struct mma_saved {
    double rho, fcur, minf, fprev, gval, wval;
    int ret, k, nevals, fcalls, inner_nevals, inner_done, point, pad;
};

template <int OBJ>
__global__ __launch_bounds__(LB_T) void mma_batch_kernel(int n, int ld, int count, const double *__restrict__ lb,
                                                          const double *__restrict__ ub, const double *__restrict__ sigma_init,
                                                          double *__restrict__ X, double *__restrict__ work, nla_mma_params P,
                                                          nla_lbfgs_result *__restrict__ out, nla_local_ext E)
{
    constexpr bool EXT = OBJ == NLA_OBJ_EXTERNAL;
    __shared__ lb_shared S;
    __shared__ double oscratch[2 * LB_W];
    __shared__ lb_exact_buf XB;
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (inst >= count) return;
    if (EXT && E.resume && E.req[inst].state != 1) return;
    double *x = X + (size_t) inst * ld;
    double *sigma = work + (size_t) inst * 6 * ld, *dfdx = sigma + ld, *dfdx_cur = dfdx + ld, *xcur = dfdx_cur + ld,
           *xprev = xcur + ld, *xprevprev = xprev + ld;
    int ret = 1 /* NLOPT_SUCCESS */, k = 0, nevals = 0, fcalls = 0, inner_nevals = 0, inner_done = 0, forced = 0, tmo = 0;
    double rho = P.rho_init, fcur = 0, minf = 0, fprev = 0, gval = 0, wval = 0;
    mma_saved *sv = EXT ? (mma_saved *) E.save + inst : nullptr;

    /* an evaluation point (see lbfgs_kernels.hip): XPT = the point, GPT = where its gradient goes */
#define MMA_EVAL(POINT, LABEL, XPT, GPT, WANTG)                                                                          \
    if (EXT) {                                                                                                           \
        for (int j = tid; j < n; j += LB_T) E.EX[(size_t) inst * ld + j] = (XPT)[j];                                     \
        if (tid == 0) {                                                                                                  \
            sv->rho = rho; sv->fcur = fcur; sv->minf = minf; sv->fprev = fprev; sv->gval = gval; sv->wval = wval;        \
            sv->ret = ret; sv->k = k; sv->nevals = nevals; sv->fcalls = fcalls; sv->inner_nevals = inner_nevals;         \
            sv->inner_done = inner_done; sv->point = POINT;                                                              \
            E.req[inst].state = 1; E.req[inst].want_grad = (WANTG);                                                      \
        }                                                                                                                \
        return;                                                                                                          \
    LABEL:                                                                                                               \
        if (WANTG) for (int j = tid; j < n; j += LB_T) (GPT)[j] = E.EG[(size_t) inst * ld + j];                          \
        __syncthreads();                                                                                                 \
        fcur = E.EF[inst];                                                                                               \
    } else fcur = lb_objgrad<EXT ? 0 : OBJ>(n, XPT, GPT, S, oscratch, P.exact, XB, P.sign)
#define MMA_POLL() do { if (!EXT && P.abort) { const int ab_ = lb_poll_abort(P.abort); forced = ab_ == -999; tmo = ab_ == 100; } } while (0)

    if (EXT) { forced = E.forced; tmo = E.timeout; }
    if (EXT && E.resume) {
        rho = sv->rho; fcur = sv->fcur; minf = sv->minf; fprev = sv->fprev; gval = sv->gval; wval = sv->wval; ret = sv->ret;
        k = sv->k; nevals = sv->nevals; fcalls = sv->fcalls; inner_nevals = sv->inner_nevals; inner_done = sv->inner_done;
        __syncthreads();
        if (tid == 0) E.req[inst].state = 0;
        if (sv->point == 0) goto resume_first;
        if (sv->point == 1) goto resume_inner;
        goto resume_regrad;
    }

    for (int j = tid; j < n; j += LB_T) {                                     /* mma.c:203-211 */
        double sg = (sigma_init && sigma_init[j] > 0) ? sigma_init[j] : (isinf(ub[j]) || isinf(lb[j])) ? 1.0 : 0.5 * (ub[j] - lb[j]);
        sigma[j] = sg > P.sigma_min ? sg : P.sigma_min;
        xcur[j] = x[j];
    }
    __syncthreads();
    MMA_EVAL(0, resume_first, x, dfdx, 1);                                    /* mma.c:219-221 */
    minf = fcur;
    if (P.ftrace && tid == 0 && fcalls < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + fcalls] = fcur;
    ++nevals; ++fcalls;
    MMA_POLL();
    if (forced) ret = -5;

    while (ret == 1) {                                                        /* outer iterations, mma.c:253 */
        inner_nevals = 0;
        fprev = fcur;
        MMA_POLL();
        if (forced) ret = -5;                                                 /* NLOPT_FORCED_STOP */
        else if (P.maxeval > 0 && nevals >= P.maxeval) ret = 5;               /* NLOPT_MAXEVAL_REACHED */
        else if (tmo) ret = 6;                                                /* NLOPT_MAXTIME_REACHED */
        else if (minf < P.minf_max) ret = 2;                                  /* NLOPT_MINF_MAX_REACHED (feasible: no constraints) */
        if (ret != 1) break;
        ++k;
        for (int j = tid; j < n; j += LB_T) {
            if (k > 1) xprevprev[j] = xprev[j];
            xprev[j] = xcur[j];
        }
        for (;;) {                                                            /* inner iterations, mma.c:265 */
            {
                double gs = 0, ws = 0;
                __syncthreads();
                for (int j = tid; j < n; j += LB_T) {                         /* dual_func with m = 0, mma.c:88-124 */
                    const double sg = sigma[j], xj = x[j], d = dfdx[j];
                    double xc = xj, tg = 0, tw = 0;
                    if (sg != 0) {
                        const double sigma2 = sg * sg, v = fabs(d) * sg + 0.5 * rho, u = d * sigma2;
                        const double q = u / (v * sg);
                        double dx = (u / v) / (-1 - sqrt(fabs(1 - q * q))), dx2, denominv;
                        xc = xj + dx;
                        if (xc > ub[j]) xc = ub[j];
                        else if (xc < lb[j]) xc = lb[j];
                        if (xc > xj + 0.9 * sg) xc = xj + 0.9 * sg;
                        else if (xc < xj - 0.9 * sg) xc = xj - 0.9 * sg;
                        dx = xc - xj;
                        dx2 = dx * dx;
                        denominv = 1.0 / (sigma2 - dx2);
                        tg = (d * (sigma2 * dx) + (fabs(d) * sg + 0.5 * rho) * dx2) * denominv;
                        tw = 0.5 * dx2 * denominv;
                        gs += tg; ws += tw;
                    }
                    xcur[j] = xc;
                }
                if (P.exact) {
                    /* the reference's order: gval starts from fval and takes the terms of j = 0 .. n-1 one by one (a
                     * coordinate with sigma = 0 is skipped: adding +0.0 is the same); the terms are recomputed from the
                     * step just stored */
                    __syncthreads();
                    gval = minf; wval = 0;
                    lb_seq_sum2<false>(0, n, &gval, &wval, [&](int j, double *pa, double *pb) {
                        const double sg = sigma[j], d = dfdx[j];
                        double tg = 0, tw = 0;
                        if (sg != 0) {
                            const double sigma2 = sg * sg, dx = xcur[j] - x[j], dx2 = dx * dx, denominv = 1.0 / (sigma2 - dx2);
                            tg = (d * (sigma2 * dx) + (fabs(d) * sg + 0.5 * rho) * dx2) * denominv;
                            tw = 0.5 * dx2 * denominv;
                        }
                        *pa = tg; *pb = tw;
                    }, XB);
                } else {
                    gval = minf + lb_block_sum(gs, S);
                    wval = lb_block_sum(ws, S);
                }
                __syncthreads();
            }
            MMA_EVAL(1, resume_inner, xcur, dfdx_cur, P.inner_gradients);     /* mma.c:297 */
            if (P.ftrace && tid == 0 && fcalls < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + fcalls] = fcur;
            ++nevals; ++inner_nevals; ++fcalls;
            MMA_POLL();
            if (forced) { ret = -5; break; }
            inner_done = (gval >= fcur) || (P.inner_maxeval > 0 && inner_nevals == P.inner_maxeval);
            if (P.always_improve ? fcur < minf : inner_done) {               /* mma.c:329-331 with feasible = feasible_cur = 1 */
                if (!P.inner_gradients) {
                    if (EXT) { MMA_EVAL(2, resume_regrad, xcur, dfdx_cur, 3); }   /* evaluated again, with a gradient; not counted (mma.c:336-339) */
                    if (P.ftrace && tid == 0 && fcalls < P.ftrace_cap) P.ftrace[(size_t) inst * P.ftrace_cap + fcalls] = fcur;
                    ++fcalls;
                    if (forced) { ret = -5; break; }
                    inner_done = gval >= fcur;                                /* mma.c:343: recomputed WITHOUT the inner_maxeval clause */
                }
                minf = fcur;
                for (int j = tid; j < n; j += LB_T) { x[j] = xcur[j]; dfdx[j] = dfdx_cur[j]; }
            }
            __syncthreads();
            if (forced) ret = -5;
            else if (P.maxeval > 0 && nevals >= P.maxeval) ret = 5;
            else if (tmo) ret = 6;
            else if (minf < P.minf_max) ret = 2;
            if (ret != 1 || inner_done) break;
            if (fcur > gval) {                                                /* mma.c:394-395 */
                const double r1 = 10 * rho, r2 = 1.1 * (rho + (fcur - gval) / wval);
                rho = r1 < r2 ? r1 : r2;
            }
        }
        if (ret != 1) break;
        {                                                                     /* mma.c:408-411; nlopt_stop_x, stop.c:98-108 */
            double nx = 0, ndx = 0;
            const double *w = P.x_weights;
            if (P.exact) {
                __syncthreads();
                ndx = lb_seq_sum(n, 0., [&](int j) { return w ? w[j] * fabs(xcur[j] - xprev[j]) : fabs(xcur[j] - xprev[j]); }, XB.a);
                nx = lb_seq_sum(n, 0., [&](int j) { return w ? w[j] * fabs(xcur[j]) : fabs(xcur[j]); }, XB.a);
            } else {
                for (int j = tid; j < n; j += LB_T) {
                    if (w) { nx += w[j] * fabs(xcur[j]); ndx += w[j] * fabs(xcur[j] - xprev[j]); }
                    else { nx += fabs(xcur[j]); ndx += fabs(xcur[j] - xprev[j]); }
                }
                nx = lb_block_sum(nx, S);
                ndx = lb_block_sum(ndx, S);
            }
            if (!isinf(fprev)) {
                const double df = fabs(fcur - fprev);
                if (df < P.ftol_abs || df < P.ftol_rel * (fabs(fcur) + fabs(fprev)) * 0.5 || (P.ftol_rel > 0 && fcur == fprev)) ret = 3;
            }
            if (ndx < P.xtol_rel * nx) ret = 4;
            else if (P.xtol_abs) {
                int viol = 0;
                for (int j = tid; j < n; j += LB_T) viol += fabs(xcur[j] - xprev[j]) >= P.xtol_abs[j];
                if (lb_block_isum(viol, S) == 0) ret = 4;
            }
        }
        if (ret != 1) break;
        rho = 0.1 * rho > MMA_RHOMIN ? 0.1 * rho : MMA_RHOMIN;                /* mma.c:415 */
        if (k > 1)
            for (int j = tid; j < n; j += LB_T) {                             /* mma.c:423-435 */
                const double dx2 = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j]);
                const double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1);
                double sg = sigma[j] * gam;
                if (!isinf(ub[j]) && !isinf(lb[j])) {
                    const double hi = 10 * (ub[j] - lb[j]), lo = 0.01 * (ub[j] - lb[j]);
                    sg = sg < hi ? sg : hi;
                    sg = sg > lo ? sg : lo;
                }
                sigma[j] = sg > P.sigma_min ? sg : P.sigma_min;
            }
        __syncthreads();
    }
    /* iterm: objective calls made (what MLSL's counting wrapper sees, mlsl.c:246-251); cols: outer iterations */
    if (tid == 0) {
        out[inst].f = minf; out[inst].ret = ret; out[inst].nevals = nevals; out[inst].iterm = fcalls; out[inst].cols = k;
        if (P.done) __hip_atomic_fetch_add(P.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (EXT) E.req[inst].state = 2;
    }
#undef MMA_EVAL
#undef MMA_POLL
}

extern "C" size_t nla_mma_work_doubles(int ld, int count) { return (size_t) count * 6 * (size_t) ld; }
extern "C" size_t nla_mma_save_bytes(void) { return sizeof(mma_saved); }

extern "C" int nla_k_mma_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *sigma_init,
                               double *X, double *work, const nla_mma_params *params, nla_lbfgs_result *out,
                               const nla_local_ext *ext, void *stream)
{
    if (count <= 0) return 0;
    hipStream_t st = (hipStream_t) stream;
    nla_mma_params P = *params;
    nla_local_ext E = {};
    if (P.sign == 0.) P.sign = 1.;
    if (obj == NLA_OBJ_EXTERNAL) {
        if (!ext || !ext->req || !ext->EX || !ext->EG || !ext->EF || !ext->save) return (int) hipErrorInvalidValue;
        E = *ext;
    }
#define CALL(O) hipLaunchKernelGGL((mma_batch_kernel<O>), dim3(count), dim3(LB_T), 0, st, n, ld, count, lb, ub, sigma_init, X, work, P, out, E)
    if (obj == NLA_OBJ_EXTERNAL) { CALL(NLA_OBJ_EXTERNAL); }
    else NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}
