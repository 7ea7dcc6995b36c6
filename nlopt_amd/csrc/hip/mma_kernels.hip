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
 * Roofline: per evaluation a search reads x, sigma, dfdx, lb, ub, writes xcur, re-reads it for the objective and
 * writes the gradient (64 n bytes, L2-resident: 7 vectors of n doubles per search); the time goes into the
 * objective's transcendentals — fp64 VALU, not HBM.
 */
#include "local_common.h"
#include "../../../include/nlopt_amd.h"

#define MMA_RHOMIN 1e-5                                                       /* mma.c:40 */

template <int OBJ>
__global__ __launch_bounds__(LB_T) void mma_batch_kernel(int n, int ld, int count, const double *__restrict__ lb,
                                                          const double *__restrict__ ub, const double *__restrict__ sigma_init,
                                                          double *__restrict__ X, double *__restrict__ work, nla_mma_params P,
                                                          nla_lbfgs_result *__restrict__ out)
{
    __shared__ lb_shared S;
    __shared__ double oscratch[2 * LB_W];
    const int inst = blockIdx.x, tid = threadIdx.x;
    if (inst >= count) return;
    double *x = X + (size_t) inst * ld;
    double *sigma = work + (size_t) inst * 6 * ld, *dfdx = sigma + ld, *dfdx_cur = dfdx + ld, *xcur = dfdx_cur + ld,
           *xprev = xcur + ld, *xprevprev = xprev + ld;
    int ret = 1 /* NLOPT_SUCCESS */, k = 0, nevals = 0, fcalls = 0;
    double rho = P.rho_init, fcur, minf;

    for (int j = tid; j < n; j += LB_T) {                                     /* mma.c:203-211 */
        double sg = (sigma_init && sigma_init[j] > 0) ? sigma_init[j] : (isinf(ub[j]) || isinf(lb[j])) ? 1.0 : 0.5 * (ub[j] - lb[j]);
        sigma[j] = sg > P.sigma_min ? sg : P.sigma_min;
        xcur[j] = x[j];
    }
    __syncthreads();
    fcur = minf = lb_objgrad<OBJ>(n, x, dfdx, S, oscratch);                   /* mma.c:219-221 */
    ++nevals; ++fcalls;

    for (;;) {                                                                /* outer iterations, mma.c:253 */
        int inner_nevals = 0;
        const double fprev = fcur;
        if (P.maxeval > 0 && nevals >= P.maxeval) ret = 5;                    /* NLOPT_MAXEVAL_REACHED */
        else if (minf < P.minf_max) ret = 2;                                  /* NLOPT_MINF_MAX_REACHED (feasible: no constraints) */
        if (ret != 1) break;
        ++k;
        for (int j = tid; j < n; j += LB_T) {
            if (k > 1) xprevprev[j] = xprev[j];
            xprev[j] = xcur[j];
        }
        for (;;) {                                                            /* inner iterations, mma.c:265 */
            double gs = 0, ws = 0;
            for (int j = tid; j < n; j += LB_T) {                             /* dual_func with m = 0, mma.c:88-124 */
                const double sg = sigma[j], xj = x[j], d = dfdx[j];
                double xc = xj;
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
                    gs += (d * (sigma2 * dx) + (fabs(d) * sg + 0.5 * rho) * dx2) * denominv;
                    ws += 0.5 * dx2 * denominv;
                }
                xcur[j] = xc;
            }
            const double gval = minf + lb_block_sum(gs, S);
            const double wval = lb_block_sum(ws, S);
            __syncthreads();
            fcur = lb_objgrad<OBJ>(n, xcur, dfdx_cur, S, oscratch);           /* mma.c:308 */
            ++nevals; ++inner_nevals; ++fcalls;
            int inner_done = (gval >= fcur) || (P.inner_maxeval > 0 && inner_nevals == P.inner_maxeval);
            if (P.always_improve ? fcur < minf : inner_done) {               /* mma.c:329-331 with feasible = feasible_cur = 1 */
                if (!P.inner_gradients) {
                    ++fcalls;                                                 /* the uncounted call with a gradient, mma.c:336-339 */
                    inner_done = gval >= fcur;                                /* mma.c:343: recomputed WITHOUT the inner_maxeval clause */
                }
                minf = fcur;
                for (int j = tid; j < n; j += LB_T) { x[j] = xcur[j]; dfdx[j] = dfdx_cur[j]; }
            }
            __syncthreads();
            if (P.maxeval > 0 && nevals >= P.maxeval) ret = 5;
            else if (minf < P.minf_max) ret = 2;
            if (ret != 1 || inner_done) break;
            if (fcur > gval) {                                                /* mma.c:394-395 */
                const double r1 = 10 * rho, r2 = 1.1 * (rho + (fcur - gval) / wval);
                rho = r1 < r2 ? r1 : r2;
            }
        }
        if (ret != 1) break;
        {                                                                     /* mma.c:408-411; stop.c:87-120 (no weights, no xtol_abs here) */
            double nx = 0, ndx = 0;
            for (int j = tid; j < n; j += LB_T) { nx += fabs(xcur[j]); ndx += fabs(xcur[j] - xprev[j]); }
            nx = lb_block_sum(nx, S);
            ndx = lb_block_sum(ndx, S);
            if (!isinf(fprev)) {
                const double df = fabs(fcur - fprev);
                if (df < P.ftol_abs || df < P.ftol_rel * (fabs(fcur) + fabs(fprev)) * 0.5 || (P.ftol_rel > 0 && fcur == fprev)) ret = 3;
            }
            if (ndx < P.xtol_rel * nx) ret = 4;
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
    if (tid == 0) { out[inst].f = minf; out[inst].ret = ret; out[inst].nevals = nevals; out[inst].iterm = fcalls; out[inst].cols = k; }
}

extern "C" size_t nla_mma_work_doubles(int ld, int count) { return (size_t) count * 6 * (size_t) ld; }

extern "C" int nla_k_mma_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *sigma_init,
                               double *X, double *work, const nla_mma_params *params, nla_lbfgs_result *out, void *stream)
{
    if (count <= 0) return 0;
    hipStream_t st = (hipStream_t) stream;
    const nla_mma_params P = *params;
#define CALL(O) hipLaunchKernelGGL((mma_batch_kernel<O>), dim3(count), dim3(LB_T), 0, st, n, ld, count, lb, ub, sigma_init, X, work, P, out)
    NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}
