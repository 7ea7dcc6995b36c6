/* port_mma.c — CPU ORACLE (test infrastructure): NLOPT_LD_MMA, Svanberg's method of moving asymptotes
 * in its globally convergent CCSA form (src/algs/mma/mma.c:146-449), for the case the hot path
 * uses it in: the default local optimiser of NLOPT_GD_MLSL(_LDS) (src/api/optimize.c:763-768,
 * deprecated.c:28) — bound constraints only, no nonlinear constraints (m = 0; MLSL strips them,
 * options.c:824-846).
 *
 * With m = 0 the dual problem has no variables: nlopt_optimize on the 0-dimensional dual_opt
 * evaluates dual_func once (optimize.c:533-536) and mma.c:298 evaluates it again — twice the same
 * closed-form minimiser of the separable approximation (mma.c:58-137), written once here.
 *
 * Parameters as the dispatcher reads them (optimize.c:795-834): inner_maxeval (0), rho_init (1),
 * inner_gradients (1), always_improve (1), sigma_min (0); sigma_init = the initial step dx (NULL
 * ⇒ half the box width, 1 for an unbounded coordinate, mma.c:203-211).
 *
 * Evaluation counting: every objective call goes through `f` (MLSL counts them in its own
 * wrapper); stop->nevals counts what mma.c counts — with inner_gradients = 0 the re-evaluation
 * with a gradient (mma.c:336-338) is a call that is NOT counted.
 */
#include "port_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MMA_RHOMIN 1e-5                                                     /* mma.c:40 */

typedef struct { double gval, wval; } approx;

/* the minimiser xcur of the separable approximation around x and the approximation's value there
 * (dual_func with m = 0, mma.c:58-137) */
static approx separable_step(int n, const double *x, const double *lb, const double *ub, const double *sigma,
                             const double *dfdx, double fval, double rho, double *xcur)
{
    approx a;
    a.gval = fval; a.wval = 0;
    for (int j = 0; j < n; ++j) {
        double u, v, dx, dx2, sigma2, denominv, c;
        if (sigma[j] == 0) { xcur[j] = x[j]; continue; }                    /* mma.c:91-94 */
        u = dfdx[j];
        v = fabs(dfdx[j]) * sigma[j] + 0.5 * rho;
        sigma2 = sigma[j] * sigma[j];
        u *= sigma2;
        {
            const double q = u / (v * sigma[j]);
            dx = (u / v) / (-1 - sqrt(fabs(1 - q * q)));                    /* mma.c:103 */
        }
        xcur[j] = x[j] + dx;
        if (xcur[j] > ub[j]) xcur[j] = ub[j];
        else if (xcur[j] < lb[j]) xcur[j] = lb[j];
        if (xcur[j] > x[j] + 0.9 * sigma[j]) xcur[j] = x[j] + 0.9 * sigma[j];
        else if (xcur[j] < x[j] - 0.9 * sigma[j]) xcur[j] = x[j] - 0.9 * sigma[j];
        dx = xcur[j] - x[j];
        dx2 = dx * dx;
        denominv = 1.0 / (sigma2 - dx2);
        c = sigma2 * dx;
        a.gval += (dfdx[j] * c + (fabs(dfdx[j]) * sigma[j] + 0.5 * rho) * dx2) * denominv;   /* mma.c:119-120 */
        a.wval += 0.5 * dx2 * denominv;
    }
    return a;
}

int orc_mma_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                     orc_stop *stop, const orc_mma_params *prm)
{
    int ret = ORC_SUCCESS, k = 0, j;
    double *sigma = (double *) malloc(sizeof(double) * 6 * (size_t) n);
    double *dfdx = sigma + n, *dfdx_cur = dfdx + n, *xcur = dfdx_cur + n, *xprev = xcur + n, *xprevprev = xprev + n;
    double rho, fcur;
    if (!sigma) return ORC_OUT_OF_MEMORY;
    for (j = 0; j < n; ++j) {                                               /* mma.c:203-211 */
        if (prm->sigma_init && prm->sigma_init[j] > 0) sigma[j] = prm->sigma_init[j];
        else if (isinf(ub[j]) || isinf(lb[j])) sigma[j] = 1.0;
        else sigma[j] = 0.5 * (ub[j] - lb[j]);
        sigma[j] = sigma[j] > prm->sigma_min ? sigma[j] : prm->sigma_min;
    }
    rho = prm->rho_init;
    fcur = *minf = f((unsigned) n, x, dfdx, f_data);                        /* mma.c:219-221 */
    ++stop->nevals;
    memcpy(xcur, x, sizeof(double) * (size_t) n);
    if (stop->force_stop) { ret = ORC_FORCED_STOP; goto done; }

    for (;;) {                                                              /* outer iterations, mma.c:253 */
        long inner_nevals = 0;
        const double fprev = fcur;
        if (stop->force_stop) ret = ORC_FORCED_STOP;
        else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED;
        else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED;
        else if (*minf < stop->minf_max) ret = ORC_STOPVAL_REACHED;
        if (ret != ORC_SUCCESS) goto done;
        if (++k > 1) memcpy(xprevprev, xprev, sizeof(double) * (size_t) n);
        memcpy(xprev, xcur, sizeof(double) * (size_t) n);

        for (;;) {                                                          /* inner iterations, mma.c:265 */
            int inner_done;
            const approx a = separable_step(n, x, lb, ub, sigma, dfdx, *minf, rho, xcur);
            fcur = f((unsigned) n, xcur, prm->inner_gradients ? dfdx_cur : NULL, f_data);
            ++stop->nevals;
            ++inner_nevals;
            if (stop->force_stop) { ret = ORC_FORCED_STOP; goto done; }
            inner_done = a.gval >= fcur;
            inner_done = inner_done || (prm->inner_maxeval > 0 && inner_nevals == prm->inner_maxeval);
            if (prm->always_improve ? fcur < *minf : inner_done) {         /* mma.c:329-331 with feasible = feasible_cur = 1 */
                if (!prm->inner_gradients) {
                    fcur = f((unsigned) n, xcur, dfdx_cur, f_data);         /* not counted, mma.c:336-339 */
                    if (stop->force_stop) { ret = ORC_FORCED_STOP; goto done; }
                    inner_done = a.gval >= fcur;                           /* mma.c:343 — WITHOUT the inner_maxeval clause of :324: a
                                                                              step that only ended because the inner limit was hit goes on */
                }
                *minf = fcur;
                memcpy(x, xcur, sizeof(double) * (size_t) n);
                memcpy(dfdx, dfdx_cur, sizeof(double) * (size_t) n);
            }
            if (stop->force_stop) ret = ORC_FORCED_STOP;
            else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED;
            else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED;
            else if (*minf < stop->minf_max) ret = ORC_STOPVAL_REACHED;
            if (ret != ORC_SUCCESS) goto done;
            if (inner_done) break;
            if (fcur > a.gval) {                                            /* mma.c:394-395 */
                const double r1 = 10 * rho, r2 = 1.1 * (rho + (fcur - a.gval) / a.wval);
                rho = r1 < r2 ? r1 : r2;
            }
        }
        if (orc_stop_ftol(stop, fcur, fprev)) ret = ORC_FTOL_REACHED;       /* mma.c:408-411 */
        if (orc_stop_x(stop, xcur, xprev)) ret = ORC_XTOL_REACHED;
        if (ret != ORC_SUCCESS) goto done;
        rho = 0.1 * rho > MMA_RHOMIN ? 0.1 * rho : MMA_RHOMIN;              /* mma.c:415 */
        if (k > 1)
            for (j = 0; j < n; ++j) {                                       /* mma.c:423-435 */
                const double dx2 = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j]);
                const double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1);
                sigma[j] *= gam;
                if (!isinf(ub[j]) && !isinf(lb[j])) {
                    const double hi = 10 * (ub[j] - lb[j]), lo = 0.01 * (ub[j] - lb[j]);
                    sigma[j] = sigma[j] < hi ? sigma[j] : hi;
                    sigma[j] = sigma[j] > lo ? sigma[j] : lo;
                }
                sigma[j] = sigma[j] > prm->sigma_min ? sigma[j] : prm->sigma_min;
            }
    }
done:
    free(sigma);
    return ret;
}
