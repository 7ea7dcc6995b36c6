/* mma_host.c — NLOPT_LD_MMA with nonlinear inequality constraints (reference: src/algs/mma/mma.c:146-452, dispatched at
 * src/api/optimize.c:795-834).
 *
 * Without constraints the whole search is one device kernel (hip/mma_kernels.hip): the dual problem then has no variables and
 * the conservative-approximation step is a closed form per coordinate.  With m constraint rows the step is the minimiser of a
 * separable convex model for given multipliers y >= 0, and y maximises the dual function — an m-dimensional bound-constrained
 * problem the reference hands to a nested optimiser (by default LD_MMA again, now without constraints).  This file is that outer
 * algorithm on the host: the model (dual_value), the outer / inner iterations with their conservativeness test, the rho / sigma
 * updates, and the dual solve through the library's own nlopt_optimize — i.e. through the device kernel in its coroutine mode
 * with dual_value as the callback, in the reference's summation order (nla_exact_mode_for), so that the multipliers, and with
 * them every trial point, are the reference's bit for bit.  It is the caller's objective and constraints that cost here, one
 * call per inner iteration; the dual solves are m-dimensional.
 *
 * Kept from the reference: a constraint value of NaN switches that row off (mma.c:141-143); an infeasible start bounds the
 * multipliers by 1e40 until a feasible point is reached (mma.c:231-244); "inner_gradients" = 0 re-evaluates accepted points
 * with gradients without counting the call (mma.c:336-339).  Not kept: the "verbosity" printouts. */
#include "nla_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NLA_MMA_RHOMIN 1e-5            /* mma.c:41 */

typedef struct {
    unsigned n, m;
    const double *x, *lb, *ub;          /* centre of the model, box */
    double *sigma, *dfdx, *dfcdx;       /* asymptote widths, gradients of f and of the m rows at x (row-major m x n) */
    double *fcval, *rhoc;               /* constraint values at x (NaN = row inactive), their conservativeness parameters */
    double fval, rho;
    double *xcur, *gcval;               /* out: the model's minimiser for the last y, the rows' approximants there */
    double gval, wval;                  /* out: the objective's approximant there, the curvature weight */
    int count;
} mma_model;

static int row_active(const mma_model *M, unsigned i) { return !isnan(M->fcval[i]); }

/* minus the dual function at y and, if asked, its gradient (mma.c:58-137): the model is separable, so its minimiser over x is
 * found coordinate by coordinate in closed form.  Leaves xcur, gval, wval, gcval of that minimiser in the model. */
static double dual_value(unsigned m, const double *y, double *grad, void *model)
{
    mma_model *M = (mma_model *) model;
    const unsigned n = M->n;
    unsigned i, j;
    double val;
    M->count++;
    val = M->gval = M->fval;
    M->wval = 0;
    for (i = 0; i < m; ++i) val += y[i] * (M->gcval[i] = row_active(M, i) ? M->fcval[i] : 0);
    for (j = 0; j < n; ++j) {
        const double sg = M->sigma[j];
        double u, v, dx, dx2, sigma2, denominv, c;
        if (sg == 0) { M->xcur[j] = M->x[j]; continue; }                      /* lb == ub */
        u = M->dfdx[j];
        v = fabs(M->dfdx[j]) * sg + 0.5 * M->rho;
        for (i = 0; i < m; ++i) if (row_active(M, i)) {
            u += M->dfcdx[i * n + j] * y[i];
            v += (fabs(M->dfcdx[i * n + j]) * sg + 0.5 * M->rhoc[i]) * y[i];
        }
        u *= (sigma2 = sg * sg);
        { const double q = u / (v * sg); dx = (u / v) / (-1 - sqrt(fabs(1 - q * q))); }
        M->xcur[j] = M->x[j] + dx;
        if (M->xcur[j] > M->ub[j]) M->xcur[j] = M->ub[j];
        else if (M->xcur[j] < M->lb[j]) M->xcur[j] = M->lb[j];
        if (M->xcur[j] > M->x[j] + 0.9 * sg) M->xcur[j] = M->x[j] + 0.9 * sg;
        else if (M->xcur[j] < M->x[j] - 0.9 * sg) M->xcur[j] = M->x[j] - 0.9 * sg;
        dx = M->xcur[j] - M->x[j];
        dx2 = dx * dx;
        denominv = 1.0 / (sigma2 - dx2);
        val += (u * dx + v * dx2) * denominv;
        c = sigma2 * dx;
        M->gval += (M->dfdx[j] * c + (fabs(M->dfdx[j]) * sg + 0.5 * M->rho) * dx2) * denominv;
        M->wval += 0.5 * dx2 * denominv;
        for (i = 0; i < m; ++i) if (row_active(M, i))
            M->gcval[i] += (M->dfcdx[i * n + j] * c + (fabs(M->dfcdx[i * n + j]) * sg + 0.5 * M->rhoc[i]) * dx2) * denominv;
    }
    if (grad) for (i = 0; i < m; ++i) grad[i] = -M->gcval[i];
    return -val;
}

/* values (and gradients, if grad != NULL) of every constraint row at x, in order; 1 if a callback forced a stop */
static int eval_rows(unsigned nfc, const nla_constraint *fc, unsigned n, const double *x, double *val, double *grad, const nla_stopping *stop)
{
    unsigned i = 0, k;
    for (k = 0; k < nfc; ++k) {
        double *g = grad ? grad + (size_t) i * n : NULL;
        if (fc[k].f) val[i] = fc[k].f(n, x, g, fc[k].f_data);
        else fc[k].mf(fc[k].m, val + i, n, x, g, fc[k].f_data);
        i += fc[k].m;
        if (nla_stop_forced(stop)) return 1;
    }
    return 0;
}

static nlopt_result limits_hit(const nla_stopping *stop, int feasible, double minf)
{
    if (nla_stop_forced(stop)) return NLOPT_FORCED_STOP;
    if (nla_stop_evals(stop)) return NLOPT_MAXEVAL_REACHED;
    if (nla_stop_time(stop)) return NLOPT_MAXTIME_REACHED;
    if (feasible && minf < stop->minf_max) return NLOPT_STOPVAL_REACHED;
    return NLOPT_SUCCESS;
}

nlopt_result nla_mma_constrained(nlopt_opt opt, unsigned n, nlopt_func f, void *f_data, const double *lb, const double *ub,
                                 double *x, double *minf, nla_stopping *stop, const nla_mma_params *prm)
{
    const unsigned nfc = opt->m;
    const nla_constraint *fc = opt->fc;
    const int inner_gradients = prm->inner_gradients, always_improve = prm->always_improve, inner_maxeval = prm->inner_maxeval;
    nlopt_result ret = NLOPT_SUCCESS;
    nlopt_opt dual_opt;
    mma_model M;
    unsigned m = 0, i, j, k = 0;
    double *buf, *sigma, *dfdx, *dfdx_cur, *xcur, *xprev, *xprevprev, *fcval, *fcval_cur, *rhoc, *gcval, *dual_lb, *dual_ub, *y, *dfcdx, *dfcdx_cur;
    double rho, fcur, infeasibility = 0;
    int feasible = 1;

    for (i = 0; i < nfc; ++i) m += fc[i].m;                                   /* nlopt_count_constraints */
    {   /* the nested optimiser of the dual problem as the dispatcher sets it up (optimize.c:817-827) */
        const nlopt_opt lo = opt->local_opt;
        nlopt_algorithm dual_alg = (nlopt_algorithm) nlopt_get_param(opt, "dual_algorithm", lo ? (double) lo->algorithm : (double) nla_local_search_alg_deriv);
        dual_opt = nlopt_create(dual_alg, m);
        if (!dual_opt) { nla_stop_msg(stop, "failed creating dual optimizer"); return NLOPT_FAILURE; }
        nlopt_set_ftol_rel(dual_opt, nlopt_get_param(opt, "dual_ftol_rel", lo ? lo->ftol_rel : 1e-14));
        nlopt_set_ftol_abs(dual_opt, nlopt_get_param(opt, "dual_ftol_abs", lo ? lo->ftol_abs : 0.0));
        nlopt_set_xtol_rel(dual_opt, nlopt_get_param(opt, "dual_xtol_rel", 0.0));
        nlopt_set_xtol_abs1(dual_opt, nlopt_get_param(opt, "dual_xtol_abs", 0.0));
        nlopt_set_maxeval(dual_opt, (int) nlopt_get_param(opt, "dual_maxeval", lo ? (double) lo->maxeval : 100000.));
    }
    buf = (double *) malloc(sizeof(double) * (6 * (size_t) n + 2 * (size_t) m * n + 7 * (size_t) m + 1));
    if (!buf) { nlopt_destroy(dual_opt); return NLOPT_OUT_OF_MEMORY; }
    sigma = buf; dfdx = sigma + n; dfdx_cur = dfdx + n; xcur = dfdx_cur + n; xprev = xcur + n; xprevprev = xprev + n;
    fcval = xprevprev + n; fcval_cur = fcval + m; rhoc = fcval_cur + m; gcval = rhoc + m; dual_lb = gcval + m; dual_ub = dual_lb + m;
    y = dual_ub + m; dfcdx = y + m; dfcdx_cur = dfcdx + (size_t) m * n;

    memset(&M, 0, sizeof M);
    M.n = n; M.m = m; M.x = x; M.lb = lb; M.ub = ub; M.sigma = sigma; M.dfdx = dfdx; M.dfcdx = dfcdx; M.fcval = fcval; M.rhoc = rhoc;
    M.xcur = xcur; M.gcval = gcval;

    for (j = 0; j < n; ++j) {                                                 /* mma.c:206-214 */
        if (opt->dx && opt->dx[j] > 0) sigma[j] = opt->dx[j];
        else if (nla_isinf(ub[j]) || nla_isinf(lb[j])) sigma[j] = 1.0;
        else sigma[j] = 0.5 * (ub[j] - lb[j]);
        sigma[j] = sigma[j] > prm->sigma_min ? sigma[j] : prm->sigma_min;
    }
    rho = prm->rho_init;
    for (i = 0; i < m; ++i) { rhoc[i] = prm->rho_init; dual_lb[i] = y[i] = 0.0; dual_ub[i] = HUGE_VAL; }

    M.fval = fcur = *minf = f(n, x, dfdx, f_data);
    ++*(stop->nevals_p);
    memcpy(xcur, x, sizeof(double) * n);
    if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
    if (eval_rows(nfc, fc, n, x, fcval, dfcdx, stop)) { ret = NLOPT_FORCED_STOP; goto done; }
    for (i = 0; i < m; ++i) {
        feasible = feasible && (fcval[i] <= 0 || isnan(fcval[i]));
        if (fcval[i] > infeasibility) infeasibility = fcval[i];
    }
    if (!feasible) for (i = 0; i < m; ++i) dual_ub[i] = 1e40;                 /* mma.c:231-244 */

    nlopt_set_min_objective(dual_opt, dual_value, &M);
    nlopt_set_lower_bounds(dual_opt, dual_lb);
    nlopt_set_upper_bounds(dual_opt, dual_ub);
    nlopt_set_stopval(dual_opt, -HUGE_VAL);
    nlopt_remove_inequality_constraints(dual_opt);
    nlopt_remove_equality_constraints(dual_opt);

    for (;;) {                                                                /* outer iterations, mma.c:253 */
        int inner_nevals = 0;
        const double fprev = fcur;
        if ((ret = limits_hit(stop, feasible, *minf)) != NLOPT_SUCCESS) goto done;
        if (++k > 1) memcpy(xprevprev, xprev, sizeof(double) * n);
        memcpy(xprev, xcur, sizeof(double) * n);

        for (;;) {                                                            /* inner iterations, mma.c:265 */
            double min_dual, infeasibility_cur = 0;
            int feasible_cur = 1, inner_done, new_infeasible_constraint = 0;
            nlopt_result reti;
            unsigned k0;

            M.rho = rho; M.count = 0;
            reti = nla_optimize_limited(dual_opt, y, &min_dual, 0, stop->maxtime - (nla_seconds() - stop->start));
            if (reti < 0 || reti == NLOPT_MAXTIME_REACHED) {
                /* the reference reports a failure down there without text (restoring the limits clears it, optimize.c:1109-1110);
                 * here a message of the nested run — a device failure, a "dual_algorithm" this library does not provide — is passed on */
                const char *why = nlopt_get_errmsg(dual_opt);
                if (reti < 0 && why) nla_stop_msg(stop, "dual problem (%s): %s", nlopt_algorithm_to_string(nlopt_get_algorithm(dual_opt)), why);
                ret = reti;
                goto done;
            }
            dual_value(m, y, NULL, &M);                                       /* the step for the final multipliers */

            fcur = f(n, xcur, inner_gradients ? dfdx_cur : NULL, f_data);
            ++*(stop->nevals_p);
            ++inner_nevals;
            if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
            inner_done = M.gval >= fcur;
            if (eval_rows(nfc, fc, n, xcur, fcval_cur, inner_gradients ? dfcdx_cur : NULL, stop)) { ret = NLOPT_FORCED_STOP; goto done; }
            for (i = 0, k0 = 0; k0 < nfc; ++k0) {
                const unsigned i0 = i, inext = i + fc[k0].m;
                for (; i < inext; ++i) if (!isnan(fcval_cur[i])) {
                    feasible_cur = feasible_cur && (fcval_cur[i] <= fc[k0].tol[i - i0]);
                    if (!isnan(fcval[i])) inner_done = inner_done && (M.gcval[i] >= fcval_cur[i]);
                    else if (fcval_cur[i] > 0) new_infeasible_constraint = 1;
                    if (fcval_cur[i] > infeasibility_cur) infeasibility_cur = fcval_cur[i];
                }
            }
            inner_done = inner_done || (inner_maxeval > 0 && inner_nevals == inner_maxeval);

            if (always_improve ? ((fcur < *minf && (inner_done || feasible_cur || !feasible)) || (!feasible && infeasibility_cur < infeasibility))
                               : inner_done) {                               /* move the model's centre, mma.c:329-385 */
                if (!inner_gradients) {                                       /* again, now with gradients; not counted */
                    fcur = f(n, xcur, dfdx_cur, f_data);
                    if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
                    feasible_cur = 1; infeasibility_cur = 0; new_infeasible_constraint = 0;
                    inner_done = M.gval >= fcur;
                    if (eval_rows(nfc, fc, n, xcur, fcval_cur, dfcdx_cur, stop)) { ret = NLOPT_FORCED_STOP; goto done; }
                    for (i = 0, k0 = 0; k0 < nfc; ++k0) {
                        const unsigned i0 = i, inext = i + fc[k0].m;
                        for (; i < inext; ++i) if (!isnan(fcval_cur[i])) {
                            feasible_cur = feasible_cur && (fcval_cur[i] <= fc[k0].tol[i - i0]);
                            if (isnan(fcval[i]) && fcval_cur[i] > 0) new_infeasible_constraint = 1;
                            if (fcval_cur[i] > infeasibility_cur) infeasibility_cur = fcval_cur[i];
                        }
                    }
                }
                M.fval = *minf = fcur;
                infeasibility = infeasibility_cur;
                memcpy(fcval, fcval_cur, sizeof(double) * m);
                memcpy(x, xcur, sizeof(double) * n);
                memcpy(dfdx, dfdx_cur, sizeof(double) * n);
                memcpy(dfcdx, dfcdx_cur, sizeof(double) * (size_t) n * m);
                if (infeasibility_cur == 0) {
                    if (!feasible) {                                          /* feasible from now on: unbounded multipliers again */
                        for (i = 0; i < m; ++i) dual_ub[i] = HUGE_VAL;
                        nlopt_set_upper_bounds(dual_opt, dual_ub);
                    }
                    feasible = 1;
                } else if (new_infeasible_constraint) feasible = 0;
            }
            if ((ret = limits_hit(stop, feasible, *minf)) != NLOPT_SUCCESS) goto done;
            if (inner_done) break;

            if (fcur > M.gval) {                                              /* not conservative: stiffen the model, mma.c:394-401 */
                const double a = 10 * rho, b = 1.1 * (rho + (fcur - M.gval) / M.wval);
                rho = a < b ? a : b;
            }
            for (i = 0; i < m; ++i) if (!isnan(fcval_cur[i]) && fcval_cur[i] > M.gcval[i]) {
                const double a = 10 * rhoc[i], b = 1.1 * (rhoc[i] + (fcval_cur[i] - M.gcval[i]) / M.wval);
                rhoc[i] = a < b ? a : b;
            }
        }

        if (nla_stop_ftol(stop, fcur, fprev)) ret = NLOPT_FTOL_REACHED;
        if (nla_stop_x(stop, xcur, xprev)) ret = NLOPT_XTOL_REACHED;
        if (ret != NLOPT_SUCCESS) goto done;

        rho = 0.1 * rho > NLA_MMA_RHOMIN ? 0.1 * rho : NLA_MMA_RHOMIN;        /* mma.c:415-420 */
        for (i = 0; i < m; ++i) rhoc[i] = 0.1 * rhoc[i] > NLA_MMA_RHOMIN ? 0.1 * rhoc[i] : NLA_MMA_RHOMIN;
        if (k > 1)
            for (j = 0; j < n; ++j) {                                         /* mma.c:423-436 */
                const double dx2 = (xcur[j] - xprev[j]) * (xprev[j] - xprevprev[j]);
                const double gam = dx2 < 0 ? 0.7 : (dx2 > 0 ? 1.2 : 1);
                sigma[j] *= gam;
                if (!nla_isinf(ub[j]) && !nla_isinf(lb[j])) {
                    const double hi = 10 * (ub[j] - lb[j]), lo = 0.01 * (ub[j] - lb[j]);
                    sigma[j] = sigma[j] < hi ? sigma[j] : hi;
                    sigma[j] = sigma[j] > lo ? sigma[j] : lo;
                }
                sigma[j] = sigma[j] > prm->sigma_min ? sigma[j] : prm->sigma_min;
            }
    }

done:
    free(buf);
    nlopt_destroy(dual_opt);
    return ret;
}
