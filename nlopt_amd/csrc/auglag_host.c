/* auglag_host.c — the augmented-Lagrangian wrapper NLOPT_AUGLAG / _EQ, LN_AUGLAG / _EQ, LD_AUGLAG / _EQ (reference:
 * src/algs/auglag/auglag.c:66-291, dispatched at src/api/optimize.c:907-939).
 *
 * AUGLAG is not an optimiser of its own but a CALLER of one: it folds the nonlinear constraints (all of them, or — the _EQ
 * flavours — only the equalities) into a penalised objective and has a subsidiary optimiser minimise that, again and again,
 * updating the multipliers and the penalty strength in between (Birgin & Martinez).  The subsidiary optimiser is any algorithm
 * this library serves — the path's own CRS2_LM / ISRES / MLSL / ESCH for constrained GLOBAL searches, LD_LBFGS, LD_MMA or
 * LN_COBYLA locally — reached through the library's own nlopt_optimize with the penalised objective as an ordinary callback,
 * so everything below is host arithmetic around the caller's callbacks, in the reference's order. */
#include "nla_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    nlopt_func f; void *f_data;
    unsigned nfc, nh;                   /* constraint entries folded into the penalty (inequality, equality) */
    const nla_constraint *fc, *h;
    double rho, *lambda, *mu;           /* penalty strength; multipliers of the equality / inequality rows */
    double *res, *jac;                  /* scratch: one entry's values and gradients */
    const nla_stopping *stop;
} penalised;

static void eval_entry(double *res, double *jac, const nla_constraint *c, unsigned n, const double *x)
{
    if (c->f) res[0] = c->f(n, x, jac, c->f_data);
    else c->mf(c->m, res, n, x, jac, c->f_data);
}

/* L(x) = f + rho/2 * sum (h + lambda/rho)^2 + rho/2 * sum max(0, fc + mu/rho)^2 and its gradient (auglag.c:27-62);
 * every call counts as an evaluation of the run */
static double penalised_objective(unsigned n, const double *x, double *grad, void *data)
{
    penalised *P = (penalised *) data;
    double *jac = grad ? P->jac : NULL;
    const double rho = P->rho;
    unsigned e, k, j, row;
    double L = P->f(n, x, grad, P->f_data);
    ++*(P->stop->nevals_p);
    if (nla_stop_forced(P->stop)) return L;
    for (row = e = 0; e < P->nh; ++e) {
        eval_entry(P->res, jac, P->h + e, n, x);
        if (nla_stop_forced(P->stop)) return L;
        for (k = 0; k < P->h[e].m; ++k) {
            const double hv = P->res[k] + P->lambda[row++] / rho;
            L += 0.5 * rho * hv * hv;
            if (grad) for (j = 0; j < n; ++j) grad[j] += (rho * hv) * jac[k * n + j];
        }
    }
    for (row = e = 0; e < P->nfc; ++e) {
        eval_entry(P->res, jac, P->fc + e, n, x);
        if (nla_stop_forced(P->stop)) return L;
        for (k = 0; k < P->fc[e].m; ++k) {
            const double cv = P->res[k] + P->mu[row++] / rho;
            if (cv > 0) {
                L += 0.5 * rho * cv * cv;
                if (grad) for (j = 0; j < n; ++j) grad[j] += (rho * cv) * jac[k * n + j];
            }
        }
    }
    return L;
}

static unsigned widest(unsigned count, const nla_constraint *c) { unsigned i, w = 0; for (i = 0; i < count; ++i) if (c[i].m > w) w = c[i].m; return w; }
static unsigned rows_of(unsigned count, const nla_constraint *c) { unsigned i, r = 0; for (i = 0; i < count; ++i) r += c[i].m; return r; }
static double dmin(double a, double b) { return a < b ? a : b; }      /* MIN / MAX of auglag.c:14-15: the first operand wins ties and NaN comparisons fall to the second */
static double dmax(double a, double b) { return a > b ? a : b; }

nlopt_result nla_auglag_minimize(unsigned n, nlopt_func f, void *f_data, unsigned m, const nla_constraint *fc, unsigned p, const nla_constraint *h,
                                 const double *lb, const double *ub, double *x, double *minf, nla_stopping *stop, nlopt_opt sub_opt, int sub_has_fc)
{
    const double tau = 0.5, gam = 10, lam_min = -1e20, lam_max = 1e20, mu_max = 1e20;   /* Birgin & Martinez, auglag.c:84-85 */
    penalised P;
    nlopt_result ret;
    double ICM = HUGE_VAL, minf_penalty = HUGE_VAL, penalty, fcur, *buf = NULL, *xcur;
    int feasible, minf_feasible = 0;
    unsigned e, k, row, wide, nrows_fc, nrows_h, handed = m;

    P.f = f; P.f_data = f_data; P.fc = fc; P.h = h; P.nh = p; P.stop = stop;
    if (sub_has_fc) P.nfc = 0;          /* the inequalities go to the subsidiary optimiser as they are ... */
    else { P.nfc = m; handed = 0; }     /* ... or into the penalty */
    wide = widest(P.nfc, fc) > widest(P.nh, h) ? widest(P.nfc, fc) : widest(P.nh, h);
    nrows_fc = rows_of(P.nfc, fc); nrows_h = rows_of(P.nh, h);

    if ((ret = nlopt_set_min_objective(sub_opt, penalised_objective, &P)) < 0) return ret;
    if ((ret = nlopt_set_lower_bounds(sub_opt, lb)) < 0) return ret;
    if ((ret = nlopt_set_upper_bounds(sub_opt, ub)) < 0) return ret;
    if ((ret = nlopt_set_stopval(sub_opt, P.nfc == 0 && P.nh == 0 ? stop->minf_max : -HUGE_VAL)) < 0) return ret;
    if (P.nfc != 0 || P.nh != 0) {      /* the penalised problem needs a convergence criterion of its own, auglag.c:118-123 */
        if (nlopt_get_xtol_rel(sub_opt) <= 0 && nlopt_get_ftol_rel(sub_opt) <= 0)
            nlopt_set_xtol_rel(sub_opt, stop->xtol_rel > 0 ? stop->xtol_rel : 1e-8);
    }
    if ((ret = nlopt_remove_inequality_constraints(sub_opt)) < 0) return ret;
    if ((ret = nlopt_remove_equality_constraints(sub_opt)) < 0) return ret;
    for (e = 0; e < handed; ++e) {
        ret = fc[e].f ? nlopt_add_inequality_constraint(sub_opt, fc[e].f, fc[e].f_data, fc[e].tol[0])
                      : nlopt_add_inequality_mconstraint(sub_opt, fc[e].m, fc[e].mf, fc[e].f_data, fc[e].tol);
        if (ret < 0) return ret;
    }

    buf = (double *) malloc(sizeof(double) * ((size_t) n + (size_t) wide * (1 + (size_t) n) + nrows_h + nrows_fc + 1));
    if (!buf) return NLOPT_OUT_OF_MEMORY;
    xcur = buf;
    memcpy(xcur, x, sizeof(double) * n);
    P.res = xcur + n;
    P.jac = P.res + wide;
    memset(P.jac, 0, sizeof(double) * ((size_t) n * wide + nrows_h + nrows_fc));
    P.lambda = P.jac + (size_t) n * wide;
    P.mu = P.lambda + nrows_h;
    *minf = HUGE_VAL;

    if (P.nh > 0 || P.nfc > 0) {        /* the first penalty strength from the start point, auglag.c:151-186 */
        double con2 = 0;
        ++*(stop->nevals_p);
        fcur = f(n, xcur, NULL, f_data);
        if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
        penalty = 0; feasible = 1;
        for (e = 0; e < P.nh; ++e) {
            eval_entry(P.res, NULL, h + e, n, xcur);
            if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
            for (k = 0; k < h[e].m; ++k) {
                const double hv = P.res[k];
                penalty += fabs(hv);
                feasible = feasible && fabs(hv) <= h[e].tol[k];
                con2 += hv * hv;
            }
        }
        for (e = 0; e < P.nfc; ++e) {
            eval_entry(P.res, NULL, fc + e, n, xcur);
            if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
            for (k = 0; k < fc[e].m; ++k) {
                const double cv = P.res[k];
                penalty += cv > 0 ? cv : 0;
                feasible = feasible && cv <= fc[e].tol[k];
                if (cv > 0) con2 += cv * cv;
            }
        }
        *minf = fcur; minf_penalty = penalty; minf_feasible = feasible;
        P.rho = con2 > 0 ? dmax(1e-6, dmin(10, 2 * fabs(*minf) / con2)) : 10;
    } else P.rho = 1;

    for (;;) {
        const double prev_ICM = ICM;
        ret = nla_optimize_limited(sub_opt, xcur, &fcur, stop->maxeval - *(stop->nevals_p), stop->maxtime - (nla_seconds() - stop->start));
        if (ret < 0) {
            /* the reference reports a failure down there without text (restoring the limits clears it, optimize.c:1109-1110);
             * here a message of the subsidiary run — a device failure, an algorithm this library does not provide — is passed on */
            const char *why = nlopt_get_errmsg(sub_opt);
            if (why) nla_stop_msg(stop, "subsidiary optimiser %s: %s", nlopt_algorithm_to_string(nlopt_get_algorithm(sub_opt)), why);
            break;
        }
        ++*(stop->nevals_p);
        fcur = f(n, xcur, NULL, f_data);
        if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }

        ICM = 0; penalty = 0; feasible = 1;
        for (e = row = 0; e < P.nh; ++e) {                                   /* multiplier updates, auglag.c:213-241 */
            eval_entry(P.res, NULL, h + e, n, xcur);
            if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
            for (k = 0; k < h[e].m; ++k) {
                const double hv = P.res[k], newlam = P.lambda[row] + P.rho * hv;
                penalty += fabs(hv);
                feasible = feasible && fabs(hv) <= h[e].tol[k];
                ICM = dmax(ICM, fabs(hv));
                P.lambda[row++] = dmin(dmax(lam_min, newlam), lam_max);
            }
        }
        for (e = row = 0; e < P.nfc; ++e) {
            eval_entry(P.res, NULL, fc + e, n, xcur);
            if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
            for (k = 0; k < fc[e].m; ++k) {
                const double cv = P.res[k], newmu = P.mu[row] + P.rho * cv;
                penalty += cv > 0 ? cv : 0;
                feasible = feasible && cv <= fc[e].tol[k];
                ICM = dmax(ICM, fabs(dmax(cv, -P.mu[row] / P.rho)));
                P.mu[row++] = dmin(dmax(0.0, newmu), mu_max);
            }
        }
        if (ICM > tau * prev_ICM) P.rho *= gam;

        if ((feasible && (!minf_feasible || penalty < minf_penalty || fcur < *minf)) || (!minf_feasible && penalty < minf_penalty)) {
            ret = NLOPT_SUCCESS;
            if (feasible) {
                if (fcur < stop->minf_max) ret = NLOPT_STOPVAL_REACHED;
                else if (nla_stop_ftol(stop, fcur, *minf)) ret = NLOPT_FTOL_REACHED;
                else if (nla_stop_x(stop, xcur, x)) ret = NLOPT_XTOL_REACHED;
            }
            *minf = fcur; minf_penalty = penalty; minf_feasible = feasible;
            memcpy(x, xcur, sizeof(double) * n);
            if (ret != NLOPT_SUCCESS) break;
        }
        if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; break; }
        if (nla_stop_evals(stop)) { ret = NLOPT_MAXEVAL_REACHED; break; }
        if (nla_stop_time(stop)) { ret = NLOPT_MAXTIME_REACHED; break; }
        if (ICM == 0) { ret = NLOPT_FTOL_REACHED; break; }
    }
done:
    free(buf);
    return ret;
}
