/* api_optimize.c — nlopt_optimize and the dispatcher for the algorithms this library provides.
 * Follows the control flow of the reference's src/api/optimize.c: nlopt_optimize :991-1083
 * (force_stop reset, maximisation by sign flip), nlopt_optimize_ :514-959 (n == 0 shortcut, RNG
 * seeding, bounds check, nlopt_stopping setup, switch on the algorithm) and
 * nlopt_optimize_limited :1087-1113.  Cases outside the stochastic-global hot path are not
 * provided and say so in errmsg.  Coordinates with lb[i] == ub[i] are eliminated in front of CRS2_LM / ISRES / ESCH
 * as the reference does (elimdim, :219-445,1038-1060): the algorithm runs on the reduced problem
 * (its population default, stream consumption and results are those of the reduced dimension) through a wrapper
 * objective — a host function, so such runs take the host-callback path even for a registered device objective. */
#include "nla_internal.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define POP(opt, dflt) ((opt)->stochastic_population > 0 ? (int) (opt)->stochastic_population : \
                        (nla_stochastic_population > 0 ? nla_stochastic_population : (dflt)))   /* optimize.c:511 */

static int finite_domain(unsigned n, const double *lb, const double *ub)     /* optimize.c:187-194 */
{
    unsigned i;
    for (i = 0; i < n; ++i) if (nla_isinf(ub[i] - lb[i])) return 0;
    return 1;
}

typedef struct { nlopt_func f; nlopt_precond pre; void *f_data; } flip_data;
static double flipped_objective(unsigned n, const double *x, double *grad, void *data)   /* optimize.c:970-980 */
{
    flip_data *d = (flip_data *) data;
    double v = d->f(n, x, grad, d->f_data);
    if (grad) { unsigned i; for (i = 0; i < n; ++i) grad[i] = -grad[i]; }
    return -v;
}
static void flipped_precond(unsigned n, const double *x, const double *v, double *vpre, void *data)
{
    flip_data *d = (flip_data *) data;
    unsigned i;
    d->pre(n, x, v, vpre, d->f_data);
    for (i = 0; i < n; ++i) vpre[i] = -vpre[i];
}

/* common prologue of a run: argument and bounds checks, RNG seeding, nlopt_stopping setup
 * (optimize.c:514-566).  Returns NLOPT_SUCCESS when the algorithm may start. */
nlopt_result nla_setup_run(nlopt_opt opt, double *x, double *minf, nla_stopping *stop)
{
    unsigned n, i;
    if (!opt || !x || !minf || !opt->f || opt->maximize) { if (opt) nla_set_errmsg(opt, "NULL args to nlopt_optimize_"); return NLOPT_INVALID_ARGS; }
    n = opt->n;
    *minf = HUGE_VAL;
    nla_srand_time_default();                                                            /* optimize.c:544 */
    for (i = 0; i < n; ++i)
        if (opt->lb[i] > opt->ub[i] || x[i] < opt->lb[i] || x[i] > opt->ub[i]) {
            nla_set_errmsg(opt, "bounds %d fail %g <= %g <= %g", i, opt->lb[i], x[i], opt->ub[i]);
            return NLOPT_INVALID_ARGS;
        }
    stop->n = n;
    stop->minf_max = opt->stopval;
    stop->ftol_rel = opt->ftol_rel; stop->ftol_abs = opt->ftol_abs;
    stop->xtol_rel = opt->xtol_rel; stop->xtol_abs = opt->xtol_abs; stop->x_weights = opt->x_weights;
    opt->numevals = 0;
    stop->nevals_p = &opt->numevals;
    stop->maxeval = opt->maxeval; stop->maxtime = opt->maxtime;
    stop->start = nla_seconds();
    stop->force_stop = &opt->force_stop;
    stop->stop_msg = &opt->errmsg;
    opt->trace_len = 0;
    memset(&opt->stats, 0, sizeof opt->stats);
    return NLOPT_SUCCESS;
}

static nlopt_result minimize_dispatch(nlopt_opt opt, double *x, double *minf)
{
    nla_stopping stop;
    unsigned n;
    nlopt_result pre;
    if (!opt || !x || !minf || !opt->f || opt->maximize) { if (opt) nla_set_errmsg(opt, "NULL args to nlopt_optimize_"); return NLOPT_INVALID_ARGS; }
    n = opt->n;
    if (n == 0) { *minf = opt->f(n, x, NULL, opt->f_data); return NLOPT_SUCCESS; }    /* optimize.c:536-539 */
    if ((pre = nla_setup_run(opt, x, minf, &stop)) != NLOPT_SUCCESS) return pre;

    switch (opt->algorithm) {
    case NLOPT_GN_CRS2_LM:                                                               /* optimize.c:744-747 */
        if (!finite_domain(n, opt->lb, opt->ub)) { nla_set_errmsg(opt, "finite domain required for global algorithm"); return NLOPT_INVALID_ARGS; }
        return nla_crs_minimize(opt, (int) n, opt->f, opt->f_data, opt->lb, opt->ub, x, minf, &stop, POP(opt, 0));
    case NLOPT_LD_LBFGS:                                                                 /* optimize.c:716-718 */
        return nla_lbfgs_minimize(opt, (int) n, opt->f, opt->f_data, opt->lb, opt->ub, x, minf, &stop, (int) opt->vector_storage,
                                  nlopt_get_param(opt, "tolg", 0.));
    case NLOPT_LD_MMA:                                                                   /* optimize.c:795-834 */
        return nla_mma_minimize(opt, (int) n, opt->f, opt->f_data, opt->lb, opt->ub, x, minf, &stop);
    case NLOPT_LN_COBYLA: {                                                              /* optimize.c:836-851: a host algorithm (cobyla_host.c) */
        nlopt_result ret;
        int freedx = 0;
        /* a host algorithm, here for GN_MLSL's sake — but this library is not a CPU NLopt: like every other entry it serves a machine
         * with a HIP device only */
        if (nla_dev_count() <= 0) { nla_set_errmsg(opt, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
        if (!opt->dx) {
            freedx = 1;
            if (nlopt_set_default_initial_step(opt, x) != NLOPT_SUCCESS) { nla_set_errmsg(opt, "failed to allocate initial step"); return NLOPT_OUT_OF_MEMORY; }
        }
        ret = nla_cobyla_minimize(n, opt->f, opt->f_data, opt->m, opt->fc, opt->p, opt->h, opt->lb, opt->ub, x, minf, &stop, opt->dx);
        if (freedx) { free(opt->dx); opt->dx = NULL; }
        return ret;
    }
    case NLOPT_G_MLSL: case NLOPT_G_MLSL_LDS: case NLOPT_GN_MLSL: case NLOPT_GD_MLSL:
    case NLOPT_GN_MLSL_LDS: case NLOPT_GD_MLSL_LDS: {                                    /* optimize.c:748-793 */
        nlopt_opt local_opt = opt->local_opt;
        nlopt_algorithm alg = opt->algorithm;
        nlopt_result ret;
        unsigned i;
        int own = 0;
        if (!finite_domain(n, opt->lb, opt->ub)) { nla_set_errmsg(opt, "finite domain required for global algorithm"); return NLOPT_INVALID_ARGS; }
        if (!local_opt && (alg == NLOPT_G_MLSL || alg == NLOPT_G_MLSL_LDS)) { nla_set_errmsg(opt, "local optimizer must be specified for G_MLSL"); return NLOPT_INVALID_ARGS; }
        if (!local_opt) {                                                               /* the reference's default local optimiser */
            nlopt_algorithm local_alg = (alg == NLOPT_GN_MLSL || alg == NLOPT_GN_MLSL_LDS) ? nla_local_search_alg_nonderiv : nla_local_search_alg_deriv;
            if (local_alg >= NLOPT_GN_MLSL && local_alg <= NLOPT_GD_MLSL_LDS)
                local_alg = (alg == NLOPT_GN_MLSL || alg == NLOPT_GN_MLSL_LDS) ? NLOPT_LN_COBYLA : NLOPT_LD_MMA;
            local_opt = nlopt_create(local_alg, n);
            if (!local_opt) { nla_set_errmsg(opt, "failed to create local_opt"); return NLOPT_FAILURE; }
            own = 1;
            nlopt_set_ftol_rel(local_opt, opt->ftol_rel); nlopt_set_ftol_abs(local_opt, opt->ftol_abs);
            nlopt_set_xtol_rel(local_opt, opt->xtol_rel); nlopt_set_xtol_abs(local_opt, opt->xtol_abs);
            nlopt_set_maxeval(local_opt, nla_local_search_maxeval);
        }
        if (opt->dx) nlopt_set_initial_step(local_opt, opt->dx);                        /* optimize.c:778-779 */
        for (i = 0; i < n && stop.xtol_abs && stop.xtol_abs[i] > 0; ++i) { }
        if (local_opt->ftol_rel <= 0 && local_opt->ftol_abs <= 0 && local_opt->xtol_rel <= 0 && i < n) {
            nlopt_set_ftol_rel(local_opt, 1e-15);                                        /* optimize.c:781-786 */
            nlopt_set_xtol_rel(local_opt, 1e-7);
        }
        opt->force_stop_child = local_opt;
        ret = nla_mlsl_minimize(opt, (int) n, opt->f, opt->f_data, opt->lb, opt->ub, x, minf, &stop, local_opt, POP(opt, 0),
                                alg >= NLOPT_GN_MLSL_LDS && alg != NLOPT_G_MLSL);
        opt->force_stop_child = NULL;
        if (own) nlopt_destroy(local_opt);
        return ret;
    }
    case NLOPT_AUGLAG: case NLOPT_AUGLAG_EQ: case NLOPT_LN_AUGLAG: case NLOPT_LN_AUGLAG_EQ:
    case NLOPT_LD_AUGLAG: case NLOPT_LD_AUGLAG_EQ: {                                     /* optimize.c:907-939: a caller of the path (auglag_host.c) */
        nlopt_opt local_opt = opt->local_opt;
        const nlopt_algorithm alg = opt->algorithm;
        nlopt_result ret;
        int own = 0;
        if ((alg == NLOPT_AUGLAG || alg == NLOPT_AUGLAG_EQ) && !local_opt) { nla_set_errmsg(opt, "local optimizer must be specified for AUGLAG"); return NLOPT_INVALID_ARGS; }
        if (!local_opt) {
            local_opt = nlopt_create(alg == NLOPT_LN_AUGLAG || alg == NLOPT_LN_AUGLAG_EQ ? nla_local_search_alg_nonderiv : nla_local_search_alg_deriv, n);
            if (!local_opt) { nla_set_errmsg(opt, "failed to create local_opt"); return NLOPT_FAILURE; }
            own = 1;
            nlopt_set_ftol_rel(local_opt, opt->ftol_rel); nlopt_set_ftol_abs(local_opt, opt->ftol_abs);
            nlopt_set_xtol_rel(local_opt, opt->xtol_rel); nlopt_set_xtol_abs(local_opt, opt->xtol_abs);
            nlopt_set_maxeval(local_opt, nla_local_search_maxeval);
        }
        if (opt->dx) nlopt_set_initial_step(local_opt, opt->dx);
        opt->force_stop_child = local_opt;
        ret = nla_auglag_minimize(n, opt->f, opt->f_data, opt->m, opt->fc, opt->p, opt->h, opt->lb, opt->ub, x, minf, &stop, local_opt,
                                  alg == NLOPT_AUGLAG_EQ || alg == NLOPT_LN_AUGLAG_EQ || alg == NLOPT_LD_AUGLAG_EQ);
        opt->force_stop_child = NULL;
        if (own) nlopt_destroy(local_opt);
        return ret;
    }
    case NLOPT_GN_ISRES:                                                                 /* optimize.c:941-944 */
        if (!finite_domain(n, opt->lb, opt->ub)) { nla_set_errmsg(opt, "finite domain required for global algorithm"); return NLOPT_INVALID_ARGS; }
        return nla_isres_minimize(opt, (int) n, opt->f, opt->f_data, (int) opt->m, opt->fc, (int) opt->p, opt->h, opt->lb, opt->ub,
                                  x, minf, &stop, POP(opt, 0));
    case NLOPT_GN_ESCH:                                                                  /* optimize.c:946-949 */
        if (!finite_domain(n, opt->lb, opt->ub)) { nla_set_errmsg(opt, "finite domain required for global algorithm"); return NLOPT_INVALID_ARGS; }
        return nla_esch_minimize(opt, (int) n, opt->f, opt->f_data, opt->lb, opt->ub, x, minf, &stop, (unsigned) POP(opt, 0),
                                 (unsigned) (POP(opt, 0) * 1.5));
    default:
        nla_set_errmsg(opt, "algorithm %s is not provided by libnlopt_amd (stochastic-global hot path only)",
                       nlopt_algorithm_to_string(opt->algorithm));
        return NLOPT_INVALID_ARGS;
    }
}

/* ---- elimination of fixed coordinates (reference: optimize.c:219-445) ------------------------------------------- */
typedef struct {
    nlopt_func f; nlopt_mfunc mf; void *f_data;
    unsigned n;                 /* full dimension */
    double *xfull;              /* scratch of length n, shared by the objective and the constraints of one run */
    const double *lb, *ub;      /* the caller's bounds (length n) */
} fixdim;

static void fix_expand_into(const fixdim *d, const double *xr)
{
    unsigned i, j = 0;
    for (i = 0; i < d->n; ++i) d->xfull[i] = d->lb[i] == d->ub[i] ? d->lb[i] : xr[j++];
}
static double fix_func(unsigned nr, const double *xr, double *grad, void *data)     /* elimdim_func; grad is never requested by CRS/ISRES/ESCH */
{
    const fixdim *d = (const fixdim *) data;
    (void) nr; (void) grad;
    fix_expand_into(d, xr);
    return d->f(d->n, d->xfull, NULL, d->f_data);
}
static void fix_mfunc(unsigned m, double *result, unsigned nr, const double *xr, double *grad, void *data)   /* elimdim_mfunc */
{
    const fixdim *d = (const fixdim *) data;
    (void) nr; (void) grad;
    fix_expand_into(d, xr);
    d->mf(m, result, d->n, d->xfull, NULL, d->f_data);
}
static unsigned fix_free_count(unsigned n, const double *lb, const double *ub)
{
    unsigned i, c = 0;
    for (i = 0; i < n; ++i) c += lb[i] != ub[i];
    return c;
}
static void fix_shrink(unsigned n, double *v, const double *lb, const double *ub)    /* elimdim_shrink */
{
    unsigned i, j = 0;
    if (v) for (i = 0; i < n; ++i) if (lb[i] != ub[i]) v[j++] = v[i];
}
static void fix_expand(unsigned n, double *v, const double *lb, const double *ub)    /* elimdim_expand, in place from the back */
{
    unsigned i, j = fix_free_count(n, lb, ub);
    if (!v) return;
    for (i = n; i-- > 0;) v[i] = lb[i] != ub[i] ? v[--j] : lb[i];
}
static int fix_applies(const nlopt_opt opt)                                           /* elimdim_wrapcheck, restricted to what is provided */
{
    if (fix_free_count(opt->n, opt->lb, opt->ub) == opt->n) return 0;
    return opt->algorithm == NLOPT_GN_CRS2_LM || opt->algorithm == NLOPT_GN_ISRES || opt->algorithm == NLOPT_GN_ESCH ||
           opt->algorithm == NLOPT_LN_COBYLA;
}

/* minimise on the reduced problem; x is shrunk on entry and expanded on return */
static nlopt_result minimize_fixed_eliminated(nlopt_opt opt, double *x, double *minf)
{
    nlopt_munge save_copy = opt->munge_on_copy;
    nlopt_opt r;
    fixdim *dd;
    double *xfull;
    unsigned i, nd = 1 + opt->m + opt->p;
    nlopt_result ret;
    opt->munge_on_copy = NULL;                  /* an internal copy: leave f_data un-munged (optimize.c:332-335) */
    r = nlopt_copy(opt);
    opt->munge_on_copy = save_copy;
    if (!r) { nla_set_errmsg(opt, "failure allocating elim_opt"); return NLOPT_OUT_OF_MEMORY; }
    dd = (fixdim *) calloc(nd, sizeof *dd);
    xfull = (double *) malloc(sizeof(double) * (opt->n ? opt->n : 1));
    if (!dd || !xfull) { free(dd); free(xfull); nlopt_destroy(r); nla_set_errmsg(opt, "failure allocating elim_opt"); return NLOPT_OUT_OF_MEMORY; }
    r->munge_on_destroy = r->munge_on_copy = NULL;
    r->n = fix_free_count(opt->n, opt->lb, opt->ub);
    fix_shrink(opt->n, r->lb, opt->lb, opt->ub);
    fix_shrink(opt->n, r->ub, opt->lb, opt->ub);
    fix_shrink(opt->n, r->xtol_abs, opt->lb, opt->ub);      /* (x_weights are NOT shrunk by the reference either, optimize.c:352-355) */
    fix_shrink(opt->n, r->dx, opt->lb, opt->ub);
    for (i = 0; i < nd; ++i) { dd[i].n = opt->n; dd[i].xfull = xfull; dd[i].lb = opt->lb; dd[i].ub = opt->ub; }
    dd[0].f = opt->f; dd[0].f_data = opt->f_data;
    r->f = fix_func; r->f_data = &dd[0];
    for (i = 0; i < opt->m; ++i) {
        fixdim *d = &dd[1 + i];
        d->f = opt->fc[i].f; d->mf = opt->fc[i].mf; d->f_data = opt->fc[i].f_data;
        r->fc[i].f = opt->fc[i].f ? fix_func : NULL; r->fc[i].mf = opt->fc[i].mf ? fix_mfunc : NULL; r->fc[i].f_data = d;
    }
    for (i = 0; i < opt->p; ++i) {
        fixdim *d = &dd[1 + opt->m + i];
        d->f = opt->h[i].f; d->mf = opt->h[i].mf; d->f_data = opt->h[i].f_data;
        r->h[i].f = opt->h[i].f ? fix_func : NULL; r->h[i].mf = opt->h[i].mf ? fix_mfunc : NULL; r->h[i].f_data = d;
    }
    r->trace = opt->trace; r->trace_cap = opt->trace_cap; r->progress = opt->progress; r->progress_data = opt->progress_data;
    fix_shrink(opt->n, x, opt->lb, opt->ub);
    opt->force_stop_child = r;                  /* nlopt_force_stop(opt) reaches the running copy (optimize.c:1048) */
    ret = minimize_dispatch(r, x, minf);
    opt->force_stop_child = NULL;
    opt->numevals = r->numevals;
    opt->trace_len = r->trace_len;
    opt->stats = r->stats;
    free(opt->errmsg); opt->errmsg = r->errmsg; r->errmsg = NULL;
    fix_expand(opt->n, x, opt->lb, opt->ub);
    for (i = 0; i < opt->m; ++i) r->fc[i].f_data = NULL;      /* the copies' data are ours (dd), not user data to munge */
    for (i = 0; i < opt->p; ++i) r->h[i].f_data = NULL;
    r->f_data = NULL;
    nlopt_destroy(r);
    free(dd); free(xfull);
    return ret;
}

/* memoize_func (optimize.c:450-483): the objective, remembering the best value seen inside the box */
typedef struct { nlopt_func f; void *f_data; const double *lb, *ub; double minf; double *bestx; } best_seen;
static double best_seen_objective(unsigned n, const double *x, double *grad, void *p)
{
    best_seen *d = (best_seen *) p;
    const double val = d->f(n, x, grad, d->f_data);
    unsigned i, feasible = 1;
    for (i = 0; i < n; ++i) {
        if (d->lb && x[i] < d->lb[i]) feasible = 0;
        if (d->ub && x[i] > d->ub[i]) feasible = 0;
    }
    if (feasible && val < d->minf) { d->minf = val; memcpy(d->bestx, x, sizeof(double) * n); }
    return val;
}

/* Maximisation of a device objective (a registered one, or a user's kernel): may the run keep it on the device and let the
 * drivers negate f there (opt->dev_sign), or will some part of the run call f on the host — which must then see the flipped
 * callback like any other objective?  "On the device" only where the driver really evaluates every point there:
 *   LD_LBFGS, MLSL*: always;  LD_MMA: without nonlinear constraints (with them the outer algorithm calls f on the host, mma_host.c);
 *   CRS2_LM / ESCH: unless amd_host_eval or fixed coordinates (the elimination wrapper) put a host function in front;
 *   ISRES: additionally only if every constraint is a device constraint (isres_driver.c falls back to host calls otherwise). */
static int objective_stays_on_device(const nlopt_opt opt)
{
    const nlopt_algorithm a = opt->algorithm;
    if (!(nlopt_amd_objective_id(opt->f) >= 0 || nla_userobj_is_adapter(opt->f))) return 0;
    if (a == NLOPT_LD_LBFGS || a == NLOPT_G_MLSL || a == NLOPT_G_MLSL_LDS || (a >= NLOPT_GN_MLSL && a <= NLOPT_GD_MLSL_LDS)) return 1;
    if (a == NLOPT_LD_MMA) return opt->m == 0;
    if (a == NLOPT_GN_CRS2_LM || a == NLOPT_GN_ISRES || a == NLOPT_GN_ESCH) {
        if (fix_applies(opt) || nlopt_get_param(opt, "amd_host_eval", 0) != 0) return 0;
        return a != NLOPT_GN_ISRES || nla_isres_constraints_on_device(opt->m, opt->fc, opt->p, opt->h);
    }
    return 0;
}

nlopt_result nlopt_optimize(nlopt_opt opt, double *x, double *opt_f)
{
    nlopt_func f; void *f_data; nlopt_precond pre;
    flip_data fd;
    best_seen mm;
    int maximize, memo = 0, computed = 1;
    nlopt_result ret;
    nla_unset_errmsg(opt);
    if (!opt || !opt_f || !opt->f) { if (opt) nla_set_errmsg(opt, "NULL args to nlopt_optimize"); return NLOPT_INVALID_ARGS; }
    f = opt->f; f_data = opt->f_data; pre = opt->pre;
    nlopt_set_force_stop(opt, 0);
    opt->force_stop_child = NULL;
    if ((maximize = opt->maximize)) {          /* minimise -f (optimize.c:1014-1024) */
        if (objective_stays_on_device(opt)) {
            /* a device objective stays on the device: those drivers negate f and its gradient there (nla_evaluator.sign) */
            opt->dev_sign = -1;
        } else {
            fd.f = f; fd.f_data = f_data; fd.pre = pre;
            opt->f = flipped_objective; opt->f_data = &fd;
            if (opt->pre) opt->pre = flipped_precond;
        }
        opt->stopval = -opt->stopval;
        opt->maximize = 0;
    }
    if ((memo = opt->algorithm == NLOPT_LN_COBYLA && opt->m == 0 && opt->p == 0)) {
        /* the reference returns the best FEASIBLE point any objective call of an unconstrained COBYLA run saw, not the
         * algorithm's own answer (memoize_func, optimize.c:450-508,1026-1036,1064-1071) */
        mm.f = opt->f; mm.f_data = opt->f_data; mm.lb = opt->lb; mm.ub = opt->ub; mm.minf = DBL_MAX;
        mm.bestx = (double *) malloc(sizeof(double) * (opt->n ? opt->n : 1));
        if (!mm.bestx) { nla_set_errmsg(opt, "out of memory"); memo = 0; ret = NLOPT_OUT_OF_MEMORY; computed = 0; goto restore; }
        if (x) memcpy(mm.bestx, x, sizeof(double) * opt->n);       /* (the reference leaves it uninitialised) */
        opt->f = best_seen_objective; opt->f_data = &mm;
    }
    ret = fix_applies(opt) ? minimize_fixed_eliminated(opt, x, opt_f) : minimize_dispatch(opt, x, opt_f);
    if (memo) {
        /* (an early failure — no device, invalid arguments — made no objective call: leave x and *opt_f as the dispatcher left them) */
        if (mm.minf < DBL_MAX) { memcpy(x, mm.bestx, sizeof(double) * opt->n); *opt_f = mm.minf; }
        free(mm.bestx);
        opt->f = mm.f; opt->f_data = mm.f_data;
    }
restore:
    if (maximize) {
        opt->maximize = maximize;
        opt->dev_sign = 0;
        opt->stopval = -opt->stopval;
        opt->f = f; opt->f_data = f_data; opt->pre = pre;
        if (computed) *opt_f = -*opt_f;
    }
    return ret;
}

nlopt_result nla_optimize_limited(nlopt_opt opt, double *x, double *minf, int maxeval, double maxtime)   /* optimize.c:1087-1113 */
{
    int save_maxeval;
    double save_maxtime;
    nlopt_result ret;
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    save_maxeval = nlopt_get_maxeval(opt);
    save_maxtime = nlopt_get_maxtime(opt);
    if (save_maxeval <= 0 || (maxeval > 0 && maxeval < save_maxeval)) nlopt_set_maxeval(opt, maxeval);
    if (save_maxtime <= 0 || (maxtime > 0 && maxtime < save_maxtime)) nlopt_set_maxtime(opt, maxtime);
    ret = nlopt_optimize(opt, x, minf);
    opt->maxeval = save_maxeval;        /* restored directly: the setters would clear the run's message (as optimize.c:1109-1110 does), */
    opt->maxtime = save_maxtime;        /* and the callers below want to pass a device failure's text on */
    return ret;
}
