/* api_options.c — the nlopt_opt object: lifecycle, setters/getters, constraints, parameters.
 * Behavioural twin of the reference's src/api/options.c (create :69-129, copy :131-262, params
 * :268-318, objective :322-365, bounds :369-482, constraints :486-684, tolerances :688-796,
 * force_stop :800-816, local optimiser :824-846, population/vector storage :850-851, initial
 * step :855-950, munge :954-979, errmsg :983-1003).  Pure bookkeeping; own code. */
#include "nla_internal.h"
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

/* ---- small helpers ---------------------------------------------------------------------------- */
static double *dup_doubles(const double *src, unsigned n)
{
    double *d;
    if (!src) return NULL;
    d = (double *) malloc(sizeof(double) * (n ? n : 1));
    if (d && n) memcpy(d, src, sizeof(double) * n);
    return d;
}

const char *nla_set_errmsg(nlopt_opt opt, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    opt->errmsg = nla_vsprintf(opt->errmsg, fmt, ap);
    va_end(ap);
    return opt->errmsg;
}
void nla_unset_errmsg(nlopt_opt opt) { if (opt) { free(opt->errmsg); opt->errmsg = NULL; } }
const char *nlopt_get_errmsg(nlopt_opt opt) { return opt->errmsg; }

static nlopt_result fail_msg(nlopt_opt opt, nlopt_result r, const char *msg) { if (opt) nla_set_errmsg(opt, "%s", msg); return r; }

/* ---- lifecycle ---------------------------------------------------------------------------------- */
static void free_constraints(nlopt_opt opt, nla_constraint **c, unsigned *m, unsigned *m_alloc, int munge)
{
    unsigned i;
    if (munge && opt->munge_on_destroy) for (i = 0; i < *m; ++i) opt->munge_on_destroy((*c)[i].f_data);
    for (i = 0; i < *m; ++i) free((*c)[i].tol);
    free(*c);
    *c = NULL; *m = *m_alloc = 0;
}

void nlopt_destroy(nlopt_opt opt)
{
    unsigned i;
    if (!opt) return;
    if (opt->munge_on_destroy) opt->munge_on_destroy(opt->f_data);
    free_constraints(opt, &opt->fc, &opt->m, &opt->m_alloc, 1);
    free_constraints(opt, &opt->h, &opt->p, &opt->p_alloc, 1);
    for (i = 0; i < opt->nparams; ++i) free(opt->params[i].name);
    free(opt->params);
    free(opt->lb); free(opt->ub); free(opt->xtol_abs); free(opt->x_weights); free(opt->dx);
    nlopt_destroy(opt->local_opt);
    nla_userobj_release(opt->userobj);
    free(opt->errmsg);
    free(opt);
}

nlopt_opt nlopt_create(nlopt_algorithm algorithm, unsigned n)
{
    nlopt_opt opt;
    unsigned i;
    if ((int) algorithm < 0 || algorithm >= NLOPT_NUM_ALGORITHMS) return NULL;
    opt = (nlopt_opt) calloc(1, sizeof *opt);
    if (!opt) return NULL;
    opt->algorithm = algorithm;
    opt->n = n;
    opt->stopval = -HUGE_VAL;
    if (n > 0) {
        opt->lb = (double *) malloc(sizeof(double) * n);
        opt->ub = (double *) malloc(sizeof(double) * n);
        if (!opt->lb || !opt->ub) { nlopt_destroy(opt); return NULL; }
        for (i = 0; i < n; ++i) { opt->lb[i] = -HUGE_VAL; opt->ub[i] = +HUGE_VAL; }
    }
    return opt;
}

static int copy_constraints(nlopt_opt dst, nla_constraint **dc, unsigned *dm, unsigned *dalloc,
                            const nla_constraint *sc, unsigned sm)
{
    unsigned i;
    *dc = NULL; *dm = *dalloc = 0;
    if (!sm) return 0;
    *dc = (nla_constraint *) malloc(sizeof(nla_constraint) * sm);
    if (!*dc) return -1;
    memcpy(*dc, sc, sizeof(nla_constraint) * sm);
    for (i = 0; i < sm; ++i) (*dc)[i].tol = NULL;
    *dm = *dalloc = sm;
    for (i = 0; i < sm; ++i) {
        if (dst->munge_on_copy && (*dc)[i].f_data && !((*dc)[i].f_data = dst->munge_on_copy((*dc)[i].f_data))) return -1;
        if (sc[i].tol && !((*dc)[i].tol = dup_doubles(sc[i].tol, sc[i].m))) return -1;
    }
    return 0;
}

nlopt_opt nlopt_copy(const nlopt_opt opt)
{
    nlopt_opt c;
    unsigned i;
    if (!opt) return NULL;
    c = (nlopt_opt) malloc(sizeof *c);
    if (!c) return NULL;
    *c = *opt;
    /* everything owned is re-created below; until then keep the copy destroy-safe */
    c->lb = c->ub = c->xtol_abs = c->x_weights = c->dx = NULL;
    c->fc = c->h = NULL; c->m = c->m_alloc = c->p = c->p_alloc = 0;
    c->params = NULL; c->nparams = 0;
    c->local_opt = NULL; c->errmsg = NULL; c->force_stop_child = NULL;
    c->trace = NULL; c->trace_cap = c->trace_len = 0;
    nla_userobj_retain(c->userobj);              /* shared, reference counted */
    if (c->munge_on_copy && c->f_data && !(c->f_data = c->munge_on_copy(c->f_data))) goto oom;
    if (opt->n > 0) {
        if (!(c->lb = dup_doubles(opt->lb, opt->n)) || !(c->ub = dup_doubles(opt->ub, opt->n))) goto oom;
        if (opt->xtol_abs && !(c->xtol_abs = dup_doubles(opt->xtol_abs, opt->n))) goto oom;
        if (opt->x_weights && !(c->x_weights = dup_doubles(opt->x_weights, opt->n))) goto oom;
        if (opt->dx && !(c->dx = dup_doubles(opt->dx, opt->n))) goto oom;
    }
    if (copy_constraints(c, &c->fc, &c->m, &c->m_alloc, opt->fc, opt->m)) goto oom;
    if (copy_constraints(c, &c->h, &c->p, &c->p_alloc, opt->h, opt->p)) goto oom;
    if (opt->nparams) {
        c->params = (nla_param *) calloc(opt->nparams, sizeof(nla_param));
        if (!c->params) goto oom;
        c->nparams = opt->nparams;
        for (i = 0; i < opt->nparams; ++i) {
            size_t len = strlen(opt->params[i].name) + 1;
            if (!(c->params[i].name = (char *) malloc(len))) goto oom;
            memcpy(c->params[i].name, opt->params[i].name, len);
            c->params[i].val = opt->params[i].val;
        }
    }
    if (opt->local_opt && !(c->local_opt = nlopt_copy(opt->local_opt))) goto oom;
    return c;
oom:
    c->munge_on_destroy = NULL;     /* better to leak than to free user data twice */
    nlopt_destroy(c);
    return NULL;
}

/* ---- generic named parameters -------------------------------------------------------------------- */
nlopt_result nlopt_set_param(nlopt_opt opt, const char *name, double val)
{
    size_t len;
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    if (!name) return fail_msg(opt, NLOPT_INVALID_ARGS, "invalid NULL parameter name");
    len = strnlen(name, 1024) + 1;
    if (len > 1024) return fail_msg(opt, NLOPT_INVALID_ARGS, "parameter name must be < 1024 bytes");
    if (!strncmp(name, "amd_", 4)) {
        /* this library's own switches: a name it does not (or no longer) read is refused instead of being stored and silently ignored — an
         * A/B script written for a switch that has since been removed would otherwise compare two identical runs (advisor, round 5) */
        static const char *const known[] = { "amd_forward", "amd_shard", "amd_shard_windows", "amd_cu_share", "amd_max_spec", "amd_window_factor", "amd_host_eval",
                                             "amd_exact_dot", "amd_isres_evolve_serial", "amd_isres_overlap", "amd_lbfgs_streaming", "amd_mlsl_seg_regens", "amd_cobyla_host", "amd_cobyla_min_batch", NULL };
        int k;
        for (k = 0; known[k] && strcmp(name, known[k]); ++k) { }
        if (!known[k]) return fail_msg(opt, NLOPT_INVALID_ARGS, "nlopt_amd: no such switch (the amd_* names this build reads: include/nlopt_amd.h, INTEGRATION.md E)");
    }
    for (i = 0; i < opt->nparams; ++i) if (!strcmp(name, opt->params[i].name)) break;
    if (i == opt->nparams) {
        nla_param *np = (nla_param *) realloc(opt->params, sizeof(nla_param) * (opt->nparams + 1));
        if (!np) return NLOPT_OUT_OF_MEMORY;
        opt->params = np;
        if (!(np[i].name = (char *) malloc(len))) return NLOPT_OUT_OF_MEMORY;
        memcpy(np[i].name, name, len);
        opt->nparams++;
    }
    opt->params[i].val = val;
    return NLOPT_SUCCESS;
}

static int find_param(const nlopt_opt opt, const char *name)
{
    unsigned i;
    if (!opt || !name || strnlen(name, 1024) == 1024) return -1;
    for (i = 0; i < opt->nparams; ++i) if (!strcmp(name, opt->params[i].name)) return (int) i;
    return -1;
}
double nlopt_get_param(const nlopt_opt opt, const char *name, double defaultval)
{
    int i = find_param(opt, name);
    return i < 0 ? defaultval : opt->params[i].val;
}
int nlopt_has_param(const nlopt_opt opt, const char *name) { return find_param(opt, name) >= 0; }
unsigned nlopt_num_params(const nlopt_opt opt) { return opt ? opt->nparams : 0; }
const char *nlopt_nth_param(const nlopt_opt opt, unsigned n) { return opt && n < opt->nparams ? opt->params[n].name : NULL; }

/* ---- objective ------------------------------------------------------------------------------------ */
static nlopt_result set_objective(nlopt_opt opt, nlopt_func f, nlopt_precond pre, void *f_data, int maximize)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (opt->munge_on_destroy) opt->munge_on_destroy(opt->f_data);
    nla_userobj_release(opt->userobj);           /* a user device objective (userobj.c) goes with the objective it was bound as */
    opt->userobj = NULL;
    opt->f = f; opt->f_data = f_data; opt->pre = pre; opt->maximize = maximize;
    /* an untouched stopval follows the direction of optimisation (options.c:332-333,352-353) */
    if (nla_isinf(opt->stopval) && (maximize ? opt->stopval < 0 : opt->stopval > 0))
        opt->stopval = maximize ? +HUGE_VAL : -HUGE_VAL;
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_precond_min_objective(nlopt_opt o, nlopt_func f, nlopt_precond pre, void *d) { return set_objective(o, f, pre, d, 0); }
nlopt_result nlopt_set_precond_max_objective(nlopt_opt o, nlopt_func f, nlopt_precond pre, void *d) { return set_objective(o, f, pre, d, 1); }
nlopt_result nlopt_set_min_objective(nlopt_opt o, nlopt_func f, void *d) { return set_objective(o, f, NULL, d, 0); }
nlopt_result nlopt_set_max_objective(nlopt_opt o, nlopt_func f, void *d) { return set_objective(o, f, NULL, d, 1); }

/* ---- bounds: a denormal-width box collapses onto the other bound (options.c:376-379 etc.) ------- */
static void snap_lower(nlopt_opt opt, unsigned i) { if (opt->lb[i] < opt->ub[i] && nla_istiny(opt->ub[i] - opt->lb[i])) opt->lb[i] = opt->ub[i]; }
static void snap_upper(nlopt_opt opt, unsigned i) { if (opt->lb[i] < opt->ub[i] && nla_istiny(opt->ub[i] - opt->lb[i])) opt->ub[i] = opt->lb[i]; }

nlopt_result nlopt_set_lower_bounds(nlopt_opt opt, const double *lb)
{
    unsigned i;
    nla_unset_errmsg(opt);
    if (!opt || (opt->n && !lb)) return NLOPT_INVALID_ARGS;
    for (i = 0; i < opt->n; ++i) opt->lb[i] = lb[i];
    for (i = 0; i < opt->n; ++i) snap_lower(opt, i);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_lower_bounds1(nlopt_opt opt, double lb)
{
    unsigned i;
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    for (i = 0; i < opt->n; ++i) { opt->lb[i] = lb; snap_lower(opt, i); }
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_lower_bound(nlopt_opt opt, int i, double lb)
{
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    if (i < 0 || i >= (int) opt->n) return fail_msg(opt, NLOPT_INVALID_ARGS, "invalid bound index");
    opt->lb[i] = lb; snap_lower(opt, (unsigned) i);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_get_lower_bounds(const nlopt_opt opt, double *lb)
{
    nla_unset_errmsg(opt);
    if (!opt || (opt->n && !lb)) return NLOPT_INVALID_ARGS;
    if (opt->n) memcpy(lb, opt->lb, sizeof(double) * opt->n);     /* (n = 0: both pointers may be NULL, which memcpy must not be given) */
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_upper_bounds(nlopt_opt opt, const double *ub)
{
    unsigned i;
    nla_unset_errmsg(opt);
    if (!opt || (opt->n && !ub)) return NLOPT_INVALID_ARGS;
    for (i = 0; i < opt->n; ++i) opt->ub[i] = ub[i];
    for (i = 0; i < opt->n; ++i) snap_upper(opt, i);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_upper_bounds1(nlopt_opt opt, double ub)
{
    unsigned i;
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    for (i = 0; i < opt->n; ++i) { opt->ub[i] = ub; snap_upper(opt, i); }
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_upper_bound(nlopt_opt opt, int i, double ub)
{
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    if (i < 0 || i >= (int) opt->n) return fail_msg(opt, NLOPT_INVALID_ARGS, "invalid bound index");
    opt->ub[i] = ub; snap_upper(opt, (unsigned) i);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_get_upper_bounds(const nlopt_opt opt, double *ub)
{
    nla_unset_errmsg(opt);
    if (!opt || (opt->n && !ub)) return NLOPT_INVALID_ARGS;
    if (opt->n) memcpy(ub, opt->ub, sizeof(double) * opt->n);
    return NLOPT_SUCCESS;
}

/* ---- nonlinear constraints ------------------------------------------------------------------------ */
static int is_auglag(nlopt_algorithm a)
{
    return a == NLOPT_AUGLAG || a == NLOPT_AUGLAG_EQ || a == NLOPT_LN_AUGLAG || a == NLOPT_LN_AUGLAG_EQ ||
           a == NLOPT_LD_AUGLAG || a == NLOPT_LD_AUGLAG_EQ;
}
static int accepts_inequality(nlopt_algorithm a)      /* options.c:549-554 */
{
    return a == NLOPT_LD_MMA || a == NLOPT_LD_CCSAQ || a == NLOPT_LD_SLSQP || a == NLOPT_LN_COBYLA || is_auglag(a) ||
           a == NLOPT_GN_ISRES || a == NLOPT_GN_ORIG_DIRECT || a == NLOPT_GN_ORIG_DIRECT_L || a == NLOPT_GN_AGS;
}
static int accepts_equality(nlopt_algorithm a)        /* options.c:615-620 */
{
    return is_auglag(a) || a == NLOPT_LD_SLSQP || a == NLOPT_GN_ISRES || a == NLOPT_LN_COBYLA;
}

static nlopt_result push_constraint(nlopt_opt opt, unsigned *m, unsigned *m_alloc, nla_constraint **c, unsigned fm,
                                    nlopt_func fc, nlopt_mfunc mfc, nlopt_precond pre, void *data, const double *tol)
{
    double *tolcopy;
    unsigned i;
    if ((fc && mfc) || (fc && fm != 1) || (!fc && !mfc)) return NLOPT_INVALID_ARGS;
    if (tol) for (i = 0; i < fm; ++i) if (tol[i] < 0) return fail_msg(opt, NLOPT_INVALID_ARGS, "negative constraint tolerance");
    tolcopy = (double *) calloc(fm ? fm : 1, sizeof(double));
    if (!tolcopy) return NLOPT_OUT_OF_MEMORY;
    if (tol) memcpy(tolcopy, tol, sizeof(double) * fm);
    if (*m + 1 > *m_alloc) {
        unsigned na = 2 * (*m + 1);
        nla_constraint *nc = (nla_constraint *) realloc(*c, sizeof(nla_constraint) * na);
        if (!nc) { free(tolcopy); return NLOPT_OUT_OF_MEMORY; }
        *c = nc; *m_alloc = na;
    }
    (*c)[*m].m = fm; (*c)[*m].f = fc; (*c)[*m].mf = mfc; (*c)[*m].pre = pre; (*c)[*m].f_data = data; (*c)[*m].tol = tolcopy;
    ++*m;
    return NLOPT_SUCCESS;
}

static nlopt_result add_any(nlopt_opt opt, int equality, unsigned fm, nlopt_func fc, nlopt_mfunc mfc, nlopt_precond pre,
                            void *data, const double *tol)
{
    nlopt_result ret;
    nla_unset_errmsg(opt);
    if (mfc && !fm) {      /* empty vector constraints are always fine */
        if (opt && opt->munge_on_destroy) opt->munge_on_destroy(data);
        return NLOPT_SUCCESS;
    }
    if (!opt) ret = NLOPT_INVALID_ARGS;
    else if (!(equality ? accepts_equality(opt->algorithm) : accepts_inequality(opt->algorithm)))
        ret = fail_msg(opt, NLOPT_INVALID_ARGS, "invalid algorithm for constraints");
    else if (equality) ret = push_constraint(opt, &opt->p, &opt->p_alloc, &opt->h, fm, fc, mfc, pre, data, tol);
    else ret = push_constraint(opt, &opt->m, &opt->m_alloc, &opt->fc, fm, fc, mfc, pre, data, tol);
    if (ret < 0 && opt && opt->munge_on_destroy) opt->munge_on_destroy(data);
    return ret;
}

nlopt_result nlopt_remove_inequality_constraints(nlopt_opt opt)
{
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    free_constraints(opt, &opt->fc, &opt->m, &opt->m_alloc, 1);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_remove_equality_constraints(nlopt_opt opt)
{
    nla_unset_errmsg(opt);
    if (!opt) return NLOPT_INVALID_ARGS;
    free_constraints(opt, &opt->h, &opt->p, &opt->p_alloc, 1);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_add_inequality_mconstraint(nlopt_opt o, unsigned m, nlopt_mfunc fc, void *d, const double *tol) { return add_any(o, 0, m, NULL, fc, NULL, d, tol); }
nlopt_result nlopt_add_precond_inequality_constraint(nlopt_opt o, nlopt_func fc, nlopt_precond pre, void *d, double tol) { return add_any(o, 0, 1, fc, NULL, pre, d, &tol); }
nlopt_result nlopt_add_inequality_constraint(nlopt_opt o, nlopt_func fc, void *d, double tol) { return add_any(o, 0, 1, fc, NULL, NULL, d, &tol); }
nlopt_result nlopt_add_equality_mconstraint(nlopt_opt o, unsigned m, nlopt_mfunc fc, void *d, const double *tol) { return add_any(o, 1, m, NULL, fc, NULL, d, tol); }
nlopt_result nlopt_add_precond_equality_constraint(nlopt_opt o, nlopt_func fc, nlopt_precond pre, void *d, double tol) { return add_any(o, 1, 1, fc, NULL, pre, d, &tol); }
nlopt_result nlopt_add_equality_constraint(nlopt_opt o, nlopt_func fc, void *d, double tol) { return add_any(o, 1, 1, fc, NULL, NULL, d, &tol); }

/* ---- scalar stopping parameters ------------------------------------------------------------------- */
#define NLA_SCALAR(name, T, field)                                                             \
    nlopt_result nlopt_set_##name(nlopt_opt opt, T v)                                          \
    { if (!opt) return NLOPT_INVALID_ARGS; nla_unset_errmsg(opt); opt->field = v; return NLOPT_SUCCESS; } \
    T nlopt_get_##name(const nlopt_opt opt) { return opt->field; }
NLA_SCALAR(stopval, double, stopval)
NLA_SCALAR(ftol_rel, double, ftol_rel)
NLA_SCALAR(ftol_abs, double, ftol_abs)
NLA_SCALAR(xtol_rel, double, xtol_rel)
NLA_SCALAR(maxeval, int, maxeval)
NLA_SCALAR(maxtime, double, maxtime)
NLA_SCALAR(population, unsigned, stochastic_population)
NLA_SCALAR(vector_storage, unsigned, vector_storage)
int nlopt_get_numevals(const nlopt_opt opt) { return opt->numevals; }
nlopt_algorithm nlopt_get_algorithm(const nlopt_opt opt) { return opt->algorithm; }
unsigned nlopt_get_dimension(const nlopt_opt opt) { return opt->n; }

static nlopt_result ensure_vec(nlopt_opt opt, double **v)
{
    if (!*v && opt->n > 0) { *v = (double *) calloc(opt->n, sizeof(double)); if (!*v) return NLOPT_OUT_OF_MEMORY; }
    return NLOPT_SUCCESS;
}

nlopt_result nlopt_set_xtol_abs(nlopt_opt opt, const double *tol)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (!tol) { free(opt->xtol_abs); opt->xtol_abs = NULL; return NLOPT_SUCCESS; }
    if (ensure_vec(opt, &opt->xtol_abs) != NLOPT_SUCCESS) return NLOPT_OUT_OF_MEMORY;
    if (opt->n) memcpy(opt->xtol_abs, tol, sizeof(double) * opt->n);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_xtol_abs1(nlopt_opt opt, double tol)
{
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (ensure_vec(opt, &opt->xtol_abs) != NLOPT_SUCCESS) return NLOPT_OUT_OF_MEMORY;
    for (i = 0; i < opt->n; ++i) opt->xtol_abs[i] = tol;
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_get_xtol_abs(const nlopt_opt opt, double *tol)
{
    unsigned i;
    nla_unset_errmsg(opt);
    if (!opt || (opt->n && !tol)) return NLOPT_INVALID_ARGS;
    for (i = 0; i < opt->n; ++i) tol[i] = opt->xtol_abs ? opt->xtol_abs[i] : 0;
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_x_weights(nlopt_opt opt, const double *w)
{
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (!w) { free(opt->x_weights); opt->x_weights = NULL; return NLOPT_SUCCESS; }
    for (i = 0; i < opt->n; ++i) if (w[i] < 0) return fail_msg(opt, NLOPT_INVALID_ARGS, "invalid negative weight");
    if (ensure_vec(opt, &opt->x_weights) != NLOPT_SUCCESS) return NLOPT_OUT_OF_MEMORY;
    if (opt->n) memcpy(opt->x_weights, w, sizeof(double) * opt->n);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_x_weights1(nlopt_opt opt, double w)
{
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    if (w < 0) return fail_msg(opt, NLOPT_INVALID_ARGS, "invalid negative weight");
    nla_unset_errmsg(opt);
    if (ensure_vec(opt, &opt->x_weights) != NLOPT_SUCCESS) return NLOPT_OUT_OF_MEMORY;
    for (i = 0; i < opt->n; ++i) opt->x_weights[i] = w;
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_get_x_weights(const nlopt_opt opt, double *w)
{
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    if (opt->n && !w) return fail_msg(opt, NLOPT_INVALID_ARGS, "invalid NULL weights");
    nla_unset_errmsg(opt);
    for (i = 0; i < opt->n; ++i) w[i] = opt->x_weights ? opt->x_weights[i] : 1;
    return NLOPT_SUCCESS;
}

/* ---- cooperative cancellation (options.c:800-816) ------------------------------------------------- */
nlopt_result nlopt_set_force_stop(nlopt_opt opt, int val)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    opt->force_stop = val;
    if (opt->force_stop_child) return nlopt_set_force_stop(opt->force_stop_child, val);
    return NLOPT_SUCCESS;
}
int nlopt_get_force_stop(const nlopt_opt opt) { return opt->force_stop; }
nlopt_result nlopt_force_stop(nlopt_opt opt) { return nlopt_set_force_stop(opt, 1); }

/* ---- local optimiser: stored as a private, objective-less copy (options.c:824-846) ---------------- */
nlopt_result nlopt_set_local_optimizer(nlopt_opt opt, const nlopt_opt local_opt)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (local_opt && local_opt->n != opt->n) return fail_msg(opt, NLOPT_INVALID_ARGS, "dimension mismatch in local optimizer");
    nlopt_destroy(opt->local_opt);
    opt->local_opt = nlopt_copy(local_opt);
    if (local_opt) {
        if (!opt->local_opt) return NLOPT_OUT_OF_MEMORY;
        nlopt_set_lower_bounds(opt->local_opt, opt->lb);
        nlopt_set_upper_bounds(opt->local_opt, opt->ub);
        nlopt_remove_inequality_constraints(opt->local_opt);
        nlopt_remove_equality_constraints(opt->local_opt);
        nlopt_set_min_objective(opt->local_opt, NULL, NULL);
        nlopt_set_munge(opt->local_opt, NULL, NULL);
        opt->local_opt->force_stop = 0;
    }
    return NLOPT_SUCCESS;
}

/* ---- initial step (options.c:855-950) --------------------------------------------------------------- */
nlopt_result nlopt_set_initial_step1(nlopt_opt opt, double dx)
{
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (dx == 0) return fail_msg(opt, NLOPT_INVALID_ARGS, "zero step size");
    if (!opt->dx && opt->n > 0 && !(opt->dx = (double *) malloc(sizeof(double) * opt->n))) return NLOPT_OUT_OF_MEMORY;
    for (i = 0; i < opt->n; ++i) opt->dx[i] = dx;
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_initial_step(nlopt_opt opt, const double *dx)
{
    unsigned i;
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (!dx) { free(opt->dx); opt->dx = NULL; return NLOPT_SUCCESS; }
    for (i = 0; i < opt->n; ++i) if (dx[i] == 0) return fail_msg(opt, NLOPT_INVALID_ARGS, "zero step size");
    if (!opt->dx && nlopt_set_initial_step1(opt, 1) == NLOPT_OUT_OF_MEMORY) return NLOPT_OUT_OF_MEMORY;
    if (opt->n) memcpy(opt->dx, dx, sizeof(double) * opt->n);
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_set_default_initial_step(nlopt_opt opt, const double *x)
{
    unsigned i;
    nla_unset_errmsg(opt);
    if (!opt || !x) return NLOPT_INVALID_ARGS;
    if (!opt->dx && nlopt_set_initial_step1(opt, 1) == NLOPT_OUT_OF_MEMORY) return NLOPT_OUT_OF_MEMORY;
    for (i = 0; i < opt->n; ++i) {      /* heuristic of options.c:921-946: a quarter box, or 3/4 of the gap to a bound */
        const double lo = opt->lb[i], hi = opt->ub[i];
        double step = HUGE_VAL;
        if (!nla_isinf(hi) && !nla_isinf(lo) && (hi - lo) * 0.25 < step && hi > lo) step = (hi - lo) * 0.25;
        if (!nla_isinf(hi) && hi - x[i] < step && hi > x[i]) step = (hi - x[i]) * 0.75;
        if (!nla_isinf(lo) && x[i] - lo < step && x[i] > lo) step = (x[i] - lo) * 0.75;
        if (nla_isinf(step)) {
            if (!nla_isinf(hi) && fabs(hi - x[i]) < fabs(step)) step = (hi - x[i]) * 1.1;
            if (!nla_isinf(lo) && fabs(x[i] - lo) < fabs(step)) step = (x[i] - lo) * 1.1;
        }
        if (nla_isinf(step) || nla_istiny(step)) step = x[i];
        if (nla_isinf(step) || step == 0.0) step = 1;
        opt->dx[i] = step;
    }
    return NLOPT_SUCCESS;
}
nlopt_result nlopt_get_initial_step(const nlopt_opt opt, const double *x, double *dx)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (!opt->n) return NLOPT_SUCCESS;
    if (opt->dx) { if (opt->n) memcpy(dx, opt->dx, sizeof(double) * opt->n); return NLOPT_SUCCESS; }
    {   /* x-dependent default: compute, hand out, do not keep */
        nlopt_result ret = nlopt_set_default_initial_step(opt, x);
        if (ret != NLOPT_SUCCESS) return ret;
        if (opt->n) memcpy(dx, opt->dx, sizeof(double) * opt->n);
        free(opt->dx); opt->dx = NULL;
    }
    return NLOPT_SUCCESS;
}

/* ---- wrapper hooks (options.c:954-979) --------------------------------------------------------------- */
void nlopt_set_munge(nlopt_opt opt, nlopt_munge munge_on_destroy, nlopt_munge munge_on_copy)
{
    if (opt) { opt->munge_on_destroy = munge_on_destroy; opt->munge_on_copy = munge_on_copy; }
}
void nlopt_munge_data(nlopt_opt opt, nlopt_munge2 munge, void *data)
{
    unsigned i;
    if (!opt || !munge) return;
    opt->f_data = munge(opt->f_data, data);
    for (i = 0; i < opt->m; ++i) opt->fc[i].f_data = munge(opt->fc[i].f_data, data);
    for (i = 0; i < opt->p; ++i) opt->h[i].f_data = munge(opt->h[i].f_data, data);
}

/* ---- libnlopt_amd additions --------------------------------------------------------------------------- */
nlopt_result nlopt_amd_set_trace(nlopt_opt opt, nlopt_amd_trace_rec *buf, size_t cap)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    opt->trace = buf; opt->trace_cap = buf ? cap : 0; opt->trace_len = 0;
    return NLOPT_SUCCESS;
}
size_t nlopt_amd_trace_len(const nlopt_opt opt) { return opt ? opt->trace_len : 0; }
nlopt_result nlopt_amd_get_stats(const nlopt_opt opt, nlopt_amd_stats *out)
{
    if (!opt || !out) return NLOPT_INVALID_ARGS;
    *out = opt->stats;
    return NLOPT_SUCCESS;
}
/* the same for a caller built against another version of the header: at most `bytes` bytes are written (the struct only ever grows at
 * its end), and the library's own size comes back so that the caller can tell */
size_t nlopt_amd_get_stats_sized(const nlopt_opt opt, void *out, size_t bytes)
{
    if (!opt || !out) return 0;
    memcpy(out, &opt->stats, bytes < sizeof opt->stats ? bytes : sizeof opt->stats);
    return sizeof opt->stats;
}
