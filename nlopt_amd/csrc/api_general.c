/* api_general.c — names, versions and enum<->string maps of the public API.
 * Same observable strings as the reference's src/api/general.c (:28-33 version, :37-93 long
 * names, :107-158 short names, :161-197 result names) so that programs printing or parsing them
 * behave identically; the table is keyed by the ABI enum values of include/nlopt.h. */
#include "nla_internal.h"
#include <string.h>

void nlopt_version(int *major, int *minor, int *bugfix)
{
    *major = NLA_VERSION_MAJOR; *minor = NLA_VERSION_MINOR; *bugfix = NLA_VERSION_BUGFIX;
}

static const struct { const char *key, *descr; } g_alg[NLOPT_NUM_ALGORITHMS] = {
    [NLOPT_GN_DIRECT] = { "GN_DIRECT", "DIRECT (global, no-derivative)" },
    [NLOPT_GN_DIRECT_L] = { "GN_DIRECT_L", "DIRECT-L (global, no-derivative)" },
    [NLOPT_GN_DIRECT_L_RAND] = { "GN_DIRECT_L_RAND", "Randomized DIRECT-L (global, no-derivative)" },
    [NLOPT_GN_DIRECT_NOSCAL] = { "GN_DIRECT_NOSCAL", "Unscaled DIRECT (global, no-derivative)" },
    [NLOPT_GN_DIRECT_L_NOSCAL] = { "GN_DIRECT_L_NOSCAL", "Unscaled DIRECT-L (global, no-derivative)" },
    [NLOPT_GN_DIRECT_L_RAND_NOSCAL] = { "GN_DIRECT_L_RAND_NOSCAL", "Unscaled Randomized DIRECT-L (global, no-derivative)" },
    [NLOPT_GN_ORIG_DIRECT] = { "GN_ORIG_DIRECT", "Original DIRECT version (global, no-derivative)" },
    [NLOPT_GN_ORIG_DIRECT_L] = { "GN_ORIG_DIRECT_L", "Original DIRECT-L version (global, no-derivative)" },
    [NLOPT_GD_STOGO] = { "GD_STOGO", "StoGO (NOT COMPILED)" },
    [NLOPT_GD_STOGO_RAND] = { "GD_STOGO_RAND", "StoGO randomized (NOT COMPILED)" },
    [NLOPT_LD_LBFGS_NOCEDAL] = { "NLOPT_LD_LBFGS_NOCEDAL", "original L-BFGS code by Nocedal et al. (NOT COMPILED)" },
    [NLOPT_LD_LBFGS] = { "LD_LBFGS", "Limited-memory BFGS (L-BFGS) (local, derivative-based)" },
    [NLOPT_LN_PRAXIS] = { "LN_PRAXIS", "Principal-axis, praxis (local, no-derivative)" },
    [NLOPT_LD_VAR1] = { "LD_VAR1", "Limited-memory variable-metric, rank 1 (local, derivative-based)" },
    [NLOPT_LD_VAR2] = { "LD_VAR2", "Limited-memory variable-metric, rank 2 (local, derivative-based)" },
    [NLOPT_LD_TNEWTON] = { "LD_TNEWTON", "Truncated Newton (local, derivative-based)" },
    [NLOPT_LD_TNEWTON_RESTART] = { "LD_TNEWTON_RESTART", "Truncated Newton with restarting (local, derivative-based)" },
    [NLOPT_LD_TNEWTON_PRECOND] = { "LD_TNEWTON_PRECOND", "Preconditioned truncated Newton (local, derivative-based)" },
    [NLOPT_LD_TNEWTON_PRECOND_RESTART] = { "LD_TNEWTON_PRECOND_RESTART", "Preconditioned truncated Newton with restarting (local, derivative-based)" },
    [NLOPT_GN_CRS2_LM] = { "GN_CRS2_LM", "Controlled random search (CRS2) with local mutation (global, no-derivative)" },
    [NLOPT_GN_MLSL] = { "GN_MLSL", "Multi-level single-linkage (MLSL), random (global, no-derivative)" },
    [NLOPT_GD_MLSL] = { "GD_MLSL", "Multi-level single-linkage (MLSL), random (global, derivative)" },
    [NLOPT_GN_MLSL_LDS] = { "GN_MLSL_LDS", "Multi-level single-linkage (MLSL), quasi-random (global, no-derivative)" },
    [NLOPT_GD_MLSL_LDS] = { "GD_MLSL_LDS", "Multi-level single-linkage (MLSL), quasi-random (global, derivative)" },
    [NLOPT_LD_MMA] = { "LD_MMA", "Method of Moving Asymptotes (MMA) (local, derivative)" },
    [NLOPT_LN_COBYLA] = { "LN_COBYLA", "COBYLA (Constrained Optimization BY Linear Approximations) (local, no-derivative)" },
    [NLOPT_LN_NEWUOA] = { "LN_NEWUOA", "NEWUOA unconstrained optimization via quadratic models (local, no-derivative)" },
    [NLOPT_LN_NEWUOA_BOUND] = { "LN_NEWUOA_BOUND", "Bound-constrained optimization via NEWUOA-based quadratic models (local, no-derivative)" },
    [NLOPT_LN_NELDERMEAD] = { "LN_NELDERMEAD", "Nelder-Mead simplex algorithm (local, no-derivative)" },
    [NLOPT_LN_SBPLX] = { "LN_SBPLX", "Sbplx variant of Nelder-Mead (re-implementation of Rowan's Subplex) (local, no-derivative)" },
    [NLOPT_LN_AUGLAG] = { "LN_AUGLAG", "Augmented Lagrangian method (local, no-derivative)" },
    [NLOPT_LD_AUGLAG] = { "LD_AUGLAG", "Augmented Lagrangian method (local, derivative)" },
    [NLOPT_LN_AUGLAG_EQ] = { "LN_AUGLAG_EQ", "Augmented Lagrangian method for equality constraints (local, no-derivative)" },
    [NLOPT_LD_AUGLAG_EQ] = { "LD_AUGLAG_EQ", "Augmented Lagrangian method for equality constraints (local, derivative)" },
    [NLOPT_LN_BOBYQA] = { "LN_BOBYQA", "BOBYQA bound-constrained optimization via quadratic models (local, no-derivative)" },
    [NLOPT_GN_ISRES] = { "GN_ISRES", "ISRES evolutionary constrained optimization (global, no-derivative)" },
    [NLOPT_AUGLAG] = { "AUGLAG", "Augmented Lagrangian method (needs sub-algorithm)" },
    [NLOPT_AUGLAG_EQ] = { "AUGLAG_EQ", "Augmented Lagrangian method for equality constraints (needs sub-algorithm)" },
    [NLOPT_G_MLSL] = { "G_MLSL", "Multi-level single-linkage (MLSL), random (global, needs sub-algorithm)" },
    [NLOPT_G_MLSL_LDS] = { "G_MLSL_LDS", "Multi-level single-linkage (MLSL), quasi-random (global, needs sub-algorithm)" },
    [NLOPT_LD_SLSQP] = { "LD_SLSQP", "Sequential Quadratic Programming (SQP) (local, derivative)" },
    [NLOPT_LD_CCSAQ] = { "LD_CCSAQ", "CCSA (Conservative Convex Separable Approximations) with simple quadratic approximations (local, derivative)" },
    [NLOPT_GN_ESCH] = { "GN_ESCH", "ESCH evolutionary strategy" },
    [NLOPT_GN_AGS] = { "GN_AGS", "AGS (NOT COMPILED)" },
};

const char *nlopt_algorithm_name(nlopt_algorithm a)
{
    if ((int) a < 0 || a >= NLOPT_NUM_ALGORITHMS) return "UNKNOWN";
    return g_alg[a].descr;
}

const char *nlopt_algorithm_to_string(nlopt_algorithm a)
{
    if ((int) a < 0 || a >= NLOPT_NUM_ALGORITHMS) return NULL;
    return g_alg[a].key;
}

nlopt_algorithm nlopt_algorithm_from_string(const char *name)
{
    int i;
    if (!name) return (nlopt_algorithm) -1;
    for (i = 0; i < NLOPT_NUM_ALGORITHMS; ++i)
        if (g_alg[i].key && !strcmp(name, g_alg[i].key)) return (nlopt_algorithm) i;
    return (nlopt_algorithm) -1;
}

const char *nlopt_result_to_string(nlopt_result r)
{
    switch (r) {
    case NLOPT_FAILURE: return "FAILURE";
    case NLOPT_INVALID_ARGS: return "INVALID_ARGS";
    case NLOPT_OUT_OF_MEMORY: return "OUT_OF_MEMORY";
    case NLOPT_ROUNDOFF_LIMITED: return "ROUNDOFF_LIMITED";
    case NLOPT_FORCED_STOP: return "FORCED_STOP";
    case NLOPT_SUCCESS: return "SUCCESS";
    case NLOPT_STOPVAL_REACHED: return "STOPVAL_REACHED";
    case NLOPT_FTOL_REACHED: return "FTOL_REACHED";
    case NLOPT_XTOL_REACHED: return "XTOL_REACHED";
    case NLOPT_MAXEVAL_REACHED: return "MAXEVAL_REACHED";
    case NLOPT_MAXTIME_REACHED: return "MAXTIME_REACHED";
    default: return NULL;
    }
}

nlopt_result nlopt_result_from_string(const char *name)
{
    int i;
    if (!name) return (nlopt_result) -1;
    for (i = NLOPT_NUM_FAILURES + 1; i < NLOPT_NUM_RESULTS; ++i) {
        const char *s = nlopt_result_to_string((nlopt_result) i);
        if (s && !strcmp(name, s)) return (nlopt_result) i;
    }
    return (nlopt_result) -1;
}

/* deprecated-API globals (src/api/deprecated.c:28-61) still read by the dispatcher */
int nla_stochastic_population = 0;
nlopt_algorithm nla_local_search_alg_deriv = NLOPT_LD_MMA;
nlopt_algorithm nla_local_search_alg_nonderiv = NLOPT_LN_COBYLA;
int nla_local_search_maxeval = -1;

int nlopt_get_stochastic_population(void) { return nla_stochastic_population; }
void nlopt_set_stochastic_population(int pop) { nla_stochastic_population = pop < 0 ? 0 : pop; }
void nlopt_get_local_search_algorithm(nlopt_algorithm *deriv, nlopt_algorithm *nonderiv, int *maxeval)
{
    *deriv = nla_local_search_alg_deriv; *nonderiv = nla_local_search_alg_nonderiv; *maxeval = nla_local_search_maxeval;
}
void nlopt_set_local_search_algorithm(nlopt_algorithm deriv, nlopt_algorithm nonderiv, int maxeval)
{
    nla_local_search_alg_deriv = deriv; nla_local_search_alg_nonderiv = nonderiv; nla_local_search_maxeval = maxeval;
}

/* The pre-2.0 one-call interface (reference: src/api/nlopt.h:317-337, src/api/deprecated.c:65-189): everything the object API
 * sets, as positional arguments.  Kept because callers of the global-search path written against it — nlopt_minimize(NLOPT_GN_CRS2_LM,
 * ...) — otherwise cannot link.  The old callback type differs from nlopt_func only in the signedness of n; like the reference,
 * the pointers are handed on as they are.  Constraint i gets the datum at byte offset i * datum_size.  htol_rel is ignored
 * (deprecated.c:97), and nlopt_minimize_constrained passes its ftol pair in the htol slots (deprecated.c:177). */
nlopt_result nlopt_minimize_econstrained(nlopt_algorithm algorithm, int n, nlopt_func_old f, void *f_data,
                                         int m, nlopt_func_old fc, void *fc_data, ptrdiff_t fc_datum_size,
                                         int p, nlopt_func_old h, void *h_data, ptrdiff_t h_datum_size,
                                         const double *lb, const double *ub, double *x, double *minf,
                                         double minf_max, double ftol_rel, double ftol_abs, double xtol_rel, const double *xtol_abs,
                                         double htol_rel, double htol_abs, int maxeval, double maxtime)
{
    nlopt_result ret = NLOPT_INVALID_ARGS;
    nlopt_opt opt;
    int i;
    (void) htol_rel;
    if (n < 0 || m < 0 || p < 0) return ret;
    opt = nlopt_create(algorithm, (unsigned) n);
    if (!opt) return ret;
#define LEGACY_STEP(call) do { ret = (call); if (ret != NLOPT_SUCCESS) goto out; } while (0)
    LEGACY_STEP(nlopt_set_min_objective(opt, (nlopt_func) f, f_data));
    for (i = 0; i < m; ++i)
        LEGACY_STEP(nlopt_add_inequality_constraint(opt, (nlopt_func) fc, (char *) fc_data + i * fc_datum_size, 0.0));
    for (i = 0; i < p; ++i)
        LEGACY_STEP(nlopt_add_equality_constraint(opt, (nlopt_func) h, (char *) h_data + i * h_datum_size, htol_abs));
    LEGACY_STEP(nlopt_set_lower_bounds(opt, lb));
    LEGACY_STEP(nlopt_set_upper_bounds(opt, ub));
    LEGACY_STEP(nlopt_set_stopval(opt, minf_max));
    LEGACY_STEP(nlopt_set_ftol_rel(opt, ftol_rel));
    LEGACY_STEP(nlopt_set_ftol_abs(opt, ftol_abs));
    LEGACY_STEP(nlopt_set_xtol_rel(opt, xtol_rel));
    if (xtol_abs) LEGACY_STEP(nlopt_set_xtol_abs(opt, xtol_abs));
    LEGACY_STEP(nlopt_set_maxeval(opt, maxeval));
    LEGACY_STEP(nlopt_set_maxtime(opt, maxtime));
#undef LEGACY_STEP
    ret = nlopt_optimize(opt, x, minf);
out:
    nlopt_destroy(opt);
    return ret;
}

nlopt_result nlopt_minimize_constrained(nlopt_algorithm algorithm, int n, nlopt_func_old f, void *f_data,
                                        int m, nlopt_func_old fc, void *fc_data, ptrdiff_t fc_datum_size,
                                        const double *lb, const double *ub, double *x, double *minf,
                                        double minf_max, double ftol_rel, double ftol_abs, double xtol_rel, const double *xtol_abs,
                                        int maxeval, double maxtime)
{
    return nlopt_minimize_econstrained(algorithm, n, f, f_data, m, fc, fc_data, fc_datum_size, 0, NULL, NULL, 0, lb, ub, x, minf,
                                       minf_max, ftol_rel, ftol_abs, xtol_rel, xtol_abs, ftol_rel, ftol_abs, maxeval, maxtime);
}

nlopt_result nlopt_minimize(nlopt_algorithm algorithm, int n, nlopt_func_old f, void *f_data, const double *lb, const double *ub,
                            double *x, double *minf, double minf_max, double ftol_rel, double ftol_abs, double xtol_rel,
                            const double *xtol_abs, int maxeval, double maxtime)
{
    return nlopt_minimize_constrained(algorithm, n, f, f_data, 0, NULL, NULL, 0, lb, ub, x, minf,
                                      minf_max, ftol_rel, ftol_abs, xtol_rel, xtol_abs, maxeval, maxtime);
}
