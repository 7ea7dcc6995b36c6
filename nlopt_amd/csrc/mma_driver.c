/* mma_driver.c — NLOPT_LD_MMA behind the reference's entry point mma_minimize (src/algs/mma/mma.c:146-159) for
 * the case the stochastic-global path needs: no nonlinear constraints (m = 0).  That is what MLSL calls (it strips
 * the constraints of its local optimiser, options.c:824-846) and LD_MMA is the default local optimiser of
 * NLOPT_GD_MLSL(_LDS) (optimize.c:763-768, deprecated.c:28).  The optimisation runs in one launch of the batched
 * device kernel (hip/mma_kernels.hip): count = 1 from nlopt_optimize(LD_MMA), one workgroup per start from MLSL.
 *
 * Objectives: compiled-in device objectives (the whole search is one launch), user device objectives and ordinary host
 * callbacks (the kernel runs as a coroutine, lbfgs_driver.c).  maxtime and nlopt_force_stop are observed inside a search
 * (mma.c:258-260,394-396).  With nonlinear constraints the dual problem has variables: the outer algorithm then runs on the host
 * (mma_host.c) and solves each dual problem through this file, as the reference solves it through a nested LD_MMA. */
#include "nla_internal.h"
#include <math.h>
#include <stdio.h>
#include <string.h>

/* the algorithm's parameters as the dispatcher reads and validates them (optimize.c:798-815) */
int nla_mma_read_params(nlopt_opt opt, nla_mma_params *out)
{
    const double rho_init = nlopt_get_param(opt, "rho_init", 1.0), sigma_min = nlopt_get_param(opt, "sigma_min", 0.0);
    const int inner_maxeval = (int) nlopt_get_param(opt, "inner_maxeval", 0);
    const int inner_gradients = (int) nlopt_get_param(opt, "inner_gradients", 1);
    const int always_improve = (int) nlopt_get_param(opt, "always_improve", 1);
    if (!(rho_init > 0) && !isinf(rho_init)) { nla_set_errmsg(opt, "rho_init must be positive and finite"); return NLOPT_INVALID_ARGS; }
    if (inner_gradients != 0 && inner_gradients != 1) { nla_set_errmsg(opt, "inner_gradients must be 0 or 1"); return NLOPT_INVALID_ARGS; }
    if (always_improve != 0 && always_improve != 1) { nla_set_errmsg(opt, "always_improve must be 0 or 1"); return NLOPT_INVALID_ARGS; }
    if (sigma_min < 0.0) { nla_set_errmsg(opt, "sigma_min must be non-negative"); return NLOPT_INVALID_ARGS; }
    memset(out, 0, sizeof *out);
    out->rho_init = rho_init; out->sigma_min = sigma_min;
    out->inner_maxeval = inner_maxeval; out->inner_gradients = inner_gradients; out->always_improve = always_improve;
    return 0;
}

nlopt_result nla_mma_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x,
                              double *minf, nla_stopping *stop)
{
    nla_evaluator ev;
    nla_mma_params mma;
    nla_lbfgs_params prm;
    nla_lbfgs_result res;
    char err[200];
    int rc;
    if ((rc = nla_mma_read_params(opt, &mma))) return (nlopt_result) rc;
    if (nla_dev_count() <= 0) { nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
    if (opt->m > 0) return nla_mma_constrained(opt, (unsigned) n, f, f_data, lb, ub, x, minf, stop, &mma);   /* mma_host.c */
    nla_evaluator_resolve(&ev, opt, f, f_data);
    memset(&prm, 0, sizeof prm);
    prm.minf_max = stop->minf_max; prm.ftol_rel = stop->ftol_rel; prm.ftol_abs = stop->ftol_abs; prm.xtol_rel = stop->xtol_rel;
    prm.maxeval = stop->maxeval;
    if (nla_local_run_batch(1, &ev, n, 1, lb, ub, x, 0, &mma, opt->dx, &prm, &res, stop, nla_exact_mode_for(opt, NULL, &ev),
                            ev.kind == NLA_EVAL_HOST ? stop->nevals_p : NULL, opt, err, sizeof err)) { nla_stop_msg(stop, "device engine: %s", err); return NLOPT_FAILURE; }
    *minf = res.f;
    if (ev.kind != NLA_EVAL_HOST) *stop->nevals_p += res.nevals;
    return (nlopt_result) res.ret;
}
