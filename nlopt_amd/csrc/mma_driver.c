/* mma_driver.c — NLOPT_LD_MMA behind the reference's entry point mma_minimize (src/algs/mma/mma.c:146-159) for
 * the case the stochastic-global path needs: no nonlinear constraints (m = 0).  That is what MLSL calls (it strips
 * the constraints of its local optimiser, options.c:824-846) and LD_MMA is the default local optimiser of
 * NLOPT_GD_MLSL(_LDS) (optimize.c:763-768, deprecated.c:28).  The optimisation runs in one launch of the batched
 * device kernel (hip/mma_kernels.hip): count = 1 from nlopt_optimize(LD_MMA), one workgroup per start from MLSL.
 *
 * Not provided, and refused with a message: nonlinear constraints (the dual problem then has variables and the
 * reference solves it with a nested optimiser), host-callback objectives, xtol_abs / x_weights, and a run whose only stopping
 * criterion is maxtime (the search is one kernel launch; the clock is watched between launches only). */
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
    const int obj = nlopt_amd_objective_id(f);
    nla_mma_params mma;
    nla_lbfgs_params prm;
    nla_lbfgs_result res;
    char err[200];
    int rc;
    (void) f_data;
    if ((rc = nla_mma_read_params(opt, &mma))) return (nlopt_result) rc;
    if (opt->m > 0) { nla_stop_msg(stop, "nlopt_amd: LD_MMA is provided without nonlinear constraints only (the MLSL local-search case)"); return NLOPT_INVALID_ARGS; }
    /* the whole search is one kernel launch: it cannot watch the wall clock, so something it can test must be able to end it
     * (MLSL gives its local optimiser tolerances when the caller did not, optimize.c:781-786) */
    if (stop->ftol_rel <= 0 && stop->ftol_abs <= 0 && stop->xtol_rel <= 0 && stop->maxeval <= 0 && !(stop->minf_max > -HUGE_VAL)) {
        nla_stop_msg(stop, "nlopt_amd: LD_MMA on the device needs a stopping criterion it can test (ftol, xtol_rel, maxeval or stopval); maxtime alone is not watched inside a search");
        return NLOPT_INVALID_ARGS;
    }
    if (nla_dev_count() <= 0) { nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
    if (obj < 0) { nla_stop_msg(stop, "nlopt_amd: LD_MMA is provided for device objectives (nlopt_amd_objective) only"); return NLOPT_INVALID_ARGS; }
    if (stop->xtol_abs || stop->x_weights) { nla_stop_msg(stop, "nlopt_amd: LD_MMA on the device does not take xtol_abs / x_weights"); return NLOPT_INVALID_ARGS; }
    memset(&prm, 0, sizeof prm);
    prm.minf_max = stop->minf_max; prm.ftol_rel = stop->ftol_rel; prm.ftol_abs = stop->ftol_abs; prm.xtol_rel = stop->xtol_rel;
    prm.maxeval = stop->maxeval;
    if (nla_local_run_batch(1, obj, n, 1, lb, ub, x, 0, &mma, opt->dx, &prm, &res, err, sizeof err)) { nla_stop_msg(stop, "device engine: %s", err); return NLOPT_FAILURE; }
    *minf = res.f;
    *stop->nevals_p += res.nevals;
    return (nlopt_result) res.ret;
}
