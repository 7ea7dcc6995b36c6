"""Every MLSL flavour with an explicitly set local optimiser that carries its own options (tolerances, initial step, evaluation
limit) — a client of `import nlopt` only; tests/test_python_module.py requires the same printout over both libraries."""
import nlopt, numpy as np, math
def bowl(x, grad):
    c = np.arange(1, x.size + 1) * 0.3
    if grad.size > 0:
        grad[:] = 2 * (x - c) - 0.4 * np.sin(4 * x)
    return float(np.sum((x - c) ** 2) + 0.1 * np.sum(np.cos(4 * x)))
for alg in (nlopt.G_MLSL, nlopt.G_MLSL_LDS, nlopt.GN_MLSL, nlopt.GD_MLSL_LDS):
    for loc in (nlopt.LN_COBYLA, nlopt.LD_MMA, nlopt.LD_LBFGS):
        nlopt.srand(3)
        o = nlopt.opt(alg, 3); o.set_min_objective(bowl); o.set_lower_bounds(-2.0); o.set_upper_bounds(3.0); o.set_maxeval(500)
        l = nlopt.opt(loc, 3); l.set_xtol_rel(1e-5); l.set_ftol_abs(1e-12); l.set_initial_step(0.3); l.set_maxeval(60)
        o.set_local_optimizer(l); o.set_population(6)
        try:
            x = o.optimize([0.1, 0.2, 0.3]); print(alg, loc, [repr(float(v)) for v in x], repr(o.last_optimum_value()), o.last_optimize_result(), o.get_numevals())
        except Exception as e: print(alg, loc, type(e).__name__, e)
