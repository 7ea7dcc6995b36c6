"""G_MLSL_LDS + LD_LBFGS at the BASELINE config-4 shape: Ackley n=4096, 1000 samples/iteration (development timing tool).
usage: mlsl_bench.py [n] [samples] [maxeval] [cpu]   — `cpu` also times the real reference on the same run (bounded by maxeval)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nlopt_amd
import _oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
me = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
obj = "ackley"
L = nlopt_amd.lib()
xs, lo, hi = O.golden_x0(obj, n)
o = nlopt_amd.Opt(nlopt_amd.G_MLSL_LDS, n)
o.set_lower_bounds(lo); o.set_upper_bounds(hi)
o.set_min_objective(nlopt_amd.objective(obj))
loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n); loc.set_ftol_rel(1e-8)
L.nlopt_set_local_optimizer(o._h, loc._h)
o.set_population(ns); o.set_maxeval(me)
nlopt_amd.srand(42)
t0 = time.perf_counter()
x, minf, ret = o.optimize_raw(xs)
dt = time.perf_counter() - t0
st = o.stats()
print("gpu: ret", ret, "evals", o.get_numevals(), "minf %.15g" % minf, "wall %.3f s" % dt, "evals/s %.0f" % (o.get_numevals() / dt))
print({k: st[k] for k in ("generations", "evals_trial", "evals_mutation", "accepted", "t_eval_s", "t_evolve_s")}, o.get_errmsg())
if "cpu" in sys.argv:
    t0 = time.perf_counter()
    r = O.run_ref_mlsl(obj, n, ns, 42, alg=39, maxeval=me, record=True)
    dt = time.perf_counter() - t0
    print("cpu reference: ret", r["ret"], "evals", r["nevals"], "minf %.15g" % r["minf"], "wall %.3f s" % dt, "evals/s %.0f" % (r["nevals"] / dt))
