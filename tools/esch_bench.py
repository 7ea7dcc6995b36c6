"""NLOPT_GN_ESCH timing (development tool): usage esch_bench.py [n] [pop] [generations] [cpu]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nlopt_amd
import _oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 5
obj = "rastrigin"
no = int(pop * 1.5)
me = pop + gens * no
xs, lo, hi = O.golden_x0(obj, n)
o = nlopt_amd.Opt(nlopt_amd.GN_ESCH, n)
o.set_lower_bounds(lo); o.set_upper_bounds(hi)
o.set_min_objective(nlopt_amd.objective(obj))
o.set_population(pop); o.set_maxeval(me)
nlopt_amd.srand(42)
t0 = time.perf_counter()
x, minf, ret = o.optimize_raw(xs)
dt = time.perf_counter() - t0
st = o.stats()
print("gpu: ret", ret, "evals", o.get_numevals(), "minf %.15g" % minf, "wall %.3f s" % dt, "evals/s %.0f" % (o.get_numevals() / dt))
print({k: st[k] for k in ("generations", "t_eval_s", "t_rank_s", "t_evolve_s", "mt_words")}, o.get_errmsg())
if "cpu" in sys.argv:
    t0 = time.perf_counter()
    r = O.run_ref_esch(obj, n, pop, 42, maxeval=me, record=False)
    dt = time.perf_counter() - t0
    print("cpu reference: ret", r["ret"], "evals", r["nevals"], "minf %.15g" % r["minf"], "wall %.3f s" % dt, "evals/s %.0f" % (r["nevals"] / dt))
