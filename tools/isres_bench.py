"""ISRES at the BASELINE config-3 shape: Rastrigin n=256 + 4 block-sum inequality constraints, pop=5e4 (development timing tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import nlopt_amd
import _oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pop = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 3
xs, lo, hi = O.golden_x0("rastrigin", n)
o = nlopt_amd.Opt(nlopt_amd.GN_ISRES, n)
o.set_lower_bounds(lo); o.set_upper_bounds(hi)
o.set_min_objective(nlopt_amd.objective("rastrigin"))
o.add_blocksum_constraints(4, 1e-8)
o.set_population(pop); o.set_maxeval(gens * pop)
nlopt_amd.srand(42)
t0 = time.perf_counter()
x, minf, ret = o.optimize_raw(xs)
dt = time.perf_counter() - t0
st = o.stats()
print("ret", ret, "evals", o.get_numevals(), "minf", minf, "wall %.3f s" % dt, "evals/s %.0f" % (o.get_numevals() / dt))
print({k: st[k] for k in ("generations", "rank_sweeps", "t_eval_s", "t_rank_s", "t_evolve_s", "t_rng_s", "mt_words")}, o.get_errmsg())
