"""Time the REAL reference's NLOPT_GN_ISRES at BASELINE config 3's own population (pop = 5e4, n = 256, 4 inequality constraints) on this
machine's CPU, one thread: one full generation is ~85 s (the stochastic ranking is pop^2 serial steps, isres.c:207-228), so bench.py
does not time it in every run — it quotes the committed result (profiles/r03_isres_cpu_at_pop.json) with its label.
    python tools/isres_cpu_at_pop.py        (~4.5 minutes)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402

pop, n, ncon = 50000, 256, 4
res = {}
for gens in (2, 3):
    t0 = time.perf_counter()
    r = O.run_ref_isres("rastrigin", n, pop, 42, nineq=ncon, maxeval=gens * pop, record=False)
    res[gens] = (time.perf_counter() - t0, int(r["nevals"]))
per_gen = res[3][0] - res[2][0]
cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
out = dict(what="the REAL reference (oracle/_ref/libnlopt_ref.so) NLOPT_GN_ISRES rastrigin n=256 + 4 block-sum inequality constraints at the BENCHMARK's "
                "population pop=50000, seed 42, 1 thread",
           seconds_2_generations_of_evaluations=res[2][0], seconds_3_generations_of_evaluations=res[3][0],
           seconds_per_full_generation=per_gen, evals_per_s=pop / per_gen, cores=1, kind="reference",
           machine="build container (NOT the GPU box's host): " + cpu, measured_by="tools/isres_cpu_at_pop.py")
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_isres_cpu_at_pop.json"), "w"), indent=1)
print(json.dumps(out))
