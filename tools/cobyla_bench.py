"""(1) hip/cobyla_kernels.hip alone: a batch of LN_COBYLA searches (one wavefront each) against the real reference running the same starts one
after another on one host core — evaluations per second of both, results compared.  (2) GN_MLSL with its default local optimiser
LN_COBYLA on a compiled-in device objective: the searches batched on the device (round 6) against the same run with COBYLA as a host
algorithm ("amd_cobyla_host" = 1, rounds 2-5).  python tools/cobyla_bench.py [obj n samples maxeval] -> profiles/r06_cobyla_batched.txt"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nlopt_amd  # noqa: E402
import _oracle as O  # noqa: E402


def run(obj, n, ns, maxeval, host):
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(nlopt_amd.GN_MLSL_LDS, n)
    o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj))
    o.set_population(ns); o.set_maxeval(maxeval); o.set_xtol_rel(1e-6)
    if host:
        o.set_param("amd_cobyla_host", 1)
    nlopt_amd.srand(42)
    t0 = time.perf_counter()
    x, minf, ret = o.optimize_raw(xs)
    dt = time.perf_counter() - t0
    st = o.stats()
    return dict(mode="COBYLA on the host" if host else "COBYLA batched on the device", obj=obj, n=n, samples=ns, ret=ret, minf=minf, evals=o.get_numevals(),
                seconds=dt, evals_per_s=o.get_numevals() / dt, local_searches=int(st["accepted"]), launches=int(st["lbfgs_launches"]),
                launch_ms=st["t_lbfgs_ms"])


def kernel_lines():
    import numpy as np
    import test_gpu_cobyla as G
    for obj, n, count, maxeval in [("rosenbrock", 8, 1, 4000), ("rosenbrock", 8, 64, 4000), ("rosenbrock", 8, 2048, 4000), ("rosenbrock", 16, 1, 4000), ("rosenbrock", 16, 2048, 4000),
                                   ("rosenbrock", 32, 1, 4000), ("rosenbrock", 32, 512, 4000), ("sphere", 48, 256, 4000)]:
        rng = np.random.default_rng(5)
        lo, hi = nlopt_amd.objective_box(obj)
        lb, ub = np.full(n, lo), np.full(n, hi)
        starts = rng.uniform(lo, hi, (count, n))
        G.kernel_batch(obj, n, starts[:1], lb, ub, maxeval=50)              # warm-up
        t0 = time.perf_counter()
        a = G.kernel_batch(obj, n, starts, lb, ub, maxeval=maxeval)
        td = time.perf_counter() - t0
        nref = min(count, 16)
        t0 = time.perf_counter()
        r = G.reference_cobyla(obj, n, starts[:nref], lb, ub, maxeval=maxeval)
        tr = time.perf_counter() - t0
        same = a["ret"][:nref] == r["ret"] and a["nevals"][:nref] == r["nevals"] and np.array_equal(a["f"][:nref], r["f"]) and np.array_equal(a["x"][:nref], r["x"])
        print(json.dumps(dict(mode="kernel", obj=obj, n=n, searches=count, evals=int(sum(a["nevals"])), device_s=td, device_evals_per_s=sum(a["nevals"]) / td,
                              device_us_per_eval_of_one_search=1e6 * td / max(a["nevals"]), reference_searches_timed=nref,
                              reference_evals_per_s_one_core=sum(r["nevals"]) / tr, identical_to_reference=bool(same))), flush=True)


if __name__ == "__main__":
    if len(sys.argv) <= 4:
        kernel_lines()
    cases = [("rosenbrock", 8, 256, 200000), ("rastrigin", 16, 512, 400000), ("ackley", 32, 512, 600000), ("griewank", 48, 256, 600000)]
    if len(sys.argv) > 4:
        cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))]
    for obj, n, ns, me in cases:
        run(obj, n, ns, min(me, 20000), False)            # warm-up (module load, allocations)
        for host in (False, True):
            print(json.dumps(run(obj, n, ns, me, host)), flush=True)
