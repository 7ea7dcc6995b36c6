"""child of tests/test_fault_injection.py: every allocation of a run over the emulated device is made to fail in turn; the run must
come back with a negative nlopt_result (and an errmsg) — never crash, never report success with a failed allocation behind it.
Prints one line per injected failure; the parent reads the last line if the process dies."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nlopt_amd  # noqa: E402
nlopt_amd.LIB_PATH = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
import _oracle as O  # noqa: E402


def make(alg):
    obj, n = "rastrigin", 6
    xs, lo, hi = O.golden_x0(obj, n)
    L = nlopt_amd.lib()
    grow = alg == "mlsl_grow"            # enough samples for the point set to be re-allocated twice (1024 -> 2048 -> 4096)
    if grow:
        alg = "mlsl"
    a = {"crs": nlopt_amd.GN_CRS2_LM, "isres": nlopt_amd.GN_ISRES, "esch": nlopt_amd.GN_ESCH, "mlsl": nlopt_amd.G_MLSL_LDS, "mlsl_mma": nlopt_amd.GD_MLSL,
         "lbfgs": nlopt_amd.LD_LBFGS, "mma": nlopt_amd.LD_MMA, "mma_con": nlopt_amd.LD_MMA, "auglag": 31}[alg]    # 31 = NLOPT_LD_AUGLAG
    o = nlopt_amd.Opt(a, n)
    o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj))
    o.set_maxeval(400)
    if alg == "isres":
        o.set_population(30); o.add_blocksum_constraints(2, 1e-8)
    if alg in ("crs", "esch"):
        o.set_population(40)
    if alg == "mlsl":
        loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
        loc.set_ftol_rel(1e-6)
        L.nlopt_set_local_optimizer(o._h, loc._h)
        o.set_population(700 if grow else 10)
        if grow:
            o.set_maxeval(2600)
    if alg == "mlsl_mma":
        o.set_ftol_rel(1e-6); o.set_population(10)
    if alg in ("lbfgs", "mma"):
        o.set_ftol_rel(1e-8)
    if alg == "auglag":                  # the penalised problem goes to the default LD_MMA: one device context per outer iteration
        o.add_blocksum_constraints(2, 1e-8); o.set_ftol_rel(1e-4); o.set_maxeval(40)
    if alg == "mma_con":                 # nonlinear constraints: host outer algorithm, one device context per dual problem
        o.add_blocksum_constraints(2, 1e-8); o.set_ftol_rel(1e-6); o.set_maxeval(12)
    return o, xs


def main():
    alg = sys.argv[1]
    L = nlopt_amd.lib()
    L.orc_emu_fail_alloc_at.argtypes = [C.c_long]
    L.orc_emu_allocs.restype = C.c_long
    L.orc_emu_live.restype = C.c_long

    def leaked(o):
        """device-layer objects still alive after the optimiser object is destroyed"""
        L.nlopt_destroy(o._h)
        o._h = None
        return L.orc_emu_live()
    o, xs = make(alg)
    nlopt_amd.srand(1)
    L.orc_emu_fail_alloc_at(0)
    x, minf, ret0 = o.optimize_raw(xs)
    total = L.orc_emu_allocs()
    assert ret0 > 0 and total > 0, (ret0, total)
    print("baseline", alg, ret0, total, flush=True)
    bad = []
    if leaked(o):
        bad.append(("leak after a successful run", L.orc_emu_live()))
    for k in range(1, total + 1):
        o, xs = make(alg)
        nlopt_amd.srand(1)
        print("inject", alg, k, flush=True)
        L.orc_emu_fail_alloc_at(k)
        x, minf, ret = o.optimize_raw(xs)
        hit = L.orc_emu_allocs() >= k
        L.orc_emu_fail_alloc_at(0)
        if hit and ret > 0:
            bad.append((k, ret))
        if hit and ret < 0 and not o.get_errmsg():
            bad.append((k, ret, "no errmsg"))
        if leaked(o):
            bad.append((k, ret, "leak", L.orc_emu_live()))
    # the same for kernel launches: the k-th launcher call returns an error
    L.orc_emu_fail_launch_at.argtypes = [C.c_long]
    L.orc_emu_launches.restype = C.c_long
    o, xs = make(alg)
    nlopt_amd.srand(1)
    L.orc_emu_fail_launch_at(0)
    x, minf, ret0 = o.optimize_raw(xs)
    nl = L.orc_emu_launches()
    print("baseline launches", alg, ret0, nl, flush=True)
    step = max(1, nl // 150)                      # long runs: every step-th launch, plus the first 40
    for k in sorted(set(list(range(1, min(nl, 40) + 1)) + list(range(1, nl + 1, step)))):
        o, xs = make(alg)
        nlopt_amd.srand(1)
        print("inject launch", alg, k, flush=True)
        L.orc_emu_fail_launch_at(k)
        x, minf, ret = o.optimize_raw(xs)
        hit = L.orc_emu_launches() >= k
        L.orc_emu_fail_launch_at(0)
        if hit and ret > 0:
            bad.append(("launch", k, ret))
        if hit and ret < 0 and not o.get_errmsg():
            bad.append(("launch", k, ret, "no errmsg"))
        if leaked(o):
            bad.append(("launch", k, ret, "leak", L.orc_emu_live()))
    print("done", alg, total, bad, flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
