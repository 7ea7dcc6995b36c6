"""-m gpu: NLOPT_LD_MMA without nonlinear constraints (batched device kernel, one workgroup per start) and the MLSL
variants that use it — GD_MLSL / GD_MLSL_LDS with their DEFAULT local optimiser (optimize.c:763-777), G_MLSL with an
explicit LD_MMA — against the CPU oracle (oracle/port_mma.c, port_mlsl.c; pinned bit-exactly to the real reference in
test_oracle_pins.py).

The device sums the approximation's value gval and the x-tolerance norms in a fixed tree order and its objective
differs from the host's by libm/reduction rounding, so iterates agree to rounding, not bitwise: the bar is the same
result code, the same minimum within the run's tolerance and an evaluation count that matches unless a rounding-level
difference flipped a "gval >= f" decision (slack stated per case)."""
import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = pytest.mark.gpu


def run_amd(obj, n, x0=None, lb=None, ub=None, maxeval=0, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0, stopval=None, params=None, step=None):
    assert nlopt_amd.device_count() > 0
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(nlopt_amd.LD_MMA, n)
    o.set_lower_bounds(lo if lb is None else lb)
    o.set_upper_bounds(hi if ub is None else ub)
    o.set_min_objective(nlopt_amd.objective(obj))
    if maxeval:
        o.set_maxeval(maxeval)
    if ftol_rel:
        o.set_ftol_rel(ftol_rel)
    if ftol_abs:
        o.set_ftol_abs(ftol_abs)
    if xtol_rel:
        o.set_xtol_rel(xtol_rel)
    if stopval is not None:
        o.set_stopval(stopval)
    for k, v in (params or {}).items():
        o.set_param(k, v)
    if step is not None:
        assert nlopt_amd.lib().nlopt_set_initial_step1(o._h, float(step)) > 0
    x, minf, ret = o.optimize_raw(xs if x0 is None else x0)
    return dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), err=o.get_errmsg())


@pytest.mark.parametrize("obj,n,kw,slack", [
    ("sphere", 8, dict(ftol_rel=1e-10), 2),
    ("rosenbrock", 10, dict(maxeval=60), 0),            # longer Rosenbrock runs amplify rounding differences chaotically
    ("ackley", 30, dict(ftol_rel=1e-8), 2),
    ("rastrigin", 20, dict(ftol_rel=1e-8), 2),
    ("griewank", 12, dict(xtol_rel=1e-6), 4),
    ("levy", 7, dict(ftol_abs=1e-12), 4),
    ("ackley", 200, dict(ftol_rel=1e-8), 2),
    ("rastrigin", 64, dict(maxeval=37), 0),
    ("sphere", 6, dict(stopval=1e-3), 0),
    ("rastrigin", 16, dict(ftol_rel=1e-9, params=dict(inner_gradients=0)), 2),
    ("ackley", 10, dict(ftol_rel=1e-9, params=dict(inner_maxeval=2, rho_init=0.01)), 2),
    ("ackley", 10, dict(ftol_rel=1e-9, params=dict(inner_gradients=0, inner_maxeval=1)), 2),      # mma.c:343: the inner limit does not end the step here
    ("griewank", 10, dict(xtol_rel=1e-8, params=dict(sigma_min=0.5)), 8),
    ("rastrigin", 12, dict(ftol_rel=1e-9, step=0.3), 2),
    ("ackley", 4096, dict(ftol_rel=1e-8), 3),           # the config-4 shape
    ("rastrigin", 1000, dict(ftol_rel=1e-9), 3),
])
def test_mma_matches_oracle(obj, n, kw, slack):
    kw = dict(kw)
    kw.setdefault("maxeval", 20000)        # never binds in these cases; bounds the run if the device path went astray
    a = run_amd(obj, n, **kw)
    p = O.run_port_mma(obj, n, **kw)
    assert a["ret"] == p["ret"], (a, p["ret"])
    assert abs(a["nevals"] - p["nevals"]) <= slack, (a["nevals"], p["nevals"])
    scale = max(abs(p["minf"]), 1e-300)
    assert abs(a["minf"] - p["minf"]) <= 1e-8 * scale + 1e-12, (a["minf"], p["minf"])
    assert np.allclose(a["x"], p["x"], rtol=1e-6, atol=1e-7 * max(np.abs(p["x"]).max(), 1.0))


def test_mma_always_improve_off_follows_the_original_acceptance_rule():
    """always_improve = 0: the point only moves at the end of an outer iteration (mma.c:326-331)"""
    kw = dict(maxeval=80, params=dict(always_improve=0))
    a = run_amd("rosenbrock", 6, **kw)
    p = O.run_port_mma("rosenbrock", 6, **kw)
    assert a["ret"] == p["ret"] == nlopt_amd.MAXEVAL_REACHED and a["nevals"] == p["nevals"] == 80
    assert abs(a["minf"] - p["minf"]) <= 1e-6 * max(abs(p["minf"]), 1e-300)


def test_mma_active_bounds_and_fixed_coordinate():
    n = 9
    lb, ub = np.full(n, -2.0), np.full(n, 3.0)
    lb[2], ub[2] = 0.7, 3.0            # the minimiser's coordinate 2 lies on this bound
    lb[5] = ub[5] = 1.25               # sigma = 0: the coordinate never moves (mma.c:91-94)
    x0 = np.linspace(0.9, 2.6, n)
    x0[5] = 1.25
    kw = dict(x0=x0, lb=lb, ub=ub, ftol_rel=1e-10, maxeval=5000)
    a = run_amd("sphere", n, **kw)
    p = O.run_port_mma("sphere", n, **kw)
    assert a["ret"] == p["ret"]
    assert abs(a["nevals"] - p["nevals"]) <= 2
    assert a["x"][5] == 1.25 and abs(a["x"][2] - 0.7) < 1e-6
    assert np.allclose(a["x"], p["x"], atol=1e-6)


def test_mma_refusals_are_loud():
    o = nlopt_amd.Opt(nlopt_amd.LD_MMA, 3)
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    calls = []

    def f(x, g):                                                    # an ordinary host callback is served (device coroutine)
        calls.append(bool(g.size))
        if g.size:
            g[:] = 2 * x
        return float(np.sum(x * x))
    o.set_min_objective(f)
    o.set_maxeval(10)
    x, minf, ret = o.optimize_raw(np.full(3, 0.5))
    assert ret > 0 and o.get_numevals() == len(calls) and all(calls)
    o2 = nlopt_amd.Opt(nlopt_amd.LD_MMA, 3)
    o2.set_lower_bounds(-1.0)
    o2.set_upper_bounds(1.0)
    o2.set_min_objective(nlopt_amd.objective("sphere"))
    o2.set_param("inner_gradients", 2)
    x, minf, ret = o2.optimize_raw(np.full(3, 0.5))
    assert ret == nlopt_amd.INVALID_ARGS and "inner_gradients must be 0 or 1" in o2.get_errmsg()


# ---- MLSL with LD_MMA ----------------------------------------------------------------------------------------------
def run_mlsl(obj, n, nsamples, seed, alg, local, maxeval=0, stopval=None, tol=1e-8, local_params=None):
    """local = "mma": explicit LD_MMA local optimiser with ftol_rel = tol; None: no local optimiser, the tolerance goes on
    the global object and the dispatcher builds its default (LD_MMA for the GD variants)"""
    L = nlopt_amd.lib()
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective(obj))
    if local == "mma":
        loc = nlopt_amd.Opt(nlopt_amd.LD_MMA, n)
        loc.set_ftol_rel(tol)
        for k, v in (local_params or {}).items():
            loc.set_param(k, v)
        assert L.nlopt_set_local_optimizer(o._h, loc._h) > 0
    else:
        o.set_ftol_rel(tol)
    if nsamples:
        o.set_population(nsamples)
    if maxeval:
        o.set_maxeval(maxeval)
    if stopval is not None:
        o.set_stopval(stopval)
    o.enable_trace((maxeval or 200000) + 4096)
    nlopt_amd.srand(seed)
    x, minf, ret = o.optimize_raw(xs)
    return dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), stats=o.stats(), err=o.get_errmsg(), trace=o.trace())


@pytest.mark.parametrize("alg,lds,local,obj,n,ns,seed,kw", [
    (nlopt_amd.GD_MLSL, False, None, "sphere", 5, 6, 2, dict(stopval=1e-9, maxeval=5000)),
    (nlopt_amd.G_MLSL_LDS, True, "mma", "sphere", 7, 0, 3, dict(stopval=1e-9, maxeval=5000)),
    (nlopt_amd.GD_MLSL, False, None, "rastrigin", 4, 10, 42, dict(maxeval=2000)),
    (nlopt_amd.GD_MLSL_LDS, True, None, "rastrigin", 4, 10, 42, dict(maxeval=2000)),
    (nlopt_amd.G_MLSL, False, "mma", "ackley", 6, 25, 7, dict(maxeval=3000)),
    (nlopt_amd.GD_MLSL, False, None, "griewank", 30, 40, 9, dict(maxeval=4000)),
])
def test_mlsl_with_mma_reaches_the_oracles_result(alg, lds, local, obj, n, ns, seed, kw):
    a = run_mlsl(obj, n, ns, seed, alg, local, **kw)
    p = O.run_port_mlsl(obj, n, ns, seed, local="mma", local_ftol_rel=1e-8, lds=lds, **kw)
    assert a["ret"] == p["ret"], (a["err"], a["ret"], p["ret"])
    assert abs(a["minf"] - p["minf"]) <= 1e-6 * max(abs(p["minf"]), 1.0)
    nloc = len(p["floc"])
    fs = a["trace"][a["trace"]["kind"] == 3]["f"]
    m = min(len(fs), len(p["fsamp"]), ns or 4)          # the first iteration's samples do not depend on any local search
    assert np.all(np.abs(fs[:m] - p["fsamp"][:m]) <= 1e-10 * np.maximum(np.abs(p["fsamp"][:m]), 1.0))
    if p["ret"] == nlopt_amd.MAXEVAL_REACHED:
        # the budget ran out: each device search may differ from the oracle's by an evaluation or two (summation
        # order), which moves the cut-off by a few samples
        assert a["nevals"] == p["nevals"] == kw["maxeval"]
        slack = 3 * nloc + 2
        assert abs(a["stats"]["mt_words"] - p["words"]) <= 2 * n * slack
        assert abs(a["stats"]["accepted"] - nloc) <= max(2, nloc // 20)
    else:
        assert a["stats"]["mt_words"] == p["words"]                  # same number of samples drawn (0 in Sobol mode)
        assert len(fs) == len(p["fsamp"])
        assert np.all(np.abs(fs - p["fsamp"]) <= 1e-10 * np.maximum(np.abs(p["fsamp"]), 1.0))
        fl = a["trace"][a["trace"]["kind"] == 4]
        assert len(fl) == nloc
        assert np.all(np.abs(fl["f"] - p["floc"]) <= 1e-7 * np.maximum(np.abs(p["floc"]), 1.0))
        assert np.all(np.abs(fl["accepted"] - p["eloc"]) <= 3)


def test_gd_mlsl_default_runs_out_of_evaluations_like_the_oracle():
    """MAXEVAL inside a local search: the search's own limit is what is left of the global budget (optimize.c:1097-1100)"""
    a = run_mlsl("rastrigin", 6, 20, 11, nlopt_amd.GD_MLSL, None, maxeval=3000, tol=1e-7)
    p = O.run_port_mlsl("rastrigin", 6, 20, 11, local="mma", local_ftol_rel=1e-7, maxeval=3000)
    assert a["ret"] == p["ret"] == nlopt_amd.MAXEVAL_REACHED
    assert a["nevals"] == p["nevals"] == 3000
    assert abs(a["minf"] - p["minf"]) <= 1e-6 * max(abs(p["minf"]), 1.0)


def test_mlsl_counts_the_uncounted_gradient_calls_of_inner_gradients_0():
    """inner_gradients = 0: MMA re-evaluates with a gradient without counting (mma.c:336-339), MLSL's wrapper counts the call"""
    kw = dict(maxeval=1500)
    a = run_mlsl("rastrigin", 5, 12, 5, nlopt_amd.G_MLSL, "mma", local_params=dict(inner_gradients=0), **kw)
    p = O.run_port_mlsl("rastrigin", 5, 12, 5, local="mma", local_ftol_rel=1e-8, local_params=dict(inner_gradients=0), **kw)
    assert a["ret"] == p["ret"] == nlopt_amd.MAXEVAL_REACHED
    assert abs(a["nevals"] - p["nevals"]) <= 2            # the last search may overshoot by its uncounted call, on both sides
    assert abs(a["minf"] - p["minf"]) <= 1e-6 * max(abs(p["minf"]), 1.0)


def test_gn_mlsl_default_local_optimizer_is_served():
    """GN_MLSL's default local optimiser is LN_COBYLA (optimize.c:766-768): a host algorithm (cobyla_host.c) under the device's
    sampling and bookkeeping — tests/test_gpu_cobyla.py compares such runs with the real reference call by call"""
    o = nlopt_amd.Opt(nlopt_amd.GN_MLSL, 3)
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    o.set_min_objective(nlopt_amd.objective("sphere"))
    o.set_maxeval(100)
    x, minf, ret = o.optimize_raw(np.full(3, 0.5))
    assert ret == nlopt_amd.MAXEVAL_REACHED and o.get_numevals() >= 100 and minf < 1e-3, (ret, minf, o.get_errmsg())
