"""-m gpu: NLOPT_LD_LBFGS (batched device PLIS, one workgroup per start) against the CPU oracle
(oracle/port_lbfgs.c, pinned bit-exactly to the real reference).  The device sums dot products in a
different order, so iterates agree to rounding, not bitwise: the bar is the same result code, the
same minimiser / minimum within the run's own tolerance, and an evaluation count that matches
unless a rounding-level difference moved a line-search decision (allowed slack stated per case)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = pytest.mark.gpu


def run_amd(obj, n, x0=None, lb=None, ub=None, maxeval=0, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0, stopval=None, mf=0):
    assert nlopt_amd.device_count() > 0
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
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
    if mf:
        nlopt_amd.lib().nlopt_set_vector_storage(o._h, mf)
    x, minf, ret = o.optimize_raw(xs if x0 is None else x0)
    return dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), err=o.get_errmsg())


@pytest.mark.parametrize("obj,n,kw,slack", [
    ("sphere", 8, dict(), 0),
    ("rosenbrock", 10, dict(maxeval=2000), 2),
    ("rosenbrock", 2, dict(ftol_rel=1e-10), 2),
    ("ackley", 30, dict(ftol_rel=1e-8), 2),
    ("rastrigin", 20, dict(ftol_rel=1e-8), 2),
    ("griewank", 12, dict(xtol_rel=1e-6), 2),
    ("levy", 7, dict(ftol_abs=1e-12), 2),
    ("ackley", 200, dict(ftol_rel=1e-8, mf=5), 3),
    ("rastrigin", 64, dict(maxeval=37), 0),
    ("sphere", 6, dict(stopval=1e-3), 0),
    ("ackley", 4096, dict(ftol_rel=1e-8), 4),          # the config-4 shape: n = 4096, 320 history pairs
    ("rastrigin", 1000, dict(ftol_rel=1e-9), 4),
])
def test_lbfgs_matches_oracle(obj, n, kw, slack):
    a = run_amd(obj, n, **kw)
    p = O.run_port_lbfgs(obj, n, **kw)
    assert a["ret"] == p["ret"], (a, p["ret"])
    assert abs(a["nevals"] - p["nevals"]) <= slack, (a["nevals"], p["nevals"])
    scale = max(abs(p["minf"]), 1e-300)
    assert abs(a["minf"] - p["minf"]) <= 1e-8 * scale + 1e-12, (a["minf"], p["minf"])
    assert np.allclose(a["x"], p["x"], rtol=1e-6, atol=1e-7 * max(np.abs(p["x"]).max(), 1.0))


def test_lbfgs_active_bounds():
    rng = np.random.default_rng(3)
    for n, obj in ((9, "sphere"), (14, "rastrigin"), (6, "rosenbrock")):
        lb = -rng.uniform(0.1, 2.0, n)
        ub = rng.uniform(0.1, 2.0, n)
        lb[::3] = 0.3
        ub[::3] = 2.5
        x0 = np.clip(rng.uniform(-2, 2, n), lb, ub)
        x0[1] = ub[1]
        a = run_amd(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        p = O.run_port_lbfgs(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        assert a["ret"] == p["ret"] and abs(a["nevals"] - p["nevals"]) <= 2
        assert abs(a["minf"] - p["minf"]) <= 1e-9 * max(abs(p["minf"]), 1.0)
        assert np.allclose(a["x"], p["x"], rtol=1e-7, atol=1e-9)
        assert np.array_equal(a["x"][::3], p["x"][::3])          # coordinates on their bounds: exactly the bound


def test_lbfgs_host_callback_is_served():
    """an ordinary nlopt_func (here a Python callback) is served: the search runs on the device as a coroutine, f and the
    gradient are computed on the caller's thread (tests/test_gpu_host_callbacks.py compares with the reference call by call)"""
    calls = []

    def f(x, g):
        calls.append(x.copy())
        if g.size:
            g[:] = 2 * (x - 0.25)
        return float(np.sum((x - 0.25) ** 2))
    o = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, 3)
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    o.set_min_objective(f)
    o.set_ftol_rel(1e-12)
    x, minf, ret = o.optimize_raw(np.full(3, 0.5))
    assert ret > 0 and minf < 1e-20 and np.allclose(x, 0.25) and o.get_numevals() == len(calls) >= 2
