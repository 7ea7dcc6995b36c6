"""NLOPT_LD_MMA with nonlinear inequality constraints (mma_host.c: the outer algorithm on the host, every dual problem solved
through the library's own LD_MMA) against the REAL reference, call by call: drawn problems (dimension, boxes incl. open ones,
scalar and vector constraints, rows that return NaN = switched off, infeasible starts, every algorithm parameter incl. the
dual solver's, every stopping criterion, maximisation, initial steps) are given to both libraries through Python callbacks;
the point of EVERY objective and constraint call, whether a gradient was asked for, the results, the counts and the messages
must be identical.  The product runs over the emulated device layer here."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from test_api_differential import EMU, FUNC, MFUNC, vp, dpp
from test_cobyla_differential import more_bind, dp

pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")
LD_MMA, LD_LBFGS = 24, 11


def bind(L):
    more_bind(L)
    L.nlopt_set_param.argtypes = [vp, C.c_char_p, C.c_double]
    L.nlopt_set_local_optimizer.argtypes = [vp, vp]
    return L


def play(L, draw):
    rng = np.random.default_rng(83000 + draw)
    n = int(rng.integers(1, 6))
    calls = []
    opt = L.nlopt_create(LD_MMA, n)
    lb, ub = np.full(n, -2.0) - rng.random(n), np.full(n, 3.0) + rng.random(n)
    r = rng.random()
    if r < 0.15:
        lb[int(rng.integers(n))] = -np.inf
    elif r < 0.3:
        ub[:] = np.inf
    elif r < 0.4 and n > 1:
        lb[0] = ub[0] = 0.25                                   # a fixed coordinate: sigma = 0
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub))]
    centre = rng.uniform(-1, 2, n)
    kind = int(rng.integers(3))

    def f(nn, x, g, d):
        xs = np.array([x[i] for i in range(nn)])
        calls.append(np.concatenate(([0.0, 1.0 if g else 0.0], xs)))
        if kind == 0:
            val, grad = np.sum((xs - centre) ** 2 * (1 + 0.3 * np.arange(nn))), 2 * (xs - centre) * (1 + 0.3 * np.arange(nn))
        elif kind == 1:
            val, grad = np.sum(np.cosh(0.7 * (xs - centre))), 0.7 * np.sinh(0.7 * (xs - centre))
        else:
            val, grad = np.sum((xs - centre) ** 4) + np.sum(xs), 4 * (xs - centre) ** 3 + 1.0
        if g:
            for i in range(nn):
                g[i] = grad[i]
        return float(val)
    fcb = FUNC(f)
    keep = [fcb]
    maximise = rng.random() < 0.12
    if maximise:
        def negated(nn, x, g, d):
            v = f(nn, x, g, d)
            if g:
                for i in range(nn):
                    g[i] = -g[i]
            return -v
        fneg = FUNC(negated)
        keep.append(fneg)
        log.append(L.nlopt_set_max_objective(opt, C.cast(fneg, vp), None))
    else:
        log.append(L.nlopt_set_min_objective(opt, C.cast(fcb, vp), None))
    nscalar = int(rng.integers(0, 3))
    nan_row = rng.random() < 0.15
    for q in range(nscalar):
        cq = float(rng.uniform(-1.5, 1.0))                      # cq > about 0.5: infeasible at many starts

        def c(nn, x, g, d, cq=cq, q=q):
            xs = np.array([x[i] for i in range(nn)])
            calls.append(np.concatenate(([1.0 + q, 1.0 if g else 0.0], xs)))
            a, b = q % nn, (q + 1) % nn
            if g:
                for i in range(nn):
                    g[i] = 0.0
                g[a] += 2 * xs[a]
                g[b] += 1.0
            if nan_row and q == 0 and xs[a] > 1.0:
                return float("nan")
            return float(xs[a] ** 2 + xs[b] - 2 + cq)
        cb = FUNC(c)
        keep.append(cb)
        log.append(L.nlopt_add_inequality_constraint(opt, C.cast(cb, vp), None, float(rng.choice([0.0, 1e-8, 1e-3]))))
    if rng.random() < 0.4 or nscalar == 0:
        m = int(rng.integers(1, 4))

        def mf(mm, res, nn, x, g, d):
            xs = np.array([x[i] for i in range(nn)])
            calls.append(np.concatenate(([9.0, 1.0 if g else 0.0], xs)))
            for i in range(mm):
                res[i] = float(xs[i % nn] + 0.5 * xs[(i + 1) % nn] ** 2 - 2.0 - 0.3 * i)
                if g:
                    for j in range(nn):
                        g[i * nn + j] = 0.0
                    g[i * nn + i % nn] += 1.0
                    g[i * nn + (i + 1) % nn] += xs[(i + 1) % nn]
        mcb = MFUNC(mf)
        keep.append(mcb)
        tol = np.full(m, float(rng.choice([0.0, 1e-6])))
        log.append(L.nlopt_add_inequality_mconstraint(opt, m, C.cast(mcb, vp), None, dp(tol)))
    for name, values, p in (("inner_maxeval", [1, 2, 5], 0.2), ("inner_gradients", [0, 1], 0.3), ("always_improve", [0, 1], 0.3),
                            ("rho_init", [0.1, 10.0], 0.2), ("sigma_min", [1e-3, 0.05], 0.15), ("dual_ftol_rel", [1e-6, 1e-10], 0.2),
                            ("dual_maxeval", [5, 50], 0.15), ("dual_xtol_rel", [1e-6], 0.1), ("dual_ftol_abs", [1e-9], 0.1)):
        if rng.random() < p:
            log.append(L.nlopt_set_param(opt, name.encode(), float(rng.choice(values))))
    lo = None
    if rng.random() < 0.1:                                      # a local optimiser object hands its settings to the dual solver
        lo = L.nlopt_create(LD_MMA, n)
        L.nlopt_set_ftol_rel(lo, 1e-9)
        L.nlopt_set_maxeval(lo, 200)
        log.append(L.nlopt_set_local_optimizer(opt, lo))
    r = rng.random()
    if r < 0.2:
        log.append(L.nlopt_set_initial_step(opt, dp(rng.uniform(0.05, 1.5, n))))
    elif r < 0.3:
        log.append(L.nlopt_set_initial_step1(opt, float(rng.uniform(0.05, 1.0))))
    if rng.random() < 0.7:
        log.append(L.nlopt_set_xtol_rel(opt, float(rng.choice([1e-2, 1e-4, 1e-8]))))
    if rng.random() < 0.3:
        log.append(L.nlopt_set_ftol_rel(opt, float(rng.choice([1e-3, 1e-7]))))
    if rng.random() < 0.15:
        log.append(L.nlopt_set_ftol_abs(opt, 1e-6))
    if rng.random() < 0.15:
        log.append(L.nlopt_set_xtol_abs(opt, dp(np.full(n, 1e-4))))
    if rng.random() < 0.15:
        log.append(L.nlopt_set_stopval(opt, float(rng.uniform(-3, 6))))
    log.append(L.nlopt_set_maxeval(opt, int(rng.choice([1, 4, 25, 120, 400]))))
    x = np.clip(rng.uniform(-1.5, 2.5, n), lb, ub)
    minf = C.c_double(0)
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    out = dict(log=log, ret=ret, minf=minf.value, x=x.copy(), nev=L.nlopt_get_numevals(opt), msg=L.nlopt_get_errmsg(opt),
               calls=np.array(calls) if calls else np.zeros((0, n + 2)))
    L.nlopt_destroy(opt)
    if lo:
        L.nlopt_destroy(lo)
    return out


def same(a, b, draw):
    assert a["log"] == b["log"], (draw, a["log"], b["log"])
    assert a["ret"] == b["ret"], (draw, a["ret"], b["ret"], a["msg"], b["msg"])
    assert a["nev"] == b["nev"], (draw, a["nev"], b["nev"])
    assert a["calls"].shape == b["calls"].shape, (draw, a["calls"].shape, b["calls"].shape)
    bad = np.flatnonzero(np.any((a["calls"] != b["calls"]) & ~(np.isnan(a["calls"]) & np.isnan(b["calls"])), axis=1))
    assert bad.size == 0, (draw, "first differing call", int(bad[0]), a["calls"][bad[0]], b["calls"][bad[0]])
    if a["ret"] > 0 or a["ret"] == -5:
        assert (a["minf"] == b["minf"] or (np.isnan(a["minf"]) and np.isnan(b["minf"]))) and np.array_equal(a["x"], b["x"], equal_nan=True), (draw, a["minf"], b["minf"])
    if a["ret"] > 0:
        assert a["msg"] == b["msg"], (draw, a["msg"], b["msg"])


@pytest.mark.parametrize("first", range(0, 200, 40))
def test_constrained_mma_is_the_references_run_call_by_call(first):
    R, A = bind(O.ref()), bind(C.CDLL(EMU))
    ran = 0
    for draw in range(first, first + 40):
        r = play(R, draw)
        same(r, play(A, draw), draw)
        ran += r["ret"] > 0 and len(r["calls"]) > 6
    assert ran >= 20, "most drawn problems should run for a while"
