"""The augmented-Lagrangian wrapper (auglag_host.c: AUGLAG / AUGLAG_EQ with an explicit subsidiary optimiser, LN_ / LD_AUGLAG(_EQ)
with the default one) against the REAL reference, call by call.  AUGLAG is a caller of the path: drawn problems with scalar
and vector, inequality and equality constraints are solved with every optimiser this library serves underneath — the global
ones (CRS2_LM, ISRES, ESCH, MLSL) for constrained global searches, LD_LBFGS, LD_MMA, LN_COBYLA locally — through Python
callbacks; the point of EVERY objective and constraint call, whether a gradient was asked for, results, counts and messages
must be identical.  The product runs over the emulated device layer here."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from test_api_differential import EMU, FUNC, MFUNC, vp
from test_cobyla_differential import dp
from test_mma_constrained_differential import bind, same

pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")
AUGLAG, AUGLAG_EQ, LN_AUGLAG, LD_AUGLAG, LN_AUGLAG_EQ, LD_AUGLAG_EQ = 36, 37, 30, 31, 32, 33
LD_LBFGS, LD_MMA, LN_COBYLA, GN_CRS2_LM, GN_ISRES, GN_ESCH, GN_MLSL_LDS, GD_MLSL = 11, 24, 25, 19, 35, 42, 22, 21
SUBS = [LD_LBFGS, LD_MMA, LN_COBYLA, GN_CRS2_LM, GN_ISRES, GN_ESCH, GN_MLSL_LDS, GD_MLSL]


def play(L, draw):
    rng = np.random.default_rng(91000 + draw)
    n = int(rng.integers(1, 5))
    calls = []
    flavour = int(rng.integers(3))                              # 0: explicit subsidiary optimiser, 1: LD default, 2: LN default
    eq_only = rng.random() < 0.4
    if flavour == 0:
        alg, sub = (AUGLAG_EQ if eq_only else AUGLAG), int(SUBS[draw % len(SUBS)])
    elif flavour == 1:
        alg, sub = (LD_AUGLAG_EQ if eq_only else LD_AUGLAG), None
    else:
        alg, sub = (LN_AUGLAG_EQ if eq_only else LN_AUGLAG), None
    if eq_only and sub is not None and sub not in (LD_MMA, LN_COBYLA, GN_ISRES):
        eq_only, alg = False, AUGLAG                            # the others take no inequality constraints of their own
    if eq_only and sub is None and alg == LD_AUGLAG_EQ:
        pass                                                    # default LD_MMA takes the inequalities itself (mma_host.c)
    opt = L.nlopt_create(alg, n)
    lb, ub = np.full(n, -2.0) - rng.random(n), np.full(n, 3.0) + rng.random(n)
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub))]
    centre = rng.uniform(-1, 2, n)

    def f(nn, x, g, d):
        xs = np.array([x[i] for i in range(nn)])
        calls.append(np.concatenate(([0.0, 1.0 if g else 0.0], xs)))
        if g:
            for i in range(nn):
                g[i] = 2 * (xs[i] - centre[i]) * (1 + 0.5 * i) - 0.3 * np.sin(3 * xs[i])
        return float(np.sum((xs - centre) ** 2 * (1 + 0.5 * np.arange(nn))) + 0.1 * np.sum(np.cos(3 * xs)))
    fcb = FUNC(f)
    keep = [fcb]
    log.append(L.nlopt_set_min_objective(opt, C.cast(fcb, vp), None))
    nineq, neq = int(rng.integers(0, 3)), int(rng.integers(0, 2))
    if nineq + neq == 0:
        nineq = 1
    for q in range(nineq):
        cq = float(rng.uniform(-1.0, 0.6))

        def c(nn, x, g, d, cq=cq, q=q):
            xs = np.array([x[i] for i in range(nn)])
            calls.append(np.concatenate(([1.0 + q, 1.0 if g else 0.0], xs)))
            a, b = q % nn, (q + 1) % nn
            if g:
                for i in range(nn):
                    g[i] = 0.0
                g[a] += 2 * xs[a]
                g[b] += 1.0
            return float(xs[a] ** 2 + xs[b] - 2 + cq)
        cb = FUNC(c)
        keep.append(cb)
        log.append(L.nlopt_add_inequality_constraint(opt, C.cast(cb, vp), None, float(rng.choice([0.0, 1e-8, 1e-4]))))
    for q in range(neq):
        def h(nn, x, g, d, q=q):
            xs = np.array([x[i] for i in range(nn)])
            calls.append(np.concatenate(([5.0 + q, 1.0 if g else 0.0], xs)))
            if g:
                for i in range(nn):
                    g[i] = 0.0
                g[nn - 1] += 1.0
                g[0] -= 0.5
            return float(xs[nn - 1] - 0.5 * xs[0] - 0.25)
        hb = FUNC(h)
        keep.append(hb)
        log.append(L.nlopt_add_equality_constraint(opt, C.cast(hb, vp), None, float(rng.choice([1e-8, 1e-4]))))
    if rng.random() < 0.3:
        m = int(rng.integers(1, 3))

        def mf(mm, res, nn, x, g, d):
            xs = np.array([x[i] for i in range(nn)])
            calls.append(np.concatenate(([9.0, 1.0 if g else 0.0], xs)))
            for i in range(mm):
                res[i] = float(xs[i % nn] + 0.5 * xs[(i + 1) % nn] ** 2 - 2.5 - 0.3 * i)
                if g:
                    for j in range(nn):
                        g[i * nn + j] = 0.0
                    g[i * nn + i % nn] += 1.0
                    g[i * nn + (i + 1) % nn] += xs[(i + 1) % nn]
        mcb = MFUNC(mf)
        keep.append(mcb)
        tol = np.full(m, 1e-6)
        add = L.nlopt_add_equality_mconstraint if rng.random() < 0.3 else L.nlopt_add_inequality_mconstraint
        log.append(add(opt, m, C.cast(mcb, vp), None, dp(tol)))
    lo = None
    if sub is not None:
        lo = L.nlopt_create(sub, n)
        if rng.random() < 0.7:
            L.nlopt_set_xtol_rel(lo, float(rng.choice([1e-3, 1e-6])))
        if rng.random() < 0.4:
            L.nlopt_set_ftol_rel(lo, float(rng.choice([1e-4, 1e-8])))
        if sub in (GN_CRS2_LM, GN_ISRES, GN_ESCH, GN_MLSL_LDS, GD_MLSL):
            L.nlopt_set_maxeval(lo, int(rng.choice([40, 150])))
            L.nlopt_set_population(lo, int(rng.choice([0, 12])))
        elif rng.random() < 0.3:
            L.nlopt_set_maxeval(lo, 30)
        log.append(L.nlopt_set_local_optimizer(opt, lo))
    if rng.random() < 0.6:
        log.append(L.nlopt_set_xtol_rel(opt, float(rng.choice([1e-3, 1e-6]))))
    if rng.random() < 0.4:
        log.append(L.nlopt_set_ftol_rel(opt, float(rng.choice([1e-4, 1e-8]))))
    if rng.random() < 0.15:
        log.append(L.nlopt_set_ftol_abs(opt, 1e-7))
    if rng.random() < 0.15:
        log.append(L.nlopt_set_stopval(opt, float(rng.uniform(-1, 4))))
    if rng.random() < 0.2:
        log.append(L.nlopt_set_initial_step1(opt, float(rng.uniform(0.05, 0.8))))
    log.append(L.nlopt_set_maxeval(opt, int(rng.choice([3, 40, 300, 1200]))))
    L.nlopt_srand(4321 + draw)
    x = np.clip(rng.uniform(-1.5, 2.5, n), lb, ub)
    minf = C.c_double(0)
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    out = dict(log=log, ret=ret, minf=minf.value, x=x.copy(), nev=L.nlopt_get_numevals(opt), msg=L.nlopt_get_errmsg(opt),
               calls=np.array(calls) if calls else np.zeros((0, n + 2)), alg=alg, sub=sub)
    L.nlopt_destroy(opt)
    if lo:
        L.nlopt_destroy(lo)
    return out


@pytest.mark.parametrize("first", range(0, 240, 40))
def test_auglag_is_the_references_run_call_by_call(first):
    R, A = bind(O.ref()), bind(C.CDLL(EMU))
    ran = 0
    for draw in range(first, first + 40):
        r = play(R, draw)
        same(r, play(A, draw), draw)
        ran += r["ret"] > 0 and len(r["calls"]) > 10
    assert ran >= 25, "most drawn problems should run for a while"


def test_auglag_without_a_subsidiary_optimiser_is_refused_like_the_reference():
    for L in (bind(O.ref()), bind(C.CDLL(EMU))):
        opt = L.nlopt_create(AUGLAG, 2)
        cb = FUNC(lambda nn, x, g, d: 0.0)
        L.nlopt_set_min_objective(opt, C.cast(cb, vp), None)
        x, minf = np.zeros(2), C.c_double()
        assert L.nlopt_optimize(opt, dp(x), C.byref(minf)) == -2
        assert L.nlopt_get_errmsg(opt) == b"local optimizer must be specified for AUGLAG"
        L.nlopt_destroy(opt)
