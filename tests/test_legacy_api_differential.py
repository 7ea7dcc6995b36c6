"""The pre-2.0 one-call interface — nlopt_minimize / nlopt_minimize_constrained / nlopt_minimize_econstrained
(src/api/nlopt.h:317-337, src/api/deprecated.c:65-189) — in front of the global-search path, against the REAL reference:
the same positional call (per-constraint data at byte offsets i * datum_size, the ftol pair reused as htol by
nlopt_minimize_constrained, xtol_abs NULL or given, every stopping criterion, the deprecated globals for the population and
for MLSL's local optimiser) is made on both libraries through Python callbacks; the point of EVERY objective / constraint
call, the return code, x and minf must be identical.  The product runs over the emulated device layer here (the three entry
points are host code above nlopt_optimize; the device paths behind it have their own GPU tests)."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O
from test_api_differential import EMU, vp, dpp

pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")
OLDFUNC = C.CFUNCTYPE(C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
GN_CRS2_LM, GN_MLSL, GN_MLSL_LDS, LN_COBYLA, GN_ISRES, GN_ESCH, LN_NELDERMEAD, LD_MMA = 19, 20, 22, 25, 35, 42, 28, 24
dbl, cint, psz = C.c_double, C.c_int, C.c_ssize_t


def legacy_bind(L):
    L.nlopt_srand.argtypes = [C.c_ulong]
    L.nlopt_minimize.argtypes = [cint, cint, vp, vp, dpp, dpp, dpp, dpp, dbl, dbl, dbl, dbl, dpp, cint, dbl]
    L.nlopt_minimize_constrained.argtypes = [cint, cint, vp, vp, cint, vp, vp, psz, dpp, dpp, dpp, dpp,
                                             dbl, dbl, dbl, dbl, dpp, cint, dbl]
    L.nlopt_minimize_econstrained.argtypes = [cint, cint, vp, vp, cint, vp, vp, psz, cint, vp, vp, psz, dpp, dpp, dpp, dpp,
                                              dbl, dbl, dbl, dbl, dpp, dbl, dbl, cint, dbl]
    L.nlopt_set_stochastic_population.argtypes = [cint]
    L.nlopt_set_local_search_algorithm.argtypes = [cint, cint, cint]
    L.nlopt_get_local_search_algorithm.argtypes = [C.POINTER(cint)] * 3
    return L


def dp(a):
    return a.ctypes.data_as(dpp) if a is not None else None


def play(L, draw):
    rng = np.random.default_rng(47000 + draw)
    alg = [GN_CRS2_LM, GN_ISRES, GN_ESCH, GN_MLSL, GN_MLSL_LDS, LN_COBYLA][draw % 6]
    n = int(rng.integers(1, 6))
    lb, ub = -2.0 - rng.random(n), 3.0 + rng.random(n)
    x = rng.uniform(lb, ub)
    centre = rng.uniform(-1, 2, n)
    calls = []

    def f(nn, xx, g, d):
        xs = np.array([xx[i] for i in range(nn)])
        calls.append(("f", xs.copy()))
        return float(np.sum((xs - centre) ** 2) + 0.5 * np.sum(np.cos(4 * xs)))
    fcb = OLDFUNC(f)
    takes_constraints = alg in (GN_ISRES, LN_COBYLA)
    m = int(rng.integers(0, 3)) if takes_constraints else 0
    p = int(rng.integers(0, 2)) if takes_constraints else 0
    # per-constraint data: records of `stride` doubles, only the first of each record is the constraint's own datum
    stride = int(rng.integers(1, 4))
    cdata = np.ascontiguousarray(rng.uniform(-0.5, 0.5, (max(m, 1), stride)))
    hdata = np.ascontiguousarray(rng.uniform(-0.2, 0.2, (max(p, 1), stride)))

    def fc(nn, xx, g, d):
        c = C.cast(d, C.POINTER(dbl))[0]
        calls.append(("c", c))
        return float(xx[0] + (xx[1 % nn] if nn > 1 else 0.0) - 2.5 - c)

    def h(nn, xx, g, d):
        c = C.cast(d, C.POINTER(dbl))[0]
        calls.append(("h", c))
        return float(xx[(nn - 1)] - 0.5 - c)
    ccb, hcb = OLDFUNC(fc), OLDFUNC(h)
    xtol_abs = rng.uniform(1e-9, 1e-5, n) if rng.random() < 0.4 else None
    stopval = float(rng.choice([-np.inf, -np.inf, 0.4]))
    ftol_rel, ftol_abs = float(rng.choice([0.0, 1e-6])), float(rng.choice([0.0, 1e-9]))
    xtol_rel = float(rng.choice([0.0, 1e-7]))
    maxeval = int(rng.integers(60, 260))
    htol_abs = float(rng.choice([0.0, 1e-6, 1e-3]))
    pop = int(rng.choice([0, 0, 7, 23]))
    L.nlopt_set_stochastic_population(pop)
    if alg in (GN_MLSL, GN_MLSL_LDS):
        L.nlopt_set_local_search_algorithm(LD_MMA, LN_COBYLA, int(rng.choice([-1, 25])))
    L.nlopt_srand(900 + draw)
    minf = dbl(np.nan)
    which = draw % 3 if takes_constraints else 0
    try:
        if which == 0 and not (m or p):
            ret = L.nlopt_minimize(alg, n, C.cast(fcb, vp), None, dp(lb), dp(ub), dp(x), C.byref(minf),
                                   stopval, ftol_rel, ftol_abs, xtol_rel, dp(xtol_abs), maxeval, 0.0)
        elif which == 1 or not p:
            p = 0
            ret = L.nlopt_minimize_constrained(alg, n, C.cast(fcb, vp), None, m, C.cast(ccb, vp), cdata.ctypes.data, 8 * stride,
                                               dp(lb), dp(ub), dp(x), C.byref(minf),
                                               stopval, ftol_rel, ftol_abs, xtol_rel, dp(xtol_abs), maxeval, 0.0)
        else:
            ret = L.nlopt_minimize_econstrained(alg, n, C.cast(fcb, vp), None, m, C.cast(ccb, vp), cdata.ctypes.data, 8 * stride,
                                                p, C.cast(hcb, vp), hdata.ctypes.data, 8 * stride,
                                                dp(lb), dp(ub), dp(x), C.byref(minf),
                                                stopval, ftol_rel, ftol_abs, xtol_rel, dp(xtol_abs), 0.123, htol_abs, maxeval, 0.0)
    finally:
        L.nlopt_set_stochastic_population(0)
        L.nlopt_set_local_search_algorithm(LD_MMA, LN_COBYLA, -1)
    return ret, x.copy(), minf.value, calls


def same(a, b):
    assert a[0] == b[0]
    assert len(a[3]) == len(b[3])
    for (ka, va), (kb, vb) in zip(a[3], b[3]):
        assert ka == kb and np.array_equal(va, vb)
    assert np.array_equal(a[1], b[1])
    assert a[2] == b[2] or (np.isnan(a[2]) and np.isnan(b[2]))


@pytest.mark.parametrize("draw", range(36))
def test_the_one_call_interface_agrees_with_the_reference(draw):
    r = play(legacy_bind(O.ref()), draw)
    a = play(legacy_bind(C.CDLL(EMU)), draw)
    assert r[0] > 0 and len(r[3]) >= 1, "the drawn call should run"
    same(r, a)


def test_invalid_arguments_of_the_one_call_interface():
    for L in (legacy_bind(O.ref()), legacy_bind(C.CDLL(EMU))):
        fcb = OLDFUNC(lambda nn, xx, g, d: 0.0)
        lb, ub, x = np.zeros(2), np.ones(2), np.full(2, 0.5)
        minf = dbl(0)
        args = (C.cast(fcb, vp), None, dp(lb), dp(ub), dp(x), C.byref(minf), -np.inf, 0.0, 0.0, 0.0, None, 10, 0.0)
        assert L.nlopt_minimize(GN_CRS2_LM, -1, *args) == -2                      # n < 0
        assert L.nlopt_minimize(9999, 2, *args) == -2                             # no such algorithm
        assert L.nlopt_minimize_constrained(GN_CRS2_LM, 2, C.cast(fcb, vp), None, -1, None, None, 0, dp(lb), dp(ub), dp(x),
                                            C.byref(minf), -np.inf, 0.0, 0.0, 0.0, None, 10, 0.0) == -2   # m < 0
        # an algorithm that takes no inequality constraints: the setter's refusal is what comes back (deprecated.c:90-95)
        assert L.nlopt_minimize_constrained(GN_CRS2_LM, 2, C.cast(fcb, vp), None, 1, C.cast(fcb, vp), None, 0, dp(lb), dp(ub),
                                            dp(x), C.byref(minf), -np.inf, 0.0, 0.0, 0.0, None, 10, 0.0) == -2
        # bounds the wrong way round are found by nlopt_optimize, not by the setters
        assert L.nlopt_minimize(GN_CRS2_LM, 2, C.cast(fcb, vp), None, dp(ub + 1), dp(lb), dp(x), C.byref(minf),
                                -np.inf, 0.0, 0.0, 0.0, None, 10, 0.0) == -2
        # a NULL objective
        assert L.nlopt_minimize(GN_CRS2_LM, 2, None, None, dp(lb), dp(ub), dp(x), C.byref(minf),
                                -np.inf, 0.0, 0.0, 0.0, None, 10, 0.0) == -2
