"""-m gpu: the stop conditions a callback or the clock can raise, against the REAL reference: nlopt_force_stop() from inside the
objective (nlopt.h:275; checked after every evaluation, crs.c:134, isres.c:195, esch.c:178) and maxtime."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")]
FUNC = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)


def run_lib(L, alg, n, pop, seed, stop_after, maxeval=100000):
    """the same client code against either library (handles passed as void*)"""
    L.nlopt_create.restype = C.c_void_p
    L.nlopt_create.argtypes = [C.c_int, C.c_uint]
    for nm in ("nlopt_set_lower_bounds1", "nlopt_set_upper_bounds1"):
        getattr(L, nm).argtypes = [C.c_void_p, C.c_double]
    L.nlopt_set_min_objective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.nlopt_set_population.argtypes = [C.c_void_p, C.c_uint]
    L.nlopt_set_maxeval.argtypes = [C.c_void_p, C.c_int]
    L.nlopt_optimize.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.nlopt_force_stop.argtypes = [C.c_void_p]
    L.nlopt_get_numevals.argtypes = [C.c_void_p]
    L.nlopt_destroy.argtypes = [C.c_void_p]
    L.nlopt_srand.argtypes = [C.c_ulong]
    opt = L.nlopt_create(alg, n)
    calls = [0]

    def f(nn, x, g, d):
        calls[0] += 1
        v = sum((x[i] - 0.3 * i) ** 2 for i in range(nn))
        if calls[0] == stop_after:
            L.nlopt_force_stop(opt)
        return v
    cb = FUNC(f)
    L.nlopt_set_lower_bounds1(opt, -4.0)
    L.nlopt_set_upper_bounds1(opt, 5.0)
    L.nlopt_set_min_objective(opt, C.cast(cb, C.c_void_p), None)
    if pop:
        L.nlopt_set_population(opt, pop)
    L.nlopt_set_maxeval(opt, maxeval)
    x = np.linspace(-1.0, 1.0, n)
    minf = C.c_double()
    L.nlopt_srand(seed)
    ret = L.nlopt_optimize(opt, x.ctypes.data_as(C.POINTER(C.c_double)), C.byref(minf))
    nev = L.nlopt_get_numevals(opt)
    L.nlopt_destroy(opt)
    return ret, nev, calls[0], minf.value, x


@pytest.mark.parametrize("alg,pop,stop_after", [(19, 30, 10), (19, 30, 200), (35, 25, 7), (35, 25, 90), (42, 12, 5), (42, 12, 70)])
def test_force_stop_from_the_callback(alg, pop, stop_after):
    """during the initial population and inside the main loop: FORCED_STOP (-5) after as many evaluations as the reference makes
    (crs_init does not look at the flag, crs.c:205-225: a stop raised there takes effect at the first trial; ISRES and ESCH
    test it after every candidate); the best point so far is what the reference returns too (CRS2_LM exactly; ISRES / ESCH
    to rounding)"""
    r = run_lib(O.ref(), alg, 4, pop, 3, stop_after)
    a = run_lib(nlopt_amd.lib(), alg, 4, pop, 3, stop_after)
    assert a[0] == r[0] == nlopt_amd.FORCED_STOP
    assert a[1] == r[1] and a[2] == r[2]
    assert r[1] == (max(stop_after, pop + 1) if alg == 19 else stop_after)
    if alg == 19:
        assert a[3] == r[3] and np.array_equal(a[4], r[4])
    else:
        assert abs(a[3] - r[3]) <= 1e-9 * max(abs(r[3]), 1e-300) or (np.isinf(a[3]) and np.isinf(r[3]))
        assert np.allclose(a[4], r[4], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("alg", [nlopt_amd.GN_CRS2_LM, nlopt_amd.GN_ISRES, nlopt_amd.GN_ESCH])
def test_maxtime_stops_the_run(alg):
    n = 64
    xs, lo, hi = O.golden_x0("rastrigin", n)
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective("rastrigin"))
    o.set_population(2000)
    o.set_maxtime(0.3)
    nlopt_amd.srand(1)
    x, minf, ret = o.optimize_raw(xs)
    assert ret == nlopt_amd.MAXTIME_REACHED and o.get_numevals() > 2000 and np.isfinite(minf)


@pytest.mark.parametrize("alg", [nlopt_amd.LD_MMA, nlopt_amd.LD_LBFGS])
def test_maxtime_is_observed_inside_a_device_resident_local_search(alg):
    """a local search with a device objective is ONE kernel launch; the kernel polls an abort flag the waiting host raises when
    the clock runs out (plis.c:263,273,371; mma.c:258-260,394-396).  Rosenbrock n = 4096 with tolerances that cannot be met:
    without the flag LD_MMA would run for minutes (round 1 refused a maxtime-only LD_MMA run for that reason)."""
    import time
    n = 4096
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(-30.0)
    o.set_upper_bounds(30.0)
    o.set_min_objective(nlopt_amd.objective("rosenbrock"))
    o.set_maxtime(0.4)
    if alg == nlopt_amd.LD_LBFGS:
        o.set_ftol_abs(1e-300)
        o.set_param("tolg", 1e-300)
    t0 = time.time()
    x, minf, ret = o.optimize_raw(np.full(n, -1.2))
    dt = time.time() - t0
    assert ret in (nlopt_amd.MAXTIME_REACHED, nlopt_amd.SUCCESS, nlopt_amd.FTOL_REACHED, nlopt_amd.XTOL_REACHED), (ret, o.get_errmsg())
    assert dt < 5.0, dt
    if alg == nlopt_amd.LD_MMA:
        assert ret == nlopt_amd.MAXTIME_REACHED and 0.35 <= dt


def test_force_stop_from_another_thread_ends_a_device_resident_search():
    import threading
    import time
    n = 4096
    o = nlopt_amd.Opt(nlopt_amd.LD_MMA, n)
    o.set_lower_bounds(-30.0)
    o.set_upper_bounds(30.0)
    o.set_min_objective(nlopt_amd.objective("rosenbrock"))
    o.set_maxeval(2000000000)
    th = threading.Timer(0.3, o.force_stop)
    th.start()
    t0 = time.time()
    x, minf, ret = o.optimize_raw(np.full(n, -1.2))
    th.join()
    assert ret == nlopt_amd.FORCED_STOP and time.time() - t0 < 5.0
