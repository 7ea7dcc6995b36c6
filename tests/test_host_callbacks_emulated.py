"""Host-callback path of the local optimisers and of MLSL (SURVEY.md §8b "Generic host callbacks take the host-eval fallback";
VERDICT r1 item 1): an ordinary nlopt_func must be served with the reference's contract — called on the caller's thread, one x at a
time, in the reference's order, with a gradient exactly when the reference asks for one (mlsl.c:335,360,404; plis.c:260,390;
mma.c:219,297,337).  Drawn configurations are issued to the REAL reference and to the product over the emulated device (the drivers
and the external-evaluation protocol of include/nlopt_amd.h are the product's; the kernels' arithmetic is emulated in the reference's
order, so everything must agree bit for bit): every callback invocation (x, gradient requested?), return code, minimum, argmin,
evaluation count, error message.  tests/test_gpu_host_callbacks.py runs the same client against the HIP kernels."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libnlopt_amd_emu.so")
pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")
FUNC = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
vp, dbl, dpp = C.c_void_p, C.c_double, C.POINTER(C.c_double)

LD_LBFGS, LD_MMA, G_MLSL, G_MLSL_LDS, GD_MLSL, GD_MLSL_LDS = 11, 24, 38, 39, 21, 23


def bind(L):
    L.nlopt_create.restype = vp
    L.nlopt_create.argtypes = [C.c_int, C.c_uint]
    L.nlopt_destroy.argtypes = [vp]
    for nm in ("nlopt_set_lower_bounds", "nlopt_set_upper_bounds", "nlopt_set_xtol_abs", "nlopt_set_x_weights"):
        getattr(L, nm).argtypes = [vp, dpp]
    for nm in ("nlopt_set_stopval", "nlopt_set_ftol_rel", "nlopt_set_ftol_abs", "nlopt_set_xtol_rel", "nlopt_set_xtol_abs1",
               "nlopt_set_initial_step1"):
        getattr(L, nm).argtypes = [vp, dbl]
    L.nlopt_set_min_objective.argtypes = [vp, vp, vp]
    L.nlopt_set_max_objective.argtypes = [vp, vp, vp]
    L.nlopt_set_population.argtypes = [vp, C.c_uint]
    L.nlopt_set_vector_storage.argtypes = [vp, C.c_uint]
    L.nlopt_set_maxeval.argtypes = [vp, C.c_int]
    L.nlopt_set_local_optimizer.argtypes = [vp, vp]
    L.nlopt_set_param.argtypes = [vp, C.c_char_p, dbl]
    L.nlopt_optimize.argtypes = [vp, dpp, dpp]
    L.nlopt_get_numevals.argtypes = [vp]
    L.nlopt_force_stop.argtypes = [vp]
    L.nlopt_get_errmsg.argtypes = [vp]
    L.nlopt_get_errmsg.restype = C.c_char_p
    L.nlopt_srand.argtypes = [C.c_ulong]
    return L


@pytest.fixture(scope="module")
def libs():
    return bind(O.ref()), bind(C.CDLL(EMU))


def objective_factory(rng, n):
    """a smooth multimodal objective with an analytic gradient; records every call"""
    centre = rng.uniform(-1, 2, n)
    wgt = 1 + 0.3 * np.arange(n)
    amp = float(rng.uniform(0.0, 0.8))

    def make(calls, on_call=None):
        def f(nn, xp, gp, d):
            x = np.array([xp[i] for i in range(nn)])
            calls.append((x.tobytes(), bool(gp)))
            if on_call is not None:
                on_call(len(calls))
            if gp:
                g = 2 * wgt * (x - centre) - 3 * amp * np.sin(3 * x)
                for i in range(nn):
                    gp[i] = float(g[i])
            return float(np.sum(wgt * (x - centre) ** 2) + amp * np.sum(np.cos(3 * x)))
        return FUNC(f)
    return make


def play(L, draw, algs):
    rng = np.random.default_rng(4200 + draw)
    dp = lambda a: a.ctypes.data_as(dpp)
    alg = int(rng.choice(algs))
    n = int(rng.integers(1, 8))
    opt = L.nlopt_create(alg, n)
    lb, ub = np.full(n, -3.0) - rng.random(n), np.full(n, 4.0) + rng.random(n)
    if rng.random() < 0.2:
        lb[int(rng.integers(n))] = 0.7                   # a bound that becomes active near the minimum
    L.nlopt_set_lower_bounds(opt, dp(lb))
    L.nlopt_set_upper_bounds(opt, dp(ub))
    make = objective_factory(rng, n)
    calls = []
    stop_at = int(rng.integers(2, 60)) if rng.random() < 0.2 else 0
    fcb = make(calls, (lambda k: L.nlopt_force_stop(opt) if k == stop_at else None) if stop_at else None)
    maximise = rng.random() < 0.15
    if maximise:
        calls2 = calls
        inner = fcb

        def neg(nn, xp, gp, d):
            v = inner(nn, xp, gp, d)
            if gp:
                for i in range(nn):
                    gp[i] = -gp[i]
            return -v
        fcb2 = FUNC(neg)
        L.nlopt_set_max_objective(opt, C.cast(fcb2, vp), None)
        keep = [fcb, fcb2, calls2]
    else:
        L.nlopt_set_min_objective(opt, C.cast(fcb, vp), None)
        keep = [fcb]
    is_mlsl = alg not in (LD_LBFGS, LD_MMA)
    target = opt                                         # the object whose tolerances drive the local searches
    loc = None
    if is_mlsl:
        L.nlopt_set_population(opt, int(rng.choice([0, 3, 7, 12])))
        L.nlopt_set_maxeval(opt, int(rng.integers(60, 700)))
        if alg in (G_MLSL, G_MLSL_LDS) or rng.random() < 0.5:
            loc = L.nlopt_create(int(rng.choice([LD_LBFGS, LD_MMA])), n)
            target = loc
    else:
        L.nlopt_set_maxeval(opt, int(rng.integers(5, 300)))
    if rng.random() < 0.7:
        L.nlopt_set_ftol_rel(target, float(10.0 ** rng.uniform(-12, -3)))
    if rng.random() < 0.2:
        L.nlopt_set_ftol_abs(target, float(10.0 ** rng.uniform(-12, -4)))
    if rng.random() < 0.4:
        L.nlopt_set_xtol_rel(target, float(10.0 ** rng.uniform(-9, -2)))
    if rng.random() < 0.25:
        L.nlopt_set_xtol_abs1(target, float(rng.choice([0.0, 1e-9, 1e-5, 1e-2])))
    if rng.random() < 0.25:
        L.nlopt_set_x_weights(target, dp(rng.uniform(0.01, 50.0, n)))
    if rng.random() < 0.2:
        L.nlopt_set_stopval(opt, float(rng.uniform(-2, 6)) * (-1 if maximise else 1))
    if rng.random() < 0.3:
        L.nlopt_set_vector_storage(target, int(rng.choice([1, 2, 5, 40])))
    if rng.random() < 0.3:
        for name, val in (("inner_gradients", 0), ("always_improve", int(rng.integers(0, 2))), ("inner_maxeval", int(rng.integers(0, 4))),
                          ("rho_init", float(rng.choice([0.3, 1.0, 7.0])))):
            if rng.random() < 0.6:
                L.nlopt_set_param(target, name.encode(), float(val))
    if rng.random() < 0.2:
        L.nlopt_set_initial_step1(opt, float(rng.uniform(0.05, 1.5)))
    if loc is not None:
        if rng.random() < 0.3:
            L.nlopt_set_maxeval(loc, int(rng.integers(3, 40)))
        L.nlopt_set_local_optimizer(opt, loc)
        L.nlopt_destroy(loc)
    x = rng.uniform(lb + 0.8, ub - 0.5)
    minf = C.c_double(123.0)
    L.nlopt_srand(int(rng.integers(1, 2 ** 31)))
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    msg = L.nlopt_get_errmsg(opt)
    out = dict(alg=alg, n=n, ret=ret, minf=minf.value, x=x, nev=L.nlopt_get_numevals(opt), msg=msg.decode() if msg else None, calls=calls)
    L.nlopt_destroy(opt)
    del keep
    return out


def same(a, b, draw):
    ctx = "draw %d alg %d n %d" % (draw, a["alg"], a["n"])
    assert len(a["calls"]) == len(b["calls"]), (ctx, len(a["calls"]), len(b["calls"]), a["ret"], b["ret"], b["msg"])
    for k, (u, v) in enumerate(zip(a["calls"], b["calls"])):
        assert u == v, (ctx, "call", k, np.frombuffer(u[0]), np.frombuffer(v[0]), u[1], v[1])
    assert a["ret"] == b["ret"], (ctx, a["ret"], b["ret"], a["msg"], b["msg"])
    assert a["nev"] == b["nev"], ctx
    assert a["minf"] == b["minf"] or (np.isnan(a["minf"]) and np.isnan(b["minf"])), ctx
    assert np.array_equal(a["x"], b["x"]), ctx


@pytest.mark.parametrize("draw", range(120))
def test_local_optimisers_with_host_callbacks_equal_the_reference(libs, draw):
    R, E = libs
    same(play(R, draw, [LD_LBFGS, LD_MMA]), play(E, draw, [LD_LBFGS, LD_MMA]), draw)


@pytest.mark.parametrize("draw", range(100))
def test_mlsl_with_host_callbacks_equals_the_reference(libs, draw):
    R, E = libs
    algs = [G_MLSL, G_MLSL_LDS, GD_MLSL, GD_MLSL_LDS]
    same(play(R, 1000 + draw, algs), play(E, 1000 + draw, algs), 1000 + draw)


def test_summation_order_default_follows_the_kind_of_objective():
    """nla_exact_mode_for (userobj.c): a client's own nlopt_func gets the reference's sequential sums in the device local
    optimisers unless "amd_exact_dot" says otherwise; registered device objectives get the tree reductions unless it says so —
    read back from the launch parameters through the emulated device's hook"""
    import ctypes as C
    import numpy as np
    from test_api_differential import EMU, FUNC, bind, vp, dpp
    L = bind(C.CDLL(EMU))
    L.nlopt_set_param.argtypes = [vp, C.c_char_p, C.c_double]
    L.nlopt_amd_objective.restype = vp
    L.nlopt_amd_objective.argtypes = [C.c_int]
    L.nlopt_set_local_optimizer.argtypes = [vp, vp]
    hook = C.c_int.in_dll(L, "nla_emu_last_exact")
    own = FUNC(lambda n, x, g, d: (g and [g.__setitem__(i, 2 * x[i]) for i in range(n)]) and 0.0 or float(sum(x[i] * x[i] for i in range(n))))

    def run(alg, f, param=None, local=None, local_param=None):
        o = L.nlopt_create(alg, 3)
        L.nlopt_set_min_objective(o, f, None)
        lb, ub, x = np.full(3, -1.0), np.full(3, 2.0), np.full(3, 0.5)
        L.nlopt_set_lower_bounds(o, lb.ctypes.data_as(dpp))
        L.nlopt_set_upper_bounds(o, ub.ctypes.data_as(dpp))
        L.nlopt_set_maxeval(o, 40)
        if param is not None:
            L.nlopt_set_param(o, b"amd_exact_dot", float(param))
        lo = None
        if local is not None:
            lo = L.nlopt_create(local, 3)
            if local_param is not None:
                L.nlopt_set_param(lo, b"amd_exact_dot", float(local_param))
            L.nlopt_set_local_optimizer(o, lo)
        hook.value = -1
        mf = C.c_double()
        r = L.nlopt_optimize(o, x.ctypes.data_as(dpp), C.byref(mf))
        L.nlopt_destroy(o)
        if lo:
            L.nlopt_destroy(lo)
        assert r > 0
        return hook.value
    dev, host = L.nlopt_amd_objective(5), C.cast(own, vp)
    LD_LBFGS, LD_MMA, G_MLSL_LDS, GD_MLSL = 11, 24, 39, 21
    for alg in (LD_LBFGS, LD_MMA):
        assert run(alg, host) == 1 and run(alg, dev) == 0
        assert run(alg, host, param=0) == 0 and run(alg, dev, param=1) == 1
    assert run(GD_MLSL, host) == 1 and run(GD_MLSL, dev) == 0
    assert run(G_MLSL_LDS, host, local=LD_LBFGS) == 1 and run(G_MLSL_LDS, dev, local=LD_LBFGS) == 0
    assert run(G_MLSL_LDS, dev, local=LD_LBFGS, local_param=1) == 1 and run(G_MLSL_LDS, host, local=LD_MMA, local_param=0) == 0
    assert run(G_MLSL_LDS, dev, param=1, local=LD_LBFGS, local_param=0) == 1
