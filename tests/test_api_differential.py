"""Differential test of the API shell: drawn sequences of public NLopt calls (algorithm, bounds incl. fixed and invalid ones, any
subset of the stopping criteria, x weights, population, minimise / maximise, scalar and vector constraints for ISRES, start points
inside and outside the box) are issued, call by call, to the REAL reference and to the product over the emulated device; every
return code, the result, the minimum, the argmin, the evaluation count and the error message must agree.  Objectives and
constraints are Python callbacks, i.e. the host-callback path of CRS2_LM / ISRES / ESCH."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

EMU = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libnlopt_amd_emu.so")
pytestmark = pytest.mark.skipif(not (O.have_ref() and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")
FUNC = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
MFUNC = C.CFUNCTYPE(None, C.c_uint, C.POINTER(C.c_double), C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)
vp, dbl, dpp = C.c_void_p, C.c_double, C.POINTER(C.c_double)


def bind(L):
    L.nlopt_create.restype = vp
    L.nlopt_create.argtypes = [C.c_int, C.c_uint]
    L.nlopt_destroy.argtypes = [vp]
    L.nlopt_copy.restype = vp
    L.nlopt_copy.argtypes = [vp]
    for nm in ("nlopt_set_lower_bounds", "nlopt_set_upper_bounds", "nlopt_set_xtol_abs", "nlopt_set_x_weights"):
        getattr(L, nm).argtypes = [vp, dpp]
    for nm in ("nlopt_set_stopval", "nlopt_set_ftol_rel", "nlopt_set_ftol_abs", "nlopt_set_xtol_rel", "nlopt_set_xtol_abs1", "nlopt_set_lower_bounds1"):
        getattr(L, nm).argtypes = [vp, dbl]
    L.nlopt_set_min_objective.argtypes = [vp, vp, vp]
    L.nlopt_set_max_objective.argtypes = [vp, vp, vp]
    L.nlopt_add_inequality_constraint.argtypes = [vp, vp, vp, dbl]
    L.nlopt_add_equality_constraint.argtypes = [vp, vp, vp, dbl]
    L.nlopt_add_inequality_mconstraint.argtypes = [vp, C.c_uint, vp, vp, dpp]
    L.nlopt_set_population.argtypes = [vp, C.c_uint]
    L.nlopt_set_maxeval.argtypes = [vp, C.c_int]
    L.nlopt_optimize.argtypes = [vp, dpp, dpp]
    L.nlopt_get_numevals.argtypes = [vp]
    L.nlopt_get_errmsg.argtypes = [vp]
    L.nlopt_get_errmsg.restype = C.c_char_p
    L.nlopt_srand.argtypes = [C.c_ulong]
    return L


def play(L, draw):
    rng = np.random.default_rng(9000 + draw)
    dp = lambda a: a.ctypes.data_as(dpp)
    alg = int(rng.choice([19, 35, 42]))
    n = int(rng.integers(1, 7))
    log = []
    opt = L.nlopt_create(alg, n)
    lb, ub = np.full(n, -3.0) - rng.random(n), np.full(n, 4.0) + rng.random(n)
    r = rng.random()
    if r < 0.15 and n > 1:
        lb[int(rng.integers(n))] = ub[0]                 # a fixed coordinate (or an empty box if the index is 0 ... then lb == ub there)
    elif r < 0.22:
        lb[0], ub[0] = ub[0], lb[0]                      # lb > ub: invalid
    log.append(L.nlopt_set_lower_bounds(opt, dp(lb)))
    log.append(L.nlopt_set_upper_bounds(opt, dp(ub)))
    centre = rng.uniform(-1, 2, n)

    def f(nn, x, g, d):
        return float(sum((x[i] - centre[i]) ** 2 * (1 + 0.3 * i) for i in range(nn))) + float(np.cos(3 * x[0]))
    fcb = FUNC(f)
    keep = [fcb]
    maximise = rng.random() < 0.25
    log.append((L.nlopt_set_max_objective if maximise else L.nlopt_set_min_objective)(opt, C.cast(fcb, vp), None))
    if alg == 35:
        for q in range(int(rng.integers(0, 3))):
            cq = float(rng.uniform(-1, 1))
            cb = FUNC(lambda nn, x, g, d, cq=cq, q=q: float(x[q % nn] - cq))
            keep.append(cb)
            add = L.nlopt_add_equality_constraint if rng.random() < 0.3 else L.nlopt_add_inequality_constraint
            log.append(add(opt, C.cast(cb, vp), None, float(rng.choice([0.0, 1e-8, 1e-3, -1.0]))))      # -1: invalid tolerance
        if rng.random() < 0.3:
            m = int(rng.integers(1, 4))

            def mf(mm, res, nn, x, g, d):
                for i in range(mm):
                    res[i] = x[i % nn] - 2.5 + 0.1 * i
            mcb = MFUNC(mf)
            keep.append(mcb)
            tol = np.full(m, 1e-6)
            log.append(L.nlopt_add_inequality_mconstraint(opt, m, C.cast(mcb, vp), None, dp(tol)))
    if alg != 35 and rng.random() < 0.15:
        cb = FUNC(lambda nn, x, g, d: float(x[0]))
        keep.append(cb)
        log.append(L.nlopt_add_inequality_constraint(opt, C.cast(cb, vp), None, 1e-8))                     # not supported by CRS / ESCH
    if rng.random() < 0.7:
        log.append(L.nlopt_set_population(opt, int(rng.choice([0, 1, n, n + 1, 3 * n + 5, 40]))))
    log.append(L.nlopt_set_maxeval(opt, int(rng.integers(30, 900))))
    if rng.random() < 0.3:
        log.append(L.nlopt_set_stopval(opt, float(rng.uniform(-5, 40)) * (-1 if maximise else 1)))
    if rng.random() < 0.3:
        log.append(L.nlopt_set_ftol_rel(opt, float(10.0 ** rng.uniform(-9, -1))))
    if rng.random() < 0.2:
        log.append(L.nlopt_set_ftol_abs(opt, float(10.0 ** rng.uniform(-9, 0))))
    if rng.random() < 0.3:
        log.append(L.nlopt_set_xtol_rel(opt, float(10.0 ** rng.uniform(-7, -1))))
    if rng.random() < 0.2:
        log.append(L.nlopt_set_xtol_abs1(opt, float(10.0 ** rng.uniform(-7, -1))))
    if rng.random() < 0.2:
        w = rng.uniform(0.1, 3.0, n)
        if rng.random() < 0.2:
            w[0] = -1.0                                   # invalid weight
        log.append(L.nlopt_set_x_weights(opt, dp(w)))
    if rng.random() < 0.2:                               # run a copy, destroy the original first
        c2 = L.nlopt_copy(opt)
        L.nlopt_destroy(opt)
        opt = c2
    x = rng.uniform(-2.5, 3.5, n)
    if rng.random() < 0.1:
        x[0] = 99.0                                      # start outside the box
    minf = C.c_double(123.0)
    L.nlopt_srand(int(rng.integers(1, 2 ** 31)))
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    msg = L.nlopt_get_errmsg(opt)
    out = dict(log=log, ret=ret, minf=minf.value, x=x, nev=L.nlopt_get_numevals(opt), msg=msg.decode() if msg else None)
    L.nlopt_destroy(opt)
    del keep
    return out


@pytest.mark.parametrize("draw", range(150))
def test_drawn_api_sequences_agree_with_the_reference(draw):
    r = play(bind(O.ref()), draw)
    a = play(bind(C.CDLL(EMU)), draw)
    assert a["log"] == r["log"], (a["log"], r["log"])
    assert a["ret"] == r["ret"], (a["ret"], r["ret"], a["msg"], r["msg"])
    assert a["nev"] == r["nev"]
    assert (a["minf"] == r["minf"]) or (np.isnan(a["minf"]) and np.isnan(r["minf"])), (a["minf"], r["minf"])
    assert np.array_equal(a["x"], r["x"])
    assert a["msg"] == r["msg"], (a["msg"], r["msg"])


# ---- the device-objective algorithms: MLSL (all six enums), LD_LBFGS, LD_MMA with registered objectives ------------------------
def play_local(L, draw, getter):
    rng = np.random.default_rng(50000 + draw)
    dp = lambda a: a.ctypes.data_as(dpp)
    L.nlopt_set_local_optimizer.argtypes = [vp, vp]
    L.nlopt_set_vector_storage.argtypes = [vp, C.c_uint]
    L.nlopt_set_param.argtypes = [vp, C.c_char_p, dbl]
    L.nlopt_set_initial_step1.argtypes = [vp, dbl]
    g = getattr(L, getter)
    g.restype = vp
    g.argtypes = [C.c_int]
    obj = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"][int(rng.integers(6))]
    n = int(rng.integers(2, 12))
    xs, lo, hi = O.golden_x0(obj, n)
    alg = int(rng.choice([11, 24, 20, 21, 22, 23, 38, 39]))
    opt = L.nlopt_create(alg, n)
    lb, ub = np.full(n, lo), np.full(n, hi)
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub)), L.nlopt_set_min_objective(opt, g(O.OBJ[obj]), None)]
    cfg = [obj, n, alg]                                   # what was drawn, for the failure message

    def tolerances(o):
        r = rng.random()
        if r < 0.45:
            v = float(10.0 ** -int(rng.integers(3, 11)))
            cfg.append(("ftol_rel", v))
            log.append(L.nlopt_set_ftol_rel(o, v))
        elif r < 0.7:
            v = float(10.0 ** -int(rng.integers(3, 9)))
            cfg.append(("xtol_rel", v))
            log.append(L.nlopt_set_xtol_rel(o, v))
        elif r < 0.85:
            v = float(10.0 ** -int(rng.integers(3, 12)))
            cfg.append(("ftol_abs", v))
            log.append(L.nlopt_set_ftol_abs(o, v))
        else:
            cfg.append(("no tolerance",))

    def mma_params(o):
        if rng.random() < 0.3:
            cfg.append("inner_gradients=0")
            log.append(L.nlopt_set_param(o, b"inner_gradients", 0.0))
        if rng.random() < 0.3:
            cfg.append("always_improve=0")
            log.append(L.nlopt_set_param(o, b"always_improve", 0.0))
        if rng.random() < 0.3:
            v = float(10.0 ** rng.uniform(-3, 1))
            cfg.append(("rho_init", v))
            log.append(L.nlopt_set_param(o, b"rho_init", v))
        if rng.random() < 0.2:
            v = float(rng.integers(1, 6))
            cfg.append(("inner_maxeval", v))
            log.append(L.nlopt_set_param(o, b"inner_maxeval", v))
        if rng.random() < 0.2:
            v = float(rng.uniform(0.05, 2.0))
            cfg.append(("initial_step", v))
            log.append(L.nlopt_set_initial_step1(o, v))
    if alg in (11, 24):
        tolerances(opt)
        log.append(L.nlopt_set_maxeval(opt, int(rng.integers(5, 400))))
        if alg == 11 and rng.random() < 0.5:
            log.append(L.nlopt_set_vector_storage(opt, int(rng.integers(1, 9))))
        if alg == 24:
            mma_params(opt)
        if rng.random() < 0.2:
            log.append(L.nlopt_set_stopval(opt, float(rng.uniform(0, 30))))
    else:
        explicit = alg in (38, 39, 20, 22) or rng.random() < 0.5           # G_MLSL needs one; GN_MLSL's default (COBYLA) is not provided
        if explicit:
            la = int(rng.choice([11, 24]))
            cfg.append(("local", la))
            loc = L.nlopt_create(la, n)
            tolerances(loc)
            if rng.random() < 0.3:
                v = int(rng.integers(3, 60))
                cfg.append(("local maxeval", v))
                log.append(L.nlopt_set_maxeval(loc, v))
            if la == 11 and rng.random() < 0.4:
                log.append(L.nlopt_set_vector_storage(loc, int(rng.integers(1, 9))))
            if la == 24:
                mma_params(loc)
            log.append(L.nlopt_set_local_optimizer(opt, loc))
            L.nlopt_destroy(loc)
        else:
            tolerances(opt)                              # copied to the default local optimiser (LD_MMA for the GD variants)
        if rng.random() < 0.15:
            v = float(rng.uniform(0.05, 2.0))
            cfg.append(("global initial_step", v))           # handed on to the local optimiser (optimize.c:778-779)
            log.append(L.nlopt_set_initial_step1(opt, v))
        if rng.random() < 0.7:
            v = int(rng.integers(1, 40))
            cfg.append(("population", v))
            log.append(L.nlopt_set_population(opt, v))
        v = int(rng.integers(100, 3000))
        cfg.append(("maxeval", v))
        log.append(L.nlopt_set_maxeval(opt, v))
        if rng.random() < 0.2:
            v = float(rng.uniform(0, 30))
            cfg.append(("stopval", v))
            log.append(L.nlopt_set_stopval(opt, v))
    x = np.array(xs)
    minf = C.c_double(123.0)
    L.nlopt_srand(int(rng.integers(1, 2 ** 31)))
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    out = dict(log=log, ret=ret, minf=minf.value, x=x, nev=L.nlopt_get_numevals(opt), alg=alg, cfg=cfg)
    L.nlopt_destroy(opt)
    return out


@pytest.mark.parametrize("draw", range(120))
def test_drawn_local_and_mlsl_setups_agree_with_the_reference(draw):
    """registered objectives: the reference calls the oracle's callback, the product its own evaluator on the (emulated) device"""
    P = O.port()
    ref = O.ref()
    ref.orc_objective = P.orc_objective                  # the reference library is handed the oracle's callbacks
    r = play_local(bind(ref), draw, "orc_objective")
    a = play_local(bind(C.CDLL(EMU)), draw, "nlopt_amd_objective")
    assert a["log"] == r["log"] and a["ret"] == r["ret"], (a["cfg"], a["ret"], r["ret"])
    assert a["nev"] == r["nev"] and a["minf"] == r["minf"] and np.array_equal(a["x"], r["x"]), (a["cfg"], a["nev"], r["nev"], a["minf"], r["minf"])


def play_population_max(L, draw, getter):
    """CRS2_LM / ISRES / ESCH MAXIMISING (or minimising) a registered objective, with or without a fixed coordinate"""
    rng = np.random.default_rng(88000 + draw)
    dp = lambda a: a.ctypes.data_as(dpp)
    g = getattr(L, getter)
    g.restype = vp
    g.argtypes = [C.c_int]
    obj = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"][int(rng.integers(6))]
    n = int(rng.integers(2, 9))
    xs, lo, hi = O.golden_x0(obj, n)
    alg = int(rng.choice([19, 35, 42]))
    opt = L.nlopt_create(alg, n)
    lb, ub = np.full(n, lo), np.full(n, hi)
    fixed = rng.random() < 0.25
    if fixed:
        lb[n - 1] = ub[n - 1] = 0.5 * (lo + hi) + 0.1           # a fixed coordinate: the elimination wrapper sits in front of f
    maximise = rng.random() < 0.75
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub)),
           (L.nlopt_set_max_objective if maximise else L.nlopt_set_min_objective)(opt, g(O.OBJ[obj]), None)]
    log.append(L.nlopt_set_population(opt, int(rng.choice([0, 25, 60]))))
    if rng.random() < 0.3:
        log.append(L.nlopt_set_stopval(opt, float(rng.uniform(0, 50))))
    if rng.random() < 0.3:
        log.append(L.nlopt_set_ftol_rel(opt, 1e-4))
    log.append(L.nlopt_set_maxeval(opt, int(rng.choice([40, 300, 1200]))))
    L.nlopt_srand(4321 + draw)
    x = np.clip(np.array(xs, dtype=float), lb, ub)
    minf = C.c_double(0)
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    out = dict(log=log, ret=ret, minf=minf.value, x=x, nev=L.nlopt_get_numevals(opt), cfg=(obj, n, alg, maximise, fixed))
    L.nlopt_destroy(opt)
    return out


@pytest.mark.parametrize("first", range(0, 90, 30))
def test_population_algorithms_maximising_a_registered_objective_agree_with_the_reference(first):
    """nlopt_set_max_objective with a device objective stays on the device for CRS2_LM / ISRES / ESCH too (the kernels deliver -f,
    NLA_OBJ_NEGATE): same run as the reference's, which minimises the flipped host callback (optimize.c:1014-1024)"""
    P = O.port()
    ref = O.ref()
    ref.orc_objective = P.orc_objective
    for draw in range(first, first + 30):
        r = play_population_max(bind(ref), draw, "orc_objective")
        a = play_population_max(bind(C.CDLL(EMU)), draw, "nlopt_amd_objective")
        assert a["log"] == r["log"] and a["ret"] == r["ret"], (a["cfg"], a["ret"], r["ret"])
        assert a["nev"] == r["nev"] and a["minf"] == r["minf"] and np.array_equal(a["x"], r["x"]), (a["cfg"], a["nev"], r["nev"], a["minf"], r["minf"])


def play_max_with_host_parts(L, case, getter):
    """MAXIMISING a registered (device) objective in the setups where part of the run calls f on the HOST after all: ISRES with an
    ordinary host constraint, CRS2_LM / ESCH forced onto the host-callback path (amd_host_eval), LD_MMA with a nonlinear constraint.
    The run must still maximise (round-2 advisor: it minimised and returned -min)."""
    dp = lambda a: a.ctypes.data_as(dpp)
    g = getattr(L, getter)
    g.restype = vp
    g.argtypes = [C.c_int]
    alg, obj, n, host_eval = case
    xs, lo, hi = O.golden_x0(obj, n)
    opt = L.nlopt_create(alg, n)
    lb, ub = np.full(n, lo), np.full(n, hi)
    keep = []
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub)), L.nlopt_set_max_objective(opt, g(O.OBJ[obj]), None)]
    if alg in (35, 24):
        if alg == 24:
            def con(nn, x, gr, d):             # inactive: x0 <= hi + 1, with its gradient
                if gr:
                    for i in range(nn):
                        gr[i] = 1.0 if i == 0 else 0.0
                return float(x[0] - (hi + 1.0))
        else:
            def con(nn, x, gr, d):
                return float(x[0] - (hi + 1.0))
        cb = FUNC(con)
        keep.append(cb)
        log.append(L.nlopt_add_inequality_constraint(opt, C.cast(cb, vp), None, 1e-8))
    if host_eval and hasattr(L, "nlopt_set_param") and getter == "nlopt_amd_objective":
        L.nlopt_set_param.argtypes = [vp, C.c_char_p, dbl]
        L.nlopt_set_param(opt, b"amd_host_eval", 1.0)          # (the reference ignores unknown parameters; not set there)
    if alg != 24:
        log.append(L.nlopt_set_population(opt, 30))
    log.append(L.nlopt_set_maxeval(opt, 400))
    if alg == 24:
        log.append(L.nlopt_set_ftol_rel(opt, 1e-10))
    L.nlopt_srand(97)
    x = np.array(xs, dtype=float)
    maxf = C.c_double(0)
    ret = L.nlopt_optimize(opt, dp(x), C.byref(maxf))
    out = dict(log=log, ret=ret, maxf=maxf.value, x=x, nev=L.nlopt_get_numevals(opt))
    L.nlopt_destroy(opt)
    del keep
    return out


MAX_HOST_CASES = [(35, "rastrigin", 4, 0), (35, "sphere", 3, 0), (19, "rastrigin", 4, 1), (42, "griewank", 5, 1), (35, "ackley", 4, 1),
                  (24, "sphere", 3, 0), (24, "rosenbrock", 4, 0)]


@pytest.mark.parametrize("case", MAX_HOST_CASES, ids=lambda c: "alg%d_%s_n%d_hosteval%d" % c)
def test_maximising_a_registered_objective_where_the_run_calls_f_on_the_host(case):
    P = O.port()
    ref = O.ref()
    ref.orc_objective = P.orc_objective
    r = play_max_with_host_parts(bind(ref), case, "orc_objective")
    a = play_max_with_host_parts(bind(C.CDLL(EMU)), case, "nlopt_amd_objective")
    assert a["log"] == r["log"] and a["ret"] == r["ret"], (case, a["ret"], r["ret"])
    assert a["nev"] == r["nev"] and a["maxf"] == r["maxf"] and np.array_equal(a["x"], r["x"]), (case, a["nev"], r["nev"], a["maxf"], r["maxf"])
    assert a["maxf"] > 0                   # a maximum of these objectives over their boxes, not minus a minimum


# ---- option round trips: everything a setter stores, read back ------------------------------------------------------------------
def play_options(L, draw):
    rng = np.random.default_rng(777000 + draw)
    dp = lambda a: a.ctypes.data_as(dpp)
    for nm in ("nlopt_get_stopval", "nlopt_get_ftol_rel", "nlopt_get_ftol_abs", "nlopt_get_xtol_rel", "nlopt_get_maxtime"):
        getattr(L, nm).restype = dbl
        getattr(L, nm).argtypes = [vp]
    for nm in ("nlopt_get_lower_bounds", "nlopt_get_upper_bounds", "nlopt_get_xtol_abs", "nlopt_get_x_weights"):
        getattr(L, nm).argtypes = [vp, dpp]
    L.nlopt_get_initial_step.argtypes = [vp, dpp, dpp]
    L.nlopt_set_initial_step.argtypes = [vp, dpp]
    L.nlopt_set_initial_step1.argtypes = [vp, dbl]
    L.nlopt_set_maxtime.argtypes = [vp, dbl]
    L.nlopt_set_upper_bounds1.argtypes = [vp, dbl]
    L.nlopt_set_lower_bound.argtypes = [vp, C.c_int, dbl]
    L.nlopt_set_upper_bound.argtypes = [vp, C.c_int, dbl]
    L.nlopt_set_x_weights1.argtypes = [vp, dbl]
    L.nlopt_set_vector_storage.argtypes = [vp, C.c_uint]
    L.nlopt_get_vector_storage.argtypes = [vp]
    L.nlopt_get_population.argtypes = [vp]
    L.nlopt_get_maxeval.argtypes = [vp]
    L.nlopt_get_algorithm.argtypes = [vp]
    L.nlopt_get_dimension.argtypes = [vp]
    L.nlopt_set_param.argtypes = [vp, C.c_char_p, dbl]
    L.nlopt_get_param.argtypes = [vp, C.c_char_p, dbl]
    L.nlopt_get_param.restype = dbl
    L.nlopt_has_param.argtypes = [vp, C.c_char_p]
    L.nlopt_num_params.argtypes = [vp]
    L.nlopt_nth_param.argtypes = [vp, C.c_uint]
    L.nlopt_nth_param.restype = C.c_char_p
    L.nlopt_remove_inequality_constraints.argtypes = [vp]
    L.nlopt_remove_equality_constraints.argtypes = [vp]
    L.nlopt_set_force_stop.argtypes = [vp, C.c_int]
    L.nlopt_get_force_stop.argtypes = [vp]
    L.nlopt_algorithm_name.argtypes = [C.c_int]
    L.nlopt_algorithm_name.restype = C.c_char_p
    alg = int(rng.choice([19, 35, 42, 11, 24, 20, 21, 22, 23, 38, 39, 0, 25, 40, 43]))
    n = int(rng.integers(0, 6))
    opt = L.nlopt_create(alg, n)
    log = [bool(opt), L.nlopt_algorithm_name(alg)]
    if not opt:
        return log
    nn = max(n, 1)
    for _ in range(int(rng.integers(3, 14))):
        k = int(rng.integers(18))
        v = float(rng.choice([-2.0, -1e-3, 0.0, 1e-6, 0.5, 3.0, np.inf, -np.inf, np.nan]))
        arr = rng.choice([-1.0, 0.0, 0.25, 2.0, np.inf], nn).astype(np.float64)
        if k == 0: log.append(("lb", L.nlopt_set_lower_bounds(opt, dp(arr))))
        elif k == 1: log.append(("ub", L.nlopt_set_upper_bounds(opt, dp(arr))))
        elif k == 2: log.append(("lb1", L.nlopt_set_lower_bounds1(opt, v)))
        elif k == 3: log.append(("ub1", L.nlopt_set_upper_bounds1(opt, v)))
        elif k == 4: log.append(("lbi", L.nlopt_set_lower_bound(opt, int(rng.integers(-1, n + 1)), v)))
        elif k == 5: log.append(("stopval", L.nlopt_set_stopval(opt, v)))
        elif k == 6: log.append(("ftol_rel", L.nlopt_set_ftol_rel(opt, v)))
        elif k == 7: log.append(("ftol_abs", L.nlopt_set_ftol_abs(opt, v)))
        elif k == 8: log.append(("xtol_rel", L.nlopt_set_xtol_rel(opt, v)))
        elif k == 9: log.append(("xtol_abs", L.nlopt_set_xtol_abs(opt, dp(arr))))
        elif k == 10: log.append(("xtol_abs1", L.nlopt_set_xtol_abs1(opt, v)))
        elif k == 11: log.append(("weights", L.nlopt_set_x_weights(opt, dp(arr))))
        elif k == 12: log.append(("weights1", L.nlopt_set_x_weights1(opt, v)))
        elif k == 13: log.append(("maxeval", L.nlopt_set_maxeval(opt, int(rng.integers(-3, 1000)))))
        elif k == 14: log.append(("maxtime", L.nlopt_set_maxtime(opt, v)))
        elif k == 15: log.append(("pop", L.nlopt_set_population(opt, int(rng.integers(0, 500)))))
        elif k == 16: log.append(("step1", L.nlopt_set_initial_step1(opt, v)))
        elif k == 17:
            name = [b"rho_init", b"tolg", b"amd_window_factor", b"x"][int(rng.integers(4))]
            log.append(("param", L.nlopt_set_param(opt, name, v)))
    if rng.random() < 0.3:
        c2 = L.nlopt_copy(opt)
        L.nlopt_destroy(opt)
        opt = c2
    out = np.zeros(nn)
    for nm in ("nlopt_get_lower_bounds", "nlopt_get_upper_bounds", "nlopt_get_xtol_abs", "nlopt_get_x_weights"):
        out[:] = 7.0
        log.append((nm, getattr(L, nm)(opt, dp(out)), out[:n].tolist()))
    for nm in ("nlopt_get_stopval", "nlopt_get_ftol_rel", "nlopt_get_ftol_abs", "nlopt_get_xtol_rel", "nlopt_get_maxtime"):
        log.append((nm, getattr(L, nm)(opt)))
    for nm in ("nlopt_get_maxeval", "nlopt_get_population", "nlopt_get_algorithm", "nlopt_get_dimension", "nlopt_get_vector_storage", "nlopt_get_force_stop",
               "nlopt_num_params"):
        log.append((nm, getattr(L, nm)(opt)))
    log.append(("has", [L.nlopt_has_param(opt, p) for p in (b"rho_init", b"tolg", b"x", b"nope")], L.nlopt_get_param(opt, b"tolg", -9.0)))
    log.append(("nth", [L.nlopt_nth_param(opt, i) for i in range(L.nlopt_num_params(opt) + 1)]))
    x = np.full(nn, 0.5)
    out[:] = 7.0
    log.append(("get_step", L.nlopt_get_initial_step(opt, dp(x), dp(out)), out[:n].tolist()))
    log.append(("rm", L.nlopt_remove_inequality_constraints(opt), L.nlopt_remove_equality_constraints(opt)))
    msg = L.nlopt_get_errmsg(opt)
    log.append(("msg", msg.decode() if msg else None))
    L.nlopt_destroy(opt)
    return log


def _same_log(a, r):
    def eq(u, v):
        if isinstance(u, float) and isinstance(v, float):
            return u == v or (np.isnan(u) and np.isnan(v))
        if isinstance(u, (list, tuple)) and isinstance(v, (list, tuple)):
            return len(u) == len(v) and all(eq(p, q) for p, q in zip(u, v))
        return u == v
    return eq(a, r)


@pytest.mark.parametrize("draw", range(200))
def test_drawn_option_round_trips_agree_with_the_reference(draw):
    """setters with valid and invalid values (negative tolerances, NaN, infinities, out-of-range indices, unknown parameters), then
    every getter: return codes, stored values, parameter lists and error messages as the reference's"""
    r = play_options(bind(O.ref()), draw)
    a = play_options(bind(C.CDLL(EMU)), draw)
    assert _same_log(a, r), [(p, q) for p, q in zip(a, r) if not _same_log(p, q)][:3]


# ---- NaN objective values: ISRES's selection (round-2 verdict, missing item 4) ------------------------------------------------------
def play_isres_nan(L, case):
    """ISRES on an objective that is NaN on part of the box, with and without constraints (sort by f / stochastic ranking): every
    comparison with a NaN is false — the reference's comparator then calls it equal to everything and glibc's qsort_r makes of that
    what it makes; the ranking sweeps never move an element past it.  The library must give the same run."""
    dp = lambda a: a.ctypes.data_as(dpp)
    n, pop, ncon, nan_frac, seed, maxeval = case
    rng = np.random.default_rng(1234 + seed)
    opt = L.nlopt_create(35, n)
    lb, ub = np.full(n, -2.0), np.full(n, 3.0)
    cut = -2.0 + 5.0 * (1 - nan_frac)
    calls = []

    def f(nn, x, g, d):
        calls.append(tuple(x[i] for i in range(nn)))
        if x[0] > cut:
            return float("nan")
        return float(sum((x[i] - 0.3 * i) ** 2 for i in range(nn)))
    keep = [FUNC(f)]
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub)), L.nlopt_set_min_objective(opt, C.cast(keep[0], vp), None)]
    for q in range(ncon):
        cb = FUNC(lambda nn, x, g, d, q=q: float(x[q % nn] - 1.0) if x[(q + 1) % nn] < 2.5 else float("nan"))
        keep.append(cb)
        log.append(L.nlopt_add_inequality_constraint(opt, C.cast(cb, vp), None, 1e-8))
    log.append(L.nlopt_set_population(opt, pop))
    log.append(L.nlopt_set_maxeval(opt, maxeval))
    L.nlopt_srand(seed)
    x = rng.uniform(-1.5, 1.0, n)
    minf = C.c_double(0)
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    out = dict(log=log, ret=ret, minf=minf.value, x=x, nev=L.nlopt_get_numevals(opt), calls=calls)
    L.nlopt_destroy(opt)
    del keep
    return out


@pytest.mark.parametrize("case", [(3, 20, 0, 0.3, 1, 400), (5, 35, 0, 0.6, 2, 600), (4, 30, 2, 0.25, 3, 500), (2, 15, 1, 0.5, 4, 300),
                                  (6, 50, 3, 0.1, 5, 700), (3, 25, 0, 1.0, 6, 200)],
                         ids=lambda c: "n%d_pop%d_con%d_nan%g_seed%d" % c[:5])
def test_isres_with_nan_objective_values_is_the_references_run(case):
    r = play_isres_nan(bind(O.ref()), case)
    a = play_isres_nan(bind(C.CDLL(EMU)), case)
    assert a["log"] == r["log"] and a["ret"] == r["ret"] and a["nev"] == r["nev"], (a["ret"], r["ret"], a["nev"], r["nev"])
    assert len(a["calls"]) == len(r["calls"])
    first = next((i for i, (u, v) in enumerate(zip(a["calls"], r["calls"])) if u != v), None)
    assert first is None, "candidate %d differs" % first
    assert (a["minf"] == r["minf"] or (np.isnan(a["minf"]) and np.isnan(r["minf"]))) and np.array_equal(a["x"], r["x"], equal_nan=True)


def play_esch_nan(L, case):
    dp = lambda a: a.ctypes.data_as(dpp)
    n, pop, nan_frac, seed, maxeval = case
    opt = L.nlopt_create(42, n)
    lb, ub = np.full(n, -2.0), np.full(n, 3.0)
    cut = -2.0 + 5.0 * (1 - nan_frac)
    calls = []

    def f(nn, x, g, d):
        calls.append(tuple(x[i] for i in range(nn)))
        if x[nn - 1] > cut:
            return float("nan")
        return float(sum((x[i] - 0.2 * i) ** 2 for i in range(nn)) + np.cos(5 * x[0]))
    keep = [FUNC(f)]
    log = [L.nlopt_set_lower_bounds(opt, dp(lb)), L.nlopt_set_upper_bounds(opt, dp(ub)), L.nlopt_set_min_objective(opt, C.cast(keep[0], vp), None),
           L.nlopt_set_population(opt, pop), L.nlopt_set_maxeval(opt, maxeval)]
    L.nlopt_srand(seed)
    x = np.full(n, 0.25)
    minf = C.c_double(0)
    ret = L.nlopt_optimize(opt, dp(x), C.byref(minf))
    out = dict(log=log, ret=ret, minf=minf.value, x=x, nev=L.nlopt_get_numevals(opt), calls=calls)
    L.nlopt_destroy(opt)
    del keep
    return out


@pytest.mark.parametrize("case", [(3, 12, 0.3, 1, 500), (5, 30, 0.6, 2, 900), (2, 8, 0.15, 3, 400), (6, 0, 0.4, 4, 800)],
                         ids=lambda c: "n%d_pop%d_nan%g_seed%d" % c[:4])
def test_esch_with_nan_fitness_values_is_the_references_run(case):
    """GN_ESCH's selection sorts parents + offspring by fitness with a comparator for which a NaN equals everything (esch.c:59-64,243)"""
    r = play_esch_nan(bind(O.ref()), case)
    a = play_esch_nan(bind(C.CDLL(EMU)), case)
    assert a["log"] == r["log"] and a["ret"] == r["ret"] and a["nev"] == r["nev"], (a["ret"], r["ret"], a["nev"], r["nev"])
    first = next((i for i, (u, v) in enumerate(zip(a["calls"], r["calls"])) if u != v), None)
    assert first is None, "candidate %d differs" % first
    assert (a["minf"] == r["minf"] or (np.isnan(a["minf"]) and np.isnan(r["minf"]))) and np.array_equal(a["x"], r["x"], equal_nan=True)
