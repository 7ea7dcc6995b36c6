"""-m gpu: the device local optimisers in EXACT-ORDER mode (nlopt_set_param "amd_exact_dot" = 1) against
the CPU oracle, evaluation by evaluation.

Default mode sums dot products in a workgroup tree, so iterates agree with the reference to rounding only.  Exact mode
accumulates every dot product / norm / objective sum in the reference's sequential order (mssubs.c:601-641, stop.c:37-57,
mma.c:74-121), which turns the comparison into a proof that the device PLIS / MMA LOGIC is the reference's:

  host objective (the oracle's C callbacks as nlopt_func)   every x handed to the callback and every f are BIT-IDENTICAL to
        the port's (which is pinned bit for bit to the real reference), same evaluation count, result, minimiser
  device objective                                          the per-evaluation f sequence within 1e-10 relative (device libm
        vs glibc in sin / cos / exp), identical evaluation counts and result codes
"""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = pytest.mark.gpu
RTOL = 1e-10

CASES = [
    ("sphere", 8, dict()),
    ("rosenbrock", 10, dict(maxeval=2000)),
    ("rosenbrock", 2, dict(ftol_rel=1e-10)),
    ("ackley", 30, dict(ftol_rel=1e-8)),
    ("rastrigin", 20, dict(ftol_rel=1e-8)),
    ("griewank", 12, dict(xtol_rel=1e-6)),
    ("levy", 7, dict(ftol_abs=1e-12)),
    ("ackley", 200, dict(ftol_rel=1e-8, mf=5)),
    ("rastrigin", 64, dict(maxeval=37)),
    ("sphere", 6, dict(stopval=1e-3)),
    ("rastrigin", 1000, dict(ftol_rel=1e-9)),
    ("ackley", 4096, dict(ftol_rel=1e-8)),          # the config-4 shape: n = 4096, 320 history pairs
    ("griewank", 1500, dict(ftol_rel=1e-9)),        # n > 1024: the exact sums run over several LDS chunks
]


def recorder(obj, cap):
    """the oracle's recording wrapper around its own C objective: an nlopt_func that logs f and the hash of x of every call"""
    L = O.port()
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = O.Recorder(L.orc_objective(O.OBJ[obj]), None, O.dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    return rec, fbuf, hbuf, C.cast(L.orc_recording_callback, C.c_void_p).value


def run_amd(alg, obj, n, host, x0=None, lb=None, ub=None, maxeval=0, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0, stopval=None, mf=0,
            params=None, step=None, exact=True, xtol_abs=None, x_weights=None):
    assert nlopt_amd.device_count() > 0
    L = nlopt_amd.lib()
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(lo if lb is None else lb)
    o.set_upper_bounds(hi if ub is None else ub)
    cap = 2 * (maxeval or 200000) + 64
    rec = None
    if host:
        rec, fbuf, hbuf, cb = recorder(obj, cap)
        o.set_min_objective(cb, C.cast(C.pointer(rec), C.c_void_p))
    else:
        o.set_min_objective(nlopt_amd.objective(obj))
        o.enable_trace(cap)
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
        L.nlopt_set_vector_storage(o._h, mf)
    if step is not None:
        L.nlopt_set_initial_step1.argtypes = [C.c_void_p, C.c_double]
        assert L.nlopt_set_initial_step1(o._h, float(step)) > 0
    if xtol_abs is not None:
        L.nlopt_set_xtol_abs.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        assert L.nlopt_set_xtol_abs(o._h, O.dptr(np.asarray(xtol_abs, dtype=np.float64))) > 0
    if x_weights is not None:
        L.nlopt_set_x_weights.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        assert L.nlopt_set_x_weights(o._h, O.dptr(np.asarray(x_weights, dtype=np.float64))) > 0
    for k, v in (params or {}).items():
        o.set_param(k, v)
    if exact:
        o.set_param("amd_exact_dot", 1)
    x, minf, ret = o.optimize_raw(xs if x0 is None else x0)
    out = dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), err=o.get_errmsg())
    if host:
        out["fseq"], out["xhash"] = fbuf[:rec.len].copy(), hbuf[:rec.len].copy()
    else:
        out["fseq"] = o.trace()["f"]
    return out


def same_sequence(a, p, host):
    assert a["ret"] == p["ret"], (a["ret"], p["ret"], a["err"])
    assert a["nevals"] == p["nevals"], (a["nevals"], p["nevals"])
    assert len(a["fseq"]) == len(p["fseq"])
    if host:
        assert np.array_equal(a["xhash"], p["xhash"])            # every point handed to the callback, bit for bit
        assert np.array_equal(a["fseq"], p["fseq"])
        assert a["minf"] == p["minf"] and np.array_equal(a["x"], p["x"])
    else:
        scale = np.abs(p["fseq"]).max()
        bad = np.flatnonzero(np.abs(a["fseq"] - p["fseq"]) > RTOL * np.maximum(np.abs(p["fseq"]), 1e-6 * scale))
        assert len(bad) == 0, "first differing evaluation %d of %d: %r vs %r" % (bad[0], len(p["fseq"]), a["fseq"][bad[0]], p["fseq"][bad[0]])
        assert abs(a["minf"] - p["minf"]) <= RTOL * max(abs(p["minf"]), 1e-6 * scale)
        assert np.allclose(a["x"], p["x"], rtol=1e-8, atol=1e-9 * max(np.abs(p["x"]).max(), 1.0))


@pytest.mark.parametrize("host", [True, False], ids=["host-objective", "device-objective"])
@pytest.mark.parametrize("obj,n,kw", CASES)
def test_lbfgs_exact_mode_follows_the_oracle_evaluation_by_evaluation(obj, n, kw, host):
    a = run_amd(nlopt_amd.LD_LBFGS, obj, n, host, **kw)
    p = O.run_port_lbfgs(obj, n, **kw)
    same_sequence(a, p, host)


@pytest.mark.parametrize("host", [True, False], ids=["host-objective", "device-objective"])
def test_lbfgs_exact_mode_active_bounds_weights_and_xtol_abs(host):
    rng = np.random.default_rng(3)
    for n, obj in ((9, "sphere"), (14, "rastrigin"), (6, "rosenbrock"), (300, "ackley")):
        lb = -rng.uniform(0.1, 2.0, n)
        ub = rng.uniform(0.1, 2.0, n)
        lb[::3] = 0.3
        ub[::3] = 2.5
        x0 = np.clip(rng.uniform(-2, 2, n), lb, ub)
        x0[1] = ub[1]
        a = run_amd(nlopt_amd.LD_LBFGS, obj, n, host, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        p = O.run_port_lbfgs(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        same_sequence(a, p, host)
    # x_weights / xtol_abs (silently ignored before; stop.c:110-120).  The reference run for comparison.
    n = 4
    w = np.array([50.0, 0.01, 1.0, 1.0])
    x0 = np.array([2.0, -3.0, 1.5, 0.5])
    for kw in (dict(xtol_rel=0.1, x_weights=w), dict(xtol_rel=1e-3, xtol_abs=np.full(n, 0.0)), dict(xtol_abs=np.full(n, 1e-3), xtol_rel=1e-12)):
        a = run_amd(nlopt_amd.LD_LBFGS, "sphere", n, host, x0=x0 * 3, lb=np.full(n, -10.0), ub=np.full(n, 10.0), **kw)

        def setup(R, opt, kw=kw):
            R.nlopt_set_x_weights.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
            if "x_weights" in kw:
                assert R.nlopt_set_x_weights(opt, O.dptr(kw["x_weights"])) > 0
            if "xtol_abs" in kw:
                assert R.nlopt_set_xtol_abs(opt, O.dptr(kw["xtol_abs"])) > 0
            R.nlopt_set_lower_bounds(opt, O.dptr(np.full(n, -10.0)))
            R.nlopt_set_upper_bounds(opt, O.dptr(np.full(n, 10.0)))
        r = O.run_ref(11, "sphere", n, 0, 0, x0=x0 * 3, xtol_rel=kw.get("xtol_rel", 0.0), setup=setup)
        same_sequence(a, r, host)


MMA_CASES = [
    ("sphere", 8, dict(ftol_rel=1e-10)),
    ("rosenbrock", 6, dict(maxeval=800)),
    ("ackley", 30, dict(ftol_rel=1e-8)),
    ("rastrigin", 20, dict(xtol_rel=1e-6)),
    ("griewank", 12, dict(ftol_rel=1e-9)),
    ("levy", 7, dict(ftol_abs=1e-10)),
    ("rastrigin", 64, dict(maxeval=37)),
    ("ackley", 700, dict(ftol_rel=1e-7)),
    ("ackley", 4096, dict(ftol_rel=1e-6)),
    ("rastrigin", 50, dict(ftol_rel=1e-9, params=dict(inner_gradients=0))),
    ("rosenbrock", 9, dict(maxeval=600, params=dict(always_improve=0, inner_maxeval=2))),
    ("ackley", 40, dict(ftol_rel=1e-9, params=dict(inner_gradients=0, inner_maxeval=3, rho_init=4.0))),
    ("sphere", 5, dict(ftol_rel=1e-9, params=dict(sigma_min=0.05), step=0.4)),
]


@pytest.mark.parametrize("host", [True, False], ids=["host-objective", "device-objective"])
@pytest.mark.parametrize("obj,n,kw", MMA_CASES)
def test_mma_exact_mode_follows_the_oracle_evaluation_by_evaluation(obj, n, kw, host):
    a = run_amd(nlopt_amd.LD_MMA, obj, n, host, **kw)
    p = O.run_port_mma(obj, n, **kw)
    same_sequence(a, p, host)


def test_default_mode_drift_is_rounding_level():
    """what the default (tree) mode costs in parity, measured: same result code, minimum within 1e-8, evaluation counts
    within a few of the exact-order run"""
    worst = 0
    for obj, n, kw in CASES:
        e = run_amd(nlopt_amd.LD_LBFGS, obj, n, False, **kw)
        d = run_amd(nlopt_amd.LD_LBFGS, obj, n, False, exact=False, **kw)
        assert e["ret"] == d["ret"]
        assert abs(e["minf"] - d["minf"]) <= 1e-8 * max(abs(e["minf"]), 1e-300) + 1e-12
        worst = max(worst, abs(e["nevals"] - d["nevals"]))
    assert worst <= 4
    print("default-mode drift in evaluation count over %d cases: at most %d" % (len(CASES), worst))


@pytest.mark.parametrize("obj,n,ns,seed,kw", [
    ("rastrigin", 4, 10, 42, dict(stopval=1e-6, maxeval=100000)),
    ("ackley", 6, 25, 7, dict(stopval=1e-5, maxeval=20000)),
    ("griewank", 5, 0, 3, dict(stopval=1e-7, maxeval=100000)),
    ("rastrigin", 8, 40, 5, dict(stopval=1.5, maxeval=100000)),
    ("rosenbrock", 4, 12, 9, dict(stopval=1e-10, maxeval=100000, mf=3)),
    ("ackley", 64, 60, 11, dict(maxeval=9000)),
    ("griewank", 200, 100, 2, dict(maxeval=12000, local_ftol_rel=1e-6)),
])
@pytest.mark.parametrize("host", [True, False], ids=["host-objective", "device-objective"])
@pytest.mark.parametrize("local", ["lbfgs", "mma"])
def test_mlsl_exact_mode_is_the_oracles_run(obj, n, ns, seed, kw, local, host):
    """G_MLSL in exact-order mode: no 'rugged' slack — the same samples, the same local searches in the same order, each with
    the oracle's evaluation count, the same stop (incl. runs cut by MAXEVAL).  Host objective (the oracle's C callback):
    every call bit for bit.  Device objective: f within 1e-10; on Griewank (cosine product, tan in the gradient) a last-bit
    libm difference can still move a line-search decision of a long search — there the evaluation counts may drift."""
    L = nlopt_amd.lib()
    kw = dict(kw)
    if local == "mma":
        kw.pop("mf", None)
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(nlopt_amd.G_MLSL, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    rec = None
    if host:
        rec, fbuf, hbuf, cb = recorder(obj, kw["maxeval"] * 2 + 8192)
        o.set_min_objective(cb, C.cast(C.pointer(rec), C.c_void_p))
    else:
        o.set_min_objective(nlopt_amd.objective(obj))
    loc = nlopt_amd.Opt(nlopt_amd.LD_MMA if local == "mma" else nlopt_amd.LD_LBFGS, n)
    loc.set_ftol_rel(kw.get("local_ftol_rel", 1e-8))
    if kw.get("mf"):
        L.nlopt_set_vector_storage(loc._h, kw["mf"])
    assert L.nlopt_set_local_optimizer(o._h, loc._h) > 0
    if ns:
        o.set_population(ns)
    o.set_maxeval(kw["maxeval"])
    if "stopval" in kw:
        o.set_stopval(kw["stopval"])
    o.set_param("amd_exact_dot", 1)
    o.enable_trace(kw["maxeval"] + 4096)
    nlopt_amd.srand(seed)
    x, minf, ret = o.optimize_raw(xs)
    p = O.run_port_mlsl(obj, n, ns, seed, local=local, **kw)
    t = o.trace()
    if host:
        assert ret == p["ret"] and o.get_numevals() == p["nevals"]
        assert np.array_equal(hbuf[:rec.len], p["xhash"]) and np.array_equal(fbuf[:rec.len], p["fseq"])
        assert minf == p["minf"] and np.array_equal(x, p["x"])
        assert o.stats()["mt_words"] == p["words"]
        return
    rugged = obj == "griewank"
    assert ret == p["ret"], (ret, p["ret"], o.get_errmsg())
    fl = t[t["kind"] == 4]
    if not rugged:
        assert o.get_numevals() == p["nevals"]
        assert o.stats()["mt_words"] == p["words"]
        fs = t[t["kind"] == 3]["f"]
        assert len(fs) == len(p["fsamp"]) and np.all(np.abs(fs - p["fsamp"]) <= RTOL * np.maximum(np.abs(p["fsamp"]), 1.0))
        assert len(fl) == len(p["floc"])
        assert np.array_equal(fl["accepted"], p["eloc"])
        assert np.all(np.abs(fl["f"] - p["floc"]) <= 1e-9 * np.maximum(np.abs(p["floc"]), 1.0))
    else:
        assert abs(len(fl) - len(p["floc"])) <= max(2, len(p["floc"]) // 10)
    assert abs(minf - p["minf"]) <= 1e-8 * max(abs(p["minf"]), 1.0)
    assert np.allclose(x, p["x"], rtol=1e-6, atol=1e-7 * max(np.abs(p["x"]).max(), 1.0))


@pytest.mark.parametrize("alg,obj,n", [(nlopt_amd.LD_LBFGS, "sphere", 3), (nlopt_amd.LD_MMA, "sphere", 3), (nlopt_amd.LD_LBFGS, "rastrigin", 12),
                                       (nlopt_amd.LD_MMA, "ackley", 9)])
def test_maximisation_keeps_the_device_objective(alg, obj, n):
    """nlopt_set_max_objective with a registered device objective (the flip wrapper hid it and LD_LBFGS / LD_MMA / MLSL
    refused): f and its gradient are negated on the device, as the reference's f_max wrapper does on the host
    (optimize.c:970-980,1014-1024).  Against the REAL reference maximising the same function."""
    if not O.have_ref():
        pytest.skip("oracle/_ref not built")
    R, P = O.ref(), O.port()
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_max_objective(nlopt_amd.objective(obj))
    o.set_ftol_rel(1e-9)
    o.set_maxeval(300)
    o.set_param("amd_exact_dot", 1)
    x, maxf, ret = o.optimize_raw(xs)
    opt = R.nlopt_create(alg, n)
    lb, ub = np.full(n, lo), np.full(n, hi)
    R.nlopt_set_lower_bounds(opt, O.dptr(lb))
    R.nlopt_set_upper_bounds(opt, O.dptr(ub))
    R.nlopt_set_max_objective(opt, P.orc_objective(O.OBJ[obj]), None)
    R.nlopt_set_ftol_rel(opt, 1e-9)
    R.nlopt_set_maxeval(opt, 300)
    xr = np.array(xs)
    mf = C.c_double()
    rret = R.nlopt_optimize(opt, O.dptr(xr), C.byref(mf))
    nev = R.nlopt_get_numevals(opt)
    R.nlopt_destroy(opt)
    assert ret == rret, (ret, rret, o.get_errmsg())
    assert o.get_numevals() == nev
    assert abs(maxf - mf.value) <= 1e-10 * max(abs(mf.value), 1.0)
    assert np.allclose(x, xr, rtol=1e-9, atol=1e-10)


def test_mlsl_maximisation_on_the_device():
    """G_MLSL maximising a device objective: samples and local searches stay on the device (sign applied there)"""
    n = 5
    xs, lo, hi = O.golden_x0("rastrigin", n)
    L = nlopt_amd.lib()
    res = []
    for device in (True, False):
        o = nlopt_amd.Opt(nlopt_amd.G_MLSL, n)
        o.set_lower_bounds(lo)
        o.set_upper_bounds(hi)
        if device:
            o.set_max_objective(nlopt_amd.objective("rastrigin"))
        else:
            rec, fbuf, hbuf, cb = recorder("rastrigin", 20000)
            o.set_max_objective(cb, C.cast(C.pointer(rec), C.c_void_p))       # the same function as a host callback
        loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
        loc.set_ftol_rel(1e-8)
        assert L.nlopt_set_local_optimizer(o._h, loc._h) > 0
        o.set_population(12)
        o.set_maxeval(1500)
        o.set_param("amd_exact_dot", 1)
        nlopt_amd.srand(5)
        x, maxf, ret = o.optimize_raw(xs)
        res.append((ret, maxf, x, o.get_numevals(), o.stats()))
    (r1, f1, x1, n1, s1), (r2, f2, x2, n2, s2) = res
    assert r1 == r2 and n1 == n2
    assert abs(f1 - f2) <= 1e-9 * max(abs(f2), 1.0) and np.allclose(x1, x2, rtol=1e-8, atol=1e-9)
    assert s1["lbfgs_launches"] >= 1 and f1 > 0
