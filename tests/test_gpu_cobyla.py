"""-m gpu twin of tests/test_cobyla_differential.py: NLOPT_LN_COBYLA and NLOPT_GN_MLSL(_LDS) with its default local optimiser
through libnlopt_amd.so on the MI355X against the real reference — identical objective calls, results and counts (COBYLA on the host for callbacks;
batched on the device for a compiled-in objective, round 6; MLSL's samples, distances and bookkeeping on the device)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
import test_cobyla_differential as T

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")]


def test_cobyla_on_the_product_library_is_the_references_run():
    assert nlopt_amd.device_count() > 0
    R, A = T.more_bind(O.ref()), T.more_bind(C.CDLL(nlopt_amd.LIB_PATH))
    for draw in range(0, 60):
        T.same(T.play_cobyla(R, draw), T.play_cobyla(A, draw), draw)


def test_gn_mlsl_default_local_optimiser_on_the_device_library_is_the_references_run():
    assert nlopt_amd.device_count() > 0
    R, A = T.more_bind(O.ref()), T.more_bind(C.CDLL(nlopt_amd.LIB_PATH))
    for draw in range(0, 40):
        T.same(T.play_mlsl(R, draw), T.play_mlsl(A, draw), draw)


def run_gn_mlsl(lib, alg, obj, n, maxeval, params=(), xtol=1e-5, seed=77, stats=False, population=0):
    fptr = nlopt_amd.objective(obj)
    lo, hi = nlopt_amd.objective_box(obj)
    lib.nlopt_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
    opt = lib.nlopt_create(alg, n)
    lb, ub = np.full(n, lo), np.full(n, hi)
    lib.nlopt_set_lower_bounds(opt, T.dp(lb))
    lib.nlopt_set_upper_bounds(opt, T.dp(ub))
    lib.nlopt_set_min_objective(opt, C.cast(fptr, C.c_void_p), None)
    lib.nlopt_set_xtol_rel(opt, xtol)
    lib.nlopt_set_maxeval(opt, maxeval)
    if population:
        lib.nlopt_set_population.argtypes = [C.c_void_p, C.c_uint]
        lib.nlopt_set_population(opt, population)
    for k, v in params:
        assert lib.nlopt_set_param(opt, k.encode(), float(v)) > 0
    lib.nlopt_srand(seed)
    x = np.linspace(0.3 * lo, 0.4 * hi, n)
    minf = C.c_double(0)
    ret = lib.nlopt_optimize(opt, T.dp(x), C.byref(minf))
    out = dict(ret=ret, minf=minf.value, x=x.copy(), nevals=lib.nlopt_get_numevals(opt))
    if stats:
        st = nlopt_amd.Stats()
        lib.nlopt_amd_get_stats.argtypes = [C.c_void_p, C.POINTER(nlopt_amd.Stats)]
        lib.nlopt_amd_get_stats(opt, C.byref(st))
        out["launches"] = int(st.lbfgs_launches)
        out["searches"] = int(st.accepted)
        out["host_searches"] = int(st.cobyla_host_searches)
    lib.nlopt_destroy(opt)
    return out


@pytest.mark.parametrize("alg,obj,n", [(T.GN_MLSL_LDS, "rastrigin", 3), (T.GN_MLSL, "ackley", 2), (T.GN_MLSL_LDS, "griewank", 5)])
def test_gn_mlsl_with_a_registered_device_objective_on_the_host_path(alg, obj, n):
    """"amd_cobyla_host" = 1: COBYLA as a host algorithm (cobyla_host.c), the whole run on the objective's host twin — the
    reference's run evaluation by evaluation (same callback pointer given to both libraries)"""
    r = run_gn_mlsl(T.more_bind(O.ref()), alg, obj, n, 700)
    a = run_gn_mlsl(T.more_bind(C.CDLL(nlopt_amd.LIB_PATH)), alg, obj, n, 700, params=[("amd_cobyla_host", 1)], stats=True)
    assert (a["ret"], a["minf"], a["nevals"]) == (r["ret"], r["minf"], r["nevals"]) and np.array_equal(a["x"], r["x"]), (a, r)
    assert a["launches"] == 0


class _Params(C.Structure):
    _fields_ = [("minf_max", C.c_double), ("ftol_rel", C.c_double), ("ftol_abs", C.c_double), ("xtol_rel", C.c_double),
                ("maxeval", C.c_int32), ("exact", C.c_int32), ("sign", C.c_double), ("xtol_abs", C.c_void_p), ("abort", C.c_void_p), ("done", C.c_void_p)]


class _Result(C.Structure):
    _fields_ = [("f", C.c_double), ("ret", C.c_int32), ("nevals", C.c_int32), ("iterm", C.c_int32), ("cols", C.c_int32)]


def kernel_batch(obj, n, starts, lb, ub, xtol_rel=1e-6, maxeval=0, dx=None, exact=1):
    """nla_k_cobyla_batch on the device: one wavefront per row of `starts`"""
    L = nlopt_amd.lib()
    count, ld = starts.shape[0], (n + 1) & ~1
    X = np.zeros((count, ld)); X[:, :n] = starts
    bX, bl, bu = nlopt_amd.DevBuf.from_array(X), nlopt_amd.DevBuf.from_array(np.asarray(lb, dtype=np.float64)), nlopt_amd.DevBuf.from_array(np.asarray(ub, dtype=np.float64))
    bd = nlopt_amd.DevBuf.from_array(np.asarray(dx, dtype=np.float64)) if dx is not None else None
    bw, bi, bo = nlopt_amd.DevBuf(8 * max(8, count)), nlopt_amd.DevBuf(4 * max(8, count)), nlopt_amd.DevBuf(C.sizeof(_Result) * count)
    P = _Params(-np.inf, 0.0, 0.0, xtol_rel, maxeval, exact, 1.0, None, None, None)
    vp = C.c_void_p
    L.nla_k_cobyla_batch.argtypes = [C.c_int] * 4 + [vp] * 6 + [C.POINTER(_Params), vp, vp]
    rc = L.nla_k_cobyla_batch(nlopt_amd.OBJECTIVES[obj], n, ld, count, bl.ptr, bu.ptr, bd.ptr if bd else None, bX.ptr, bw.ptr, bi.ptr, C.byref(P), bo.ptr, None)
    assert rc == 0, L.nla_dev_error_string(rc)
    assert L.nla_stream_sync(None) == 0
    raw = bo.to_array(np.uint8, C.sizeof(_Result) * count)
    res = (_Result * count).from_buffer_copy(raw.tobytes())
    return dict(x=bX.to_array(np.float64, count * ld).reshape(count, ld)[:, :n], f=np.array([r.f for r in res]), ret=[r.ret for r in res], nevals=[r.nevals for r in res])


def reference_cobyla(obj, n, starts, lb, ub, xtol_rel=1e-6, maxeval=0, dx=None):
    """the same starts one after another through the REAL reference's nlopt_optimize(LN_COBYLA), the objective = the host callback"""
    R = T.more_bind(O.ref())
    fptr = nlopt_amd.objective(obj)
    out = dict(x=[], f=[], ret=[], nevals=[])
    for s in starts:
        opt = R.nlopt_create(T.LN_COBYLA, n)
        R.nlopt_set_lower_bounds(opt, T.dp(np.asarray(lb, dtype=np.float64))); R.nlopt_set_upper_bounds(opt, T.dp(np.asarray(ub, dtype=np.float64)))
        R.nlopt_set_min_objective(opt, C.cast(fptr, C.c_void_p), None)
        R.nlopt_set_xtol_rel(opt, xtol_rel)
        if maxeval:
            R.nlopt_set_maxeval(opt, maxeval)
        if dx is not None:
            R.nlopt_set_initial_step.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
            R.nlopt_set_initial_step(opt, T.dp(np.asarray(dx, dtype=np.float64)))
        x = np.array(s, dtype=np.float64)
        minf = C.c_double(0)
        out["ret"].append(R.nlopt_optimize(opt, T.dp(x), C.byref(minf)))
        out["f"].append(minf.value); out["x"].append(x); out["nevals"].append(R.nlopt_get_numevals(opt))
        R.nlopt_destroy(opt)
    out["x"], out["f"] = np.array(out["x"]), np.array(out["f"])
    return out


@pytest.mark.parametrize("obj,n,count,maxeval,kind", [("sphere", 2, 6, 0, "plain"), ("rosenbrock", 3, 5, 4000, "plain"), ("sphere", 5, 6, 0, "onbound"),
                                                      ("rosenbrock", 4, 4, 3000, "halfinf"), ("sphere", 7, 4, 0, "steps"), ("rosenbrock", 12, 6, 2500, "plain"),
                                                      ("sphere", 24, 3, 0, "plain"), ("rosenbrock", 33, 3, 3000, "onbound"), ("sphere", 51, 2, 4000, "plain"),
                                                      ("rosenbrock", 16, 300, 600, "plain")])
def test_batched_cobyla_kernel_is_the_references_search_evaluation_by_evaluation(obj, n, count, maxeval, kind):
    """hip/cobyla_kernels.hip directly (one wavefront per start, lane-parallel sums in the reference's order) against the real
    reference's LN_COBYLA on the same starts: exact-order objective (no transcendental in sphere / Rosenbrock; the device's
    + - x / sqrt are IEEE) -> result code, evaluation count, f and the minimiser bit for bit.  n = 51 is the largest dimension whose
    state fits the LDS; the 300-start case fills more than one wavefront slot per compute unit"""
    rng = np.random.default_rng(1000 + n + count)
    lo, hi = nlopt_amd.objective_box(obj)
    lb, ub = np.full(n, lo), np.full(n, hi)
    starts = rng.uniform(lo, hi, (count, n))
    dx = None
    if kind == "onbound":
        starts[0, : max(1, n // 3)] = hi
        starts[-1, -1] = lo
    if kind == "halfinf":
        ub[0] = np.inf; lb[1] = -np.inf
        if n > 2:
            lb[2] = -np.inf; ub[2] = np.inf
    if kind == "steps":
        dx = np.linspace(0.3, 1.7, n) * 0.1 * (hi - lo)
    a = kernel_batch(obj, n, starts, lb, ub, maxeval=maxeval, dx=dx)
    r = reference_cobyla(obj, n, starts, lb, ub, maxeval=maxeval, dx=dx)
    assert a["ret"] == r["ret"] and a["nevals"] == r["nevals"], (a["ret"], r["ret"], a["nevals"], r["nevals"])
    assert np.array_equal(a["f"], r["f"]) and np.array_equal(a["x"], r["x"])


def test_batched_cobyla_kernel_refuses_a_dimension_whose_state_does_not_fit_the_lds():
    L = nlopt_amd.lib()
    assert L.nla_cobyla_fits(51) == 1 and L.nla_cobyla_fits(52) == 0
    with pytest.raises(AssertionError):
        kernel_batch("sphere", 52, np.zeros((1, 52)) + 0.5, np.full(52, -1.0), np.full(52, 1.0), maxeval=10)


@pytest.mark.parametrize("alg,obj,n,maxeval", [(T.GN_MLSL, "sphere", 4, 900), (T.GN_MLSL_LDS, "rosenbrock", 3, 1500), (T.GN_MLSL, "rosenbrock", 6, 3000),
                                               (T.GN_MLSL_LDS, "sphere", 24, 6000), (T.GN_MLSL, "rosenbrock", 40, 20000)])
def test_gn_mlsl_batched_device_cobyla_exact_order_is_the_references_run(alg, obj, n, maxeval):
    """round 6: GN_MLSL's local searches by the BATCHED device COBYLA (hip/cobyla_kernels.hip: one
    wavefront per start, the searches of a batch concurrent, committed in the reference's order).  With
    amd_exact_dot = 1 the objective is summed in the host callback's order; sphere and Rosenbrock have no transcendental, and the
    device's +, -, x, /, sqrt are IEEE (profiles/r02_fp_conformance.txt) — so every f, every COBYLA decision, every evaluation
    count and the result are the reference's bit for bit"""
    r = run_gn_mlsl(T.more_bind(O.ref()), alg, obj, n, maxeval)
    a = run_gn_mlsl(T.more_bind(C.CDLL(nlopt_amd.LIB_PATH)), alg, obj, n, maxeval, params=[("amd_exact_dot", 1), ("amd_cobyla_min_batch", 1)], stats=True)
    assert (a["ret"], a["minf"], a["nevals"]) == (r["ret"], r["minf"], r["nevals"]) and np.array_equal(a["x"], r["x"]), (a, r)
    assert a["launches"] > 0 and a["searches"] > 0 and a["host_searches"] == 0         # every search ran in a batched launch on the device


@pytest.mark.parametrize("alg,obj,n,maxeval,pop", [(T.GN_MLSL_LDS, "rosenbrock", 4, 30000, 400), (T.GN_MLSL, "sphere", 12, 40000, 300)])
def test_gn_mlsl_cobyla_batches_go_where_they_run_faster_and_the_run_stays_the_references(alg, obj, n, maxeval, pop):
    """the routing: a batch of fewer searches than the device needs to beat a host core runs through the host algorithm, a large one as
    one launch (lbfgs_driver.c cob_min_batch: 24 / 12 / 6 searches by dimension; 3 here, so that the first iteration's single search
    goes one way and the second iteration's batch the other).  In exact-order mode both produce the reference's searches bit for bit,
    so the whole run is the reference's whichever way each batch went"""
    r = run_gn_mlsl(T.more_bind(O.ref()), alg, obj, n, maxeval, population=pop)
    a = run_gn_mlsl(T.more_bind(C.CDLL(nlopt_amd.LIB_PATH)), alg, obj, n, maxeval, params=[("amd_exact_dot", 1), ("amd_cobyla_min_batch", 3)], stats=True, population=pop)
    assert (a["ret"], a["minf"], a["nevals"]) == (r["ret"], r["minf"], r["nevals"]) and np.array_equal(a["x"], r["x"]), (a, r)
    assert a["launches"] > 0 and a["host_searches"] > 0, a
    d = run_gn_mlsl(T.more_bind(C.CDLL(nlopt_amd.LIB_PATH)), alg, obj, n, maxeval, params=[("amd_exact_dot", 1)], stats=True, population=pop)      # the default threshold
    assert (d["ret"], d["minf"], d["nevals"]) == (r["ret"], r["minf"], r["nevals"]) and np.array_equal(d["x"], r["x"]), (d, r)


@pytest.mark.parametrize("alg,obj,n,maxeval,tol", [(T.GN_MLSL, "sphere", 6, 2500, 1e-6), (T.GN_MLSL_LDS, "sphere", 20, 6000, 1e-6), (T.GN_MLSL_LDS, "rastrigin", 3, 3000, None),
                                                   (T.GN_MLSL, "ackley", 2, 2000, None), (T.GN_MLSL_LDS, "griewank", 5, 4000, None), (T.GN_MLSL, "levy", 8, 6000, None)])
def test_gn_mlsl_batched_device_cobyla_default_mode(alg, obj, n, maxeval, tol):
    """the default (tree-sum) objective: f differs from the host twin's by rounding (and by device libm's last bit), so a COBYLA search
    takes slightly different steps.  On the unimodal sphere every search still ends at the minimum: the reference's result code and
    minimum (1e-6 absolute).  On the multimodal objectives of the zoo rounding-level differences send single searches into neighbouring
    local minima (measured on the MI355X: Griewank n = 5 0.537 against the reference's 0.621, Levy n = 8 9.72 against 8.06): there the
    result code, batched launches and a minimum of the reference's size are asserted — the PARITY mode is amd_exact_dot = 1, above,
    where the same runs are the reference's bit for bit"""
    r = run_gn_mlsl(T.more_bind(O.ref()), alg, obj, n, maxeval)
    a = run_gn_mlsl(T.more_bind(C.CDLL(nlopt_amd.LIB_PATH)), alg, obj, n, maxeval, params=[("amd_cobyla_min_batch", 1)], stats=True)
    assert a["ret"] == r["ret"], (a, r)
    if tol is None:
        assert 0.0 <= a["minf"] <= 2.0 * max(r["minf"], 1.0), (a["minf"], r["minf"])
    else:
        assert abs(a["minf"] - r["minf"]) <= tol * max(1.0, abs(r["minf"])), (a["minf"], r["minf"])
    assert a["launches"] > 0
