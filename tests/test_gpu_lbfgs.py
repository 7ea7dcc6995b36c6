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


# ---- the two batch kernels against each other, at the kernel-level C-ABI -------------------------------------------------------
class _Params(C.Structure):
    _fields_ = [("minf_max", C.c_double), ("ftol_rel", C.c_double), ("ftol_abs", C.c_double), ("xtol_rel", C.c_double), ("tolg", C.c_double),
                ("maxeval", C.c_int32), ("exact", C.c_int32), ("sign", C.c_double), ("xtol_abs", C.c_void_p), ("x_weights", C.c_void_p),
                ("abort", C.c_void_p), ("ftrace", C.c_void_p), ("ftrace_cap", C.c_int64), ("done", C.c_void_p)]


class _Result(C.Structure):
    _fields_ = [("f", C.c_double), ("ret", C.c_int32), ("nevals", C.c_int32), ("iterm", C.c_int32), ("cols", C.c_int32)]


def _batch(streaming, obj, n, starts, lov, hiv, mf, ftol_rel=1e-8, maxeval=0, sign=1.0, xtol_abs=None, weights=None, cap=1500, exact=False):
    """nla_k_lbfgs_batch on `starts` (count x n): params.exact = 2 / 3 keeps the tree sums / the reference-order sums on the streaming
    kernel (lbfgs_kernels.hip), 0 / 1 takes the resident kernel (lbfgs_resident.hip) where it applies"""
    L = nlopt_amd.lib()
    D = nlopt_amd.DevBuf
    count, ld = starts.shape[0], (n + 1) & ~1
    X = np.zeros((count, ld)); X[:, :n] = starts
    dX, dlb, dub = D.from_array(X), D.from_array(lov), D.from_array(hiv)
    dwork = D.from_array(np.zeros(count * (4 * ld + 2 * mf)))
    diw, dhist = D.from_array(np.zeros(count * ld, dtype=np.int32)), D.from_array(np.full(count * 2 * mf * ld, np.nan))
    dft, dres = D.from_array(np.full(count * cap, np.nan)), D(C.sizeof(_Result) * count)
    dta = D.from_array(xtol_abs) if xtol_abs is not None else None
    dw = D.from_array(weights) if weights is not None else None
    P = _Params(-np.inf, ftol_rel, 0.0, 0.0, 0.0, maxeval, (3 if streaming else 1) if exact else (2 if streaming else 0), sign,
                dta.ptr if dta else None, dw.ptr if dw else None, None, dft.ptr, cap)
    vp = C.c_void_p
    L.nla_k_lbfgs_batch.argtypes = [C.c_int] * 5 + [vp] * 6 + [C.POINTER(_Params), vp, vp, vp]
    assert L.nla_k_lbfgs_batch(nlopt_amd.OBJECTIVES[obj], n, ld, mf, count, dlb.ptr, dub.ptr, dX.ptr, dwork.ptr, diw.ptr, dhist.ptr, C.byref(P),
                               dres.ptr, None, None) == 0 and L.nla_stream_sync(None) == 0
    res = np.frombuffer(dres.to_array(np.uint8, C.sizeof(_Result) * count).tobytes(), dtype=[("f", "f8"), ("ret", "i4"), ("nevals", "i4"), ("iterm", "i4"), ("cols", "i4")])
    out = dict(x=dX.to_array(np.float64, count * ld).reshape(count, ld)[:, :n].copy(), res=res.copy(), ftrace=dft.to_array(np.float64, count * cap).reshape(count, cap))
    for b in (dX, dlb, dub, dwork, diw, dhist, dft, dres, dta, dw):
        if b is not None:
            b.free()
    return out


@pytest.mark.parametrize("exact", [False, True], ids=["tree_sums", "reference_order"])
@pytest.mark.parametrize("obj,n,count,mf,kw", [
    ("ackley", 4096, 6, 320, {}),                            # the config-4 shape
    ("ackley", 300, 3, 400, {}), ("rastrigin", 40, 2, 5, {}), ("rosenbrock", 10, 2, 400, {}), ("griewank", 257, 2, 3, {}),
    ("levy", 33, 2, 400, {}), ("sphere", 5, 2, 400, {}), ("rastrigin", 513, 2, 4, {}), ("rastrigin", 4095, 2, 7, {}),
    ("ackley", 600, 2, 400, dict(maxeval=9)), ("griewank", 64, 3, 50, dict(sign=-1.0, maxeval=40)),
    ("rastrigin", 100, 2, 30, dict(weights=True)), ("ackley", 77, 2, 30, dict(xtol_abs=True)),
    # round 6: 4096 < n <= 8192 on the 32-coordinates-per-thread build (hip/lbfgs_resident32.hip: x and g 2 x 64 KB of LDS, one workgroup per CU)
    ("ackley", 8192, 3, 160, dict(maxeval=60)), ("rastrigin", 4097, 2, 6, dict(maxeval=80)), ("griewank", 6000, 2, 12, dict(maxeval=50)),
    ("rosenbrock", 5000, 2, 9, dict(maxeval=70)), ("levy", 7777, 2, 5, dict(maxeval=40, sign=-1.0)), ("sphere", 8191, 2, 3, dict(weights=True)),
])
def test_resident_kernel_is_the_streaming_kernel(obj, n, count, mf, kw, exact):
    """lbfgs_resident_kernel (x / g in LDS, direction in registers, the scalar state advanced by thread 0) must be
    lbfgs_batch_kernel's search BIT FOR BIT, in both summation modes (workgroup tree; the reference's sequential order): f of every evaluation, minimiser, result code, evaluation and column counts —
    with coordinates on their bounds from the start, a fixed coordinate, short histories that wrap, maximisation, weights,
    xtol_abs.  (The same comparison runs on the CPU in tools/lbfgs_emu_check.py; the streaming kernel's own parity tests are above.)"""
    rng = np.random.default_rng(n * 7 + count)
    _, lo, hi = O.golden_x0(obj, n)
    lov, hiv = np.full(n, lo), np.full(n, hi)
    starts = rng.uniform(lo, hi, (count, n))
    starts[0, : max(1, n // 7)] = hi
    if n > 8:
        lov[3] = hiv[3] = 0.5 * (lo + hi)
        starts[:, 3] = lov[3]
    kw = dict(kw)
    if kw.pop("weights", False):
        kw["weights"] = rng.uniform(0.5, 2.0, n)
    if kw.pop("xtol_abs", False):
        kw["xtol_abs"] = np.full(n, 1e-3)
    a = _batch(True, obj, n, starts, lov, hiv, mf, exact=exact, **kw)
    b = _batch(False, obj, n, starts, lov, hiv, mf, exact=exact, **kw)
    assert np.array_equal(a["res"], b["res"]), (a["res"], b["res"])
    assert np.array_equal(a["ftrace"], b["ftrace"], equal_nan=True)
    assert np.array_equal(a["x"], b["x"])
    assert (a["res"]["nevals"] > 1).all()


def test_lane_exchange_through_the_valu_is_the_shuffle():
    """dev_common.h nla_xor_lane<M> (DPP quad permutes / row shifts / row rotation, v_permlane16_swap, v_permlane32_swap) against
    __shfl_xor (ds_bpermute) for every distance of the butterfly on every lane of four wavefronts, doubles (incl. NaN payloads,
    infinities, -0.0, denormals) and ints: the same bits — so every reduction of the library sums what it always summed"""
    L = nlopt_amd.lib()
    rng = np.random.default_rng(5)
    v = rng.standard_normal(256) * 10.0 ** rng.integers(-8, 8, 256)
    v[[3, 70, 131, 255]] = [np.inf, -np.inf, -0.0, 5e-324]
    raw = v.view(np.uint64).copy()
    raw[17] = 0x7FF8000000ABCDEF; raw[200] = 0xFFF0000000000001          # NaNs with payloads
    v = raw.view(np.float64)
    dI, dO = nlopt_amd.DevBuf.from_array(v), nlopt_amd.DevBuf(8 * 24 * 256)
    L.nla_k_debug_xor_lane.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.nla_k_debug_xor_lane(dI.ptr, dO.ptr, None) == 0 and L.nla_stream_sync(None) == 0
    o = dO.to_array(np.uint64, 24 * 256).reshape(12, 2, 256)
    lanes = np.arange(256)
    for s, m in enumerate((32, 16, 8, 4, 2, 1)):
        assert np.array_equal(o[s, 0], raw[lanes ^ m]), m                     # the shuffle itself gives the partner's bits
        assert np.array_equal(o[s, 1], o[s, 0]), m                             # ... and so does the VALU route
        assert np.array_equal(o[6 + s, 1], o[6 + s, 0]), m                     # ints


def test_device_sincos_is_sin_and_cos():
    """lbfgs_resident.hip computes Ackley's / Rastrigin's cos(2 pi x) (for f) and sin(2 pi x) (for the gradient) with ONE sincos call;
    the streaming kernel and the population kernels call cos and sin.  The device library must return the same bits either way:
    1e6 arguments over the objectives' argument range (|2 pi x| <= 206) and beyond, plus the awkward ones."""
    L = nlopt_amd.lib()
    rng = np.random.default_rng(12)
    a = np.concatenate([rng.uniform(-206.0, 206.0, 600000), rng.uniform(-4e3, 4e3, 200000), rng.uniform(-1e-3, 1e-3, 100000), rng.uniform(-1e9, 1e9, 100000),
                        6.283185307179586 * rng.uniform(-32.768, 32.768, 100000),
                        np.array([0.0, -0.0, np.pi, -np.pi, np.pi / 2, 1e-300, 1e22, np.inf, -np.inf, np.nan, 0.5939350286162490 * 6.283185307179586])])
    dA, dO = nlopt_amd.DevBuf.from_array(a), nlopt_amd.DevBuf(8 * 4 * len(a))
    L.nla_k_debug_sincos.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.nla_k_debug_sincos(len(a), dA.ptr, dO.ptr, None) == 0 and L.nla_stream_sync(None) == 0
    o = dO.to_array(np.float64, 4 * len(a)).reshape(-1, 4)
    assert np.array_equal(o[:, 0], o[:, 2], equal_nan=True) and np.array_equal(o[:, 1], o[:, 3], equal_nan=True)
    assert np.isfinite(o[:-4]).all() and o[len(a) - 11, 1] == 1.0 and o[len(a) - 11, 0] == 0.0          # the row of a = 0.0
    dA.free(); dO.free()
