"""-m gpu: MLSL (G_MLSL / G_MLSL_LDS with an LD_LBFGS local optimiser) end to end through the public
C API against the CPU oracle (oracle/port_mlsl.c + port_lbfgs.c, pinned bit-exactly to the real
reference).  Bar: the same sample sequence (f of every sample within 1e-10 relative — same stream
offsets, bit-identical x — and the same number of samples drawn), the same result code, the same
global minimum within 1e-7.  Local searches: the device sums dot products in a fixed tree order, so
a search follows the reference's trajectory to rounding; on smooth basins that means the same
minimum to ~1e-8 and an evaluation count within one or two, on rugged objectives (Griewank's
cosine product) rounding-level differences can steer a long search into a neighbouring basin, which
changes which later candidates qualify — the test bounds that drift instead of forbidding it."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = pytest.mark.gpu


def run_amd(obj, n, nsamples, seed, alg=None, maxeval=0, stopval=None, local_ftol_rel=1e-8, local_xtol_rel=0.0, local_ftol_abs=0.0,
            local_maxeval=0, mf=0):
    assert nlopt_amd.device_count() > 0
    L = nlopt_amd.lib()
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(alg or nlopt_amd.G_MLSL, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective(obj))
    loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
    if local_ftol_rel:
        loc.set_ftol_rel(local_ftol_rel)
    if local_ftol_abs:
        loc.set_ftol_abs(local_ftol_abs)
    if local_xtol_rel:
        loc.set_xtol_rel(local_xtol_rel)
    if local_maxeval:
        loc.set_maxeval(local_maxeval)
    if mf:
        L.nlopt_set_vector_storage(loc._h, mf)
    assert L.nlopt_set_local_optimizer(o._h, loc._h) > 0
    if nsamples:
        o.set_population(nsamples)
    if maxeval:
        o.set_maxeval(maxeval)
    if stopval is not None:
        o.set_stopval(stopval)
    o.enable_trace((maxeval or 200000) + 4096)
    nlopt_amd.srand(seed)
    x, minf, ret = o.optimize_raw(xs)
    return dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), stats=o.stats(), err=o.get_errmsg(), trace=o.trace())


@pytest.mark.parametrize("obj,n,ns,seed,kw,rugged", [
    ("sphere", 5, 6, 2, dict(stopval=1e-9, maxeval=5000), False),
    ("rastrigin", 4, 10, 42, dict(stopval=1e-6, maxeval=100000), False),
    ("ackley", 6, 25, 7, dict(stopval=1e-5, maxeval=20000), True),
    ("griewank", 5, 0, 3, dict(stopval=1e-7, maxeval=100000), True),
    ("rastrigin", 8, 40, 5, dict(stopval=1.5, maxeval=100000), True),
    ("rosenbrock", 4, 12, 9, dict(stopval=1e-10, maxeval=100000, mf=3), False),
])
def test_mlsl_reaches_the_oracles_result(obj, n, ns, seed, kw, rugged):
    """runs that end on a stop value: same result code, same global minimum, same stream consumption
    (= same number of samples), evaluation count within the local searches' rounding-level slack"""
    a = run_amd(obj, n, ns, seed, **kw)
    p = O.run_port_mlsl(obj, n, ns, seed, **kw)
    assert a["ret"] == p["ret"], (a, p["ret"])
    # every sampling phase but the first was computed beside the local phase before it (mlsl_driver.c, mlsl_enqueue_ahead)
    its = a["stats"]["generations"]
    assert max(its - 1, 0) <= a["stats"]["mlsl_sampled_ahead"] <= its, a["stats"]
    assert a["stats"]["mlsl_gate_timeouts"] == 0, a["stats"]        # (every gate opened: the searches' stream ran beside the generator's)
    nloc = len(p["floc"])
    assert abs(a["minf"] - p["minf"]) <= 1e-7 * max(abs(p["minf"]), 1.0)
    assert np.allclose(a["x"], p["x"], rtol=1e-5, atol=1e-6 * max(np.abs(p["x"]).max(), 1.0))
    if p["ret"] == nlopt_amd.MAXEVAL_REACHED:
        # the evaluation budget ran out: the device's local searches may each differ from the oracle's by an
        # evaluation or two (different summation order), which moves the cut-off by a few samples
        slack = 3 * nloc + 2
        assert abs(a["stats"]["mt_words"] - p["words"]) <= 2 * n * slack
        assert abs(a["stats"]["accepted"] - nloc) <= max(2, nloc // 50)
    else:
        assert a["stats"]["mt_words"] == p["words"]                  # same number of samples drawn
        fs = a["trace"][a["trace"]["kind"] == 3]["f"]
        assert len(fs) == len(p["fsamp"])
        assert np.all(np.abs(fs - p["fsamp"]) <= 1e-10 * np.maximum(np.abs(p["fsamp"]), 1.0))   # the same samples, in order
        fl = a["trace"][a["trace"]["kind"] == 4]
        if rugged:
            assert abs(len(fl) - nloc) <= max(2, nloc // 10)
        else:
            assert len(fl) == nloc                                       # the same local searches, in order, to the same minima
            assert np.all(np.abs(fl["f"] - p["floc"]) <= 1e-7 * np.maximum(np.abs(p["floc"]), 1.0))
            assert np.all(np.abs(fl["accepted"] - p["eloc"]) <= 3)


def test_mlsl_lds_variant_large_n_is_pseudo_random_like_the_reference():
    """n > 1111: the reference's Sobol generator does not exist, G_MLSL_LDS samples pseudo-randomly
    (SURVEY.md fact 7).  Same run as the oracle's non-LDS port; n = 1200 also exercises 64 concurrent
    searches with 1092 history pairs each."""
    kw = dict(maxeval=2600, local_ftol_rel=1e-6)
    a = run_amd("ackley", 1200, 30, 4, alg=nlopt_amd.G_MLSL_LDS, **kw)
    p = O.run_port_mlsl("ackley", 1200, 30, 4, **kw)
    assert a["ret"] == p["ret"] == nlopt_amd.MAXEVAL_REACHED
    assert abs(a["minf"] - p["minf"]) <= 1e-6 * max(abs(p["minf"]), 1.0)


def test_mlsl_argument_errors():
    o = nlopt_amd.Opt(nlopt_amd.G_MLSL, 3)
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    o.set_min_objective(nlopt_amd.objective("sphere"))
    x, minf, ret = o.optimize_raw(np.zeros(3))
    assert ret == nlopt_amd.INVALID_ARGS and "local optimizer must be specified" in o.get_errmsg()
    o2 = nlopt_amd.Opt(nlopt_amd.G_MLSL_LDS, 3)
    o2.set_lower_bounds(-1.0)
    o2.set_upper_bounds(1.0)
    o2.set_min_objective(nlopt_amd.objective("sphere"))
    loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, 3)
    nlopt_amd.lib().nlopt_set_local_optimizer(o2._h, loc._h)
    o2.set_maxeval(200)
    x, minf, ret = o2.optimize_raw(np.zeros(3))
    assert ret == nlopt_amd.MAXEVAL_REACHED          # Sobol sampling (n <= 1111) is served


def test_sobol_rows_kernel_is_bit_identical_to_the_stateful_generator():
    L = nlopt_amd.lib()
    L.nla_sobol_directions.argtypes = [C.c_uint, C.c_void_p]
    L.nla_k_mlsl_sobol_rows.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.nla_sobol_skip_count.argtypes = [C.c_uint]
    L.nla_sobol_skip_count.restype = C.c_uint32
    for n, count, skip_n in ((3, 100, 0), (37, 300, 10 * 37 + 50), (1111, 64, 11114)):
        ld = (n + 1) & ~1
        V = np.zeros(32 * n, dtype=np.uint32)
        assert L.nla_sobol_directions(n, V.ctypes.data) == 1
        lb, ub = np.linspace(-5, -1, n), np.linspace(0.5, 9, n)
        dV, dlb, dub = nlopt_amd.DevBuf.from_array(V), nlopt_amd.DevBuf.from_array(lb), nlopt_amd.DevBuf.from_array(ub)
        dP = nlopt_amd.DevBuf(8 * ld * count)
        first = (L.nla_sobol_skip_count(skip_n) if skip_n else 0) + 1
        assert L.nla_k_mlsl_sobol_rows(n, ld, dlb.ptr, dub.ptr, dV.ptr, first, count, dP.ptr, None) == 0
        assert L.nla_stream_sync(None) == 0
        got = dP.to_array(np.float64, ld * count).reshape(count, ld)[:, :n]
        assert np.array_equal(got, O.port_sobol_points(n, skip_n, count, lb, ub))


@pytest.mark.parametrize("obj,n,ns,seed,kw", [
    ("sphere", 5, 6, 2, dict(stopval=1e-9, maxeval=5000)),
    ("rastrigin", 4, 10, 42, dict(stopval=1e-6, maxeval=100000)),
    ("rosenbrock", 4, 12, 9, dict(stopval=1e-10, maxeval=100000, mf=3)),
    ("ackley", 30, 50, 1, dict(maxeval=4000)),
])
def test_mlsl_lds_sobol_sampling_matches_the_oracle(obj, n, ns, seed, kw):
    """G_MLSL_LDS with a live Sobol generator (n <= 1111): the same samples in order (bit-identical x), no MT word drawn"""
    a = run_amd(obj, n, ns, seed, alg=nlopt_amd.G_MLSL_LDS, **kw)
    p = O.run_port_mlsl(obj, n, ns, seed, lds=True, **kw)
    assert a["ret"] == p["ret"], (a["err"], p["ret"])
    assert a["stats"]["mt_words"] == 0 and p["words"] == 0
    assert max(a["stats"]["generations"] - 1, 0) <= a["stats"]["mlsl_sampled_ahead"] <= a["stats"]["generations"], a["stats"]   # (Sobol rows ahead too)
    assert abs(a["minf"] - p["minf"]) <= 1e-7 * max(abs(p["minf"]), 1.0)
    fs = a["trace"][a["trace"]["kind"] == 3]["f"]
    if p["ret"] != nlopt_amd.MAXEVAL_REACHED:
        assert len(fs) == len(p["fsamp"])
    m = min(len(fs), len(p["fsamp"]), ns or 4)          # the first iteration's samples do not depend on any local search
    assert np.all(np.abs(fs[:m] - p["fsamp"][:m]) <= 1e-10 * np.maximum(np.abs(p["fsamp"][:m]), 1.0))


def _dist2_reference(A, B):
    """mlsl.c's distance2 for every pair: one accumulator per pair over k ascending, (a - b) * (a - b) then add, nothing fused"""
    d = np.zeros((A.shape[0], B.shape[0]))
    for k in range(A.shape[1]):
        dx = A[:, k][:, None] - B[:, k][None, :]
        d += dx * dx
    return d


@pytest.mark.parametrize("n,na,nb", [(4096, 70, 130), (257, 64, 64), (5, 1, 3), (33, 129, 65), (512, 200, 1000), (1, 1, 1), (64, 63, 193),
                                     (31, 33, 129), (34, 9, 257), (2, 40, 128), (63, 8, 127), (96, 31, 1),
                                     (33, 1025, 2049), (40, 1100, 2000)])      # (more than 400 tiles of 64 x 64: the large-tile instance)
def test_pair_distance_kernel_is_the_sequential_sum(n, na, nb):
    """mlsl_dist2_kernel (hip/mlsl_kernels.hip): the squared distance of every (new point, point) pair, bit for bit the serial sum of
    mlsl.c:119-125 — what the closest-point tests (`cpd <= R*R`, mlsl.c:208-209) are decided on.  Ragged tile edges, n not a multiple
    of the coordinate tile (round 4: ran on the device, the register-tiled kernel became the only one; round 5: both tile sizes — the
    32 x 32 instance serves calls with few pairs — odd n, a single coordinate, and the rows' padding poisoned with NaN: it must not be
    read.  A variant with the rows of A as scalar operands was measured slower in round 5 and is not in the tree)."""
    L = nlopt_amd.lib()
    L.nla_k_mlsl_dist2.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(n * 1000 + na)
    ld = (n + 1) & ~1
    A = np.full((na, ld), np.nan); B = np.full((nb, ld), np.nan)
    A[:, :n] = rng.uniform(-32.768, 32.768, (na, n)); B[:, :n] = rng.uniform(-32.768, 32.768, (nb, n))
    B[0, :n] = A[0, :n]                                   # a zero distance
    dA, dB, dD = nlopt_amd.DevBuf.from_array(A), nlopt_amd.DevBuf.from_array(B), nlopt_amd.DevBuf(8 * na * nb)
    assert L.nla_k_mlsl_dist2(n, ld, dA.ptr, na, dB.ptr, nb, dD.ptr, None) == 0 and L.nla_stream_sync(None) == 0
    got = dD.to_array(np.float64, na * nb).reshape(na, nb)
    assert np.array_equal(got, _dist2_reference(A[:, :n], B[:, :n])) and got[0, 0] == 0.0
