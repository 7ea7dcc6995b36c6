"""Pin the CPU oracle (oracle/port_*.c) — the checker every GPU parity test relies on:
(1) MT19937 against the published known-answer values and, word for word, against the real
reference's samplers; (2) CRS2_LM evaluation-by-evaluation against the real reference
(oracle/_ref/libnlopt_ref.so, when present); (3) against the committed golden vectors that were
generated from the real reference (tests/golden/make_golden.py)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as O

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "crs_golden.json")))
need_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")


def test_mt19937_known_answers():
    L = O.port()
    L.orc_srand(5489)
    first = [L.orc_genrand_int32() for _ in range(5)]
    # published MT19937 outputs for the default seed 5489
    assert first == [3499211612, 581869302, 3890346734, 3586334585, 545404204]
    for _ in range(10000 - 5 - 1):
        L.orc_genrand_int32()
    assert L.orc_genrand_int32() == 4123659995      # the 10000th output (C++11 [rand.predef] check value)


@need_ref
def test_samplers_match_reference_word_for_word():
    L, R = O.port(), O.ref()
    for seed in (0, 1, 42, 2**31 + 5):
        L.orc_srand(seed)
        R.nlopt_srand(seed)
        for i in range(3000):
            k = i % 4
            if k == 0:
                assert L.orc_urand(-3.5, 7.25) == R.nlopt_urand(-3.5, 7.25)
            elif k == 1:
                assert L.orc_iurand(977) == R.nlopt_iurand(977)
            elif k == 2:
                assert L.orc_nrand(1.5, 2.0) == R.nlopt_nrand(1.5, 2.0)
            else:
                assert L.orc_urand(0.0, 1.0) == R.nlopt_urand(0.0, 1.0)


def _fhash(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(GOLD))
def test_port_crs_matches_golden(name):
    g = GOLD[name]
    r = O.run_port_crs(g["obj"], g["n"], g["pop"], g["seed"], record=True, **g["kwargs"])
    assert r["ret"] == g["ret"]
    assert r["nevals"] == g["nevals"]
    assert float(r["minf"]).hex() == g["minf"]
    assert [float(v).hex() for v in r["x"]] == g["x"]
    assert _fhash(r["fseq"]) == g["fseq_sha256"]
    assert hashlib.sha256(r["xhash"].tobytes()).hexdigest() == g["xhash_sha256"]


@need_ref
@pytest.mark.parametrize("obj,n,pop,seed,kw", [
    ("rastrigin", 32, 400, 11, dict(maxeval=5000)),
    ("ackley", 12, 0, 99, dict(maxeval=4000)),
    ("levy", 7, 90, 5, dict(ftol_abs=1e-6, maxeval=30000)),
    ("rosenbrock", 2, 3, 1, dict(maxeval=500)),          # N == n+1: every trial uses every other row
])
def test_port_crs_matches_reference_live(obj, n, pop, seed, kw):
    a = O.run_port_crs(obj, n, pop, seed, record=True, **kw)
    b = O.run_ref(19, obj, n, pop, seed, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert a["minf"] == b["minf"]
    assert np.array_equal(a["x"], b["x"])
    assert np.array_equal(a["fseq"], b["fseq"])
    assert np.array_equal(a["xhash"], b["xhash"])


def test_port_crs_rejects_small_population():
    r = O.run_port_crs("sphere", 5, 5, 1, maxeval=100)     # N < n+1 (crs.c:180-184)
    assert r["ret"] == -2


def test_word_accounting_matches_survey_appendix_a():
    # CRS2_LM n=8, N=50, 5002 evals consumes 2n(N-1) + 2n(evals-N) = 80016 words
    r = O.run_port_crs("griewank", 8, 50, 42, maxeval=5002)
    assert r["nevals"] == 5002 and r["words"] == 80016


# ---- ISRES ------------------------------------------------------------------------------------
@need_ref
@pytest.mark.parametrize("obj,n,pop,seed,nineq,neq,kw", [
    ("rastrigin", 8, 40, 42, 0, 0, dict(maxeval=2000)),          # unconstrained: qsort ranking path
    ("rastrigin", 12, 60, 42, 4, 0, dict(maxeval=3000)),         # stochastic ranking, 4 inequality constraints
    ("griewank", 6, 0, 7, 2, 1, dict(maxeval=2500)),             # default population, inequality + equality
    ("sphere", 5, 30, 3, 1, 0, dict(ftol_abs=1e-9, maxeval=20000)),
    ("rosenbrock", 4, 35, 11, 0, 0, dict(xtol_rel=1e-4, maxeval=20000)),
    ("levy", 3, 7, 5, 0, 2, dict(maxeval=700)),                  # pop=7: survivors = 1 (last-survivor mutation only)
    ("sphere", 4, 50, 9, 2, 0, dict(stopval=0.05, maxeval=20000)),
])
def test_port_isres_matches_reference_live(obj, n, pop, seed, nineq, neq, kw):
    a = O.run_port_isres(obj, n, pop, seed, nineq, neq, **kw)
    b = O.run_ref_isres(obj, n, pop, seed, nineq, neq, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert a["minf"] == b["minf"]
    assert np.array_equal(a["x"], b["x"])
    assert np.array_equal(a["fseq"], b["fseq"])
    assert np.array_equal(a["xhash"], b["xhash"])


IGOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "isres_golden.json")))


@pytest.mark.parametrize("name", sorted(IGOLD))
def test_port_isres_matches_golden(name):
    g = IGOLD[name]
    r = O.run_port_isres(g["obj"], g["n"], g["pop"], g["seed"], g["nineq"], g["neq"], **g["kwargs"])
    assert r["ret"] == g["ret"] and r["nevals"] == g["nevals"]
    assert float(r["minf"]).hex() == g["minf"]
    assert [float(v).hex() for v in r["x"]] == g["x"]
    assert _fhash(r["fseq"]) == g["fseq_sha256"]
    assert hashlib.sha256(r["xhash"].tobytes()).hexdigest() == g["xhash_sha256"]


# ---- LD_LBFGS ---------------------------------------------------------------------------------
@need_ref
@pytest.mark.parametrize("obj,n,kw", [
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
])
def test_port_lbfgs_matches_reference_live(obj, n, kw):
    a = O.run_port_lbfgs(obj, n, **kw)
    b = O.run_ref_lbfgs(obj, n, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])


@need_ref
def test_port_lbfgs_active_bounds_match_reference():
    """start points and boxes that put coordinates on their bounds: projection, active set, release"""
    rng = np.random.default_rng(3)
    for n, obj in ((9, "sphere"), (14, "rastrigin"), (6, "rosenbrock")):
        lb = -rng.uniform(0.1, 2.0, n)
        ub = rng.uniform(0.1, 2.0, n)
        lb[::3] = 0.3          # the unconstrained minimiser (0 or 1) lies outside on these coordinates
        ub[::3] = 2.5
        x0 = np.clip(rng.uniform(-2, 2, n), lb, ub)
        x0[1] = ub[1]
        a = O.run_port_lbfgs(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        b = O.run_ref_lbfgs(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
        assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
        assert np.array_equal(a["x"], b["x"])


# ---- MLSL + LD_LBFGS ----------------------------------------------------------------------------
@need_ref
@pytest.mark.parametrize("obj,n,ns,seed,kw", [
    ("rastrigin", 4, 10, 42, dict(maxeval=3000)),
    ("ackley", 6, 25, 7, dict(maxeval=5000)),
    ("griewank", 5, 0, 3, dict(maxeval=2500)),                     # default 4 samples per iteration
    ("levy", 3, 8, 11, dict(maxeval=2000, local_xtol_rel=1e-6, local_ftol_rel=0.0)),
    ("rastrigin", 8, 40, 5, dict(maxeval=6000, local_maxeval=30)),
    ("sphere", 5, 6, 2, dict(stopval=1e-9, maxeval=5000)),
    ("rosenbrock", 4, 12, 9, dict(maxeval=4000, mf=3)),
])
def test_port_mlsl_matches_reference_live(obj, n, ns, seed, kw):
    a = O.run_port_mlsl(obj, n, ns, seed, **kw)
    b = O.run_ref_mlsl(obj, n, ns, seed, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])


# ---- LD_MMA without nonlinear constraints, and MLSL with it (GD_MLSL's default local optimiser) ----
@need_ref
@pytest.mark.parametrize("obj,n,kw", [
    ("sphere", 8, dict(ftol_rel=1e-10)),
    ("rosenbrock", 10, dict(maxeval=500)),
    ("rosenbrock", 2, dict(ftol_rel=1e-10)),
    ("ackley", 30, dict(ftol_rel=1e-8)),
    ("rastrigin", 20, dict(ftol_rel=1e-8)),
    ("griewank", 12, dict(xtol_rel=1e-6)),
    ("levy", 7, dict(ftol_abs=1e-12)),
    ("ackley", 200, dict(ftol_rel=1e-8)),
    ("rastrigin", 64, dict(maxeval=37)),
    ("sphere", 6, dict(stopval=1e-3)),
    ("rastrigin", 16, dict(ftol_rel=1e-9, params=dict(inner_gradients=0))),        # calls != counted evaluations
    ("rosenbrock", 6, dict(maxeval=3000, params=dict(always_improve=0))),
    ("ackley", 10, dict(ftol_rel=1e-9, params=dict(inner_maxeval=2, rho_init=0.01))),
    ("griewank", 10, dict(xtol_rel=1e-8, params=dict(sigma_min=0.5))),
    ("rastrigin", 12, dict(ftol_rel=1e-9, step=0.3)),                                # initial step = sigma_init
    # inner_gradients = 0 together with inner_maxeval: after the re-evaluation with a gradient mma.c:343 recomputes inner_done
    # WITHOUT the inner_maxeval clause — a step that only ended because the inner limit was hit carries on (found by
    # tests/test_api_differential.py; the first version of the port and of the kernel ended the step there)
    ("rastrigin", 8, dict(ftol_rel=1e-9, params=dict(inner_gradients=0, inner_maxeval=2))),
    ("ackley", 6, dict(maxeval=300, params=dict(inner_gradients=0, inner_maxeval=1, always_improve=0))),
    ("griewank", 5, dict(xtol_rel=1e-7, params=dict(inner_gradients=0, inner_maxeval=4, rho_init=0.1))),
    ("levy", 5, dict(maxeval=400, params=dict(inner_gradients=0, inner_maxeval=5, always_improve=0))),
])
def test_port_mma_matches_reference_live(obj, n, kw):
    a = O.run_port_mma(obj, n, **kw)
    b = O.run_ref_mma(obj, n, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])


@need_ref
def test_port_mma_bounds_and_fixed_coordinate_match_reference():
    n = 9
    lb, ub = np.full(n, -2.0), np.full(n, 3.0)
    lb[2] = 0.7
    lb[5] = ub[5] = 1.25                     # sigma = 0 (mma.c:91-94)
    x0 = np.linspace(0.9, 2.6, n)
    x0[5] = 1.25
    for obj in ("sphere", "rastrigin"):
        a = O.run_port_mma(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        b = O.run_ref_mma(obj, n, x0=x0, lb=lb, ub=ub, ftol_rel=1e-10)
        assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
        assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["x"], b["x"])


@need_ref
@pytest.mark.parametrize("alg,local,obj,n,ns,seed,kw", [
    (38, "mma", "rastrigin", 6, 20, 7, dict(maxeval=3000)),
    (38, "mma", "ackley", 10, 30, 7, dict(maxeval=4000)),
    (38, "mma", "levy", 5, 16, 7, dict(maxeval=2000, local_params=dict(inner_gradients=0))),
    (21, None, "rastrigin", 6, 20, 11, dict(maxeval=3000, ftol_rel=1e-7)),              # GD_MLSL, default local optimiser
    (21, None, "sphere", 5, 6, 2, dict(stopval=1e-9, maxeval=5000, ftol_rel=1e-8)),
    (23, None, "ackley", 10, 0, 11, dict(maxeval=3000, ftol_rel=1e-7)),                 # GD_MLSL_LDS, Sobol sampling
])
def test_port_mlsl_with_mma_matches_reference_live(alg, local, obj, n, ns, seed, kw):
    pk = dict(kw)
    tol = pk.pop("ftol_rel", None)
    if tol is not None:
        pk["local_ftol_rel"] = tol           # the dispatcher copies the global tolerances to its default local optimiser
    a = O.run_port_mlsl(obj, n, ns, seed, local="mma", lds=(alg in (23, 39)), **pk)
    b = O.run_ref_mlsl(obj, n, ns, seed, alg=alg, local=local, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])


# ---- Sobol LDS (a19) -----------------------------------------------------------------------------
@need_ref
@pytest.mark.parametrize("sdim,skip_n,count", [(1, 0, 300), (2, 0, 1025), (7, 110, 200), (40, 1000, 300), (1111, 11114, 40)])
def test_port_sobol_matches_reference(sdim, skip_n, count):
    a = O.port_sobol_points(sdim, skip_n, count)
    b = O.ref_sobol_points(sdim, skip_n, count)
    assert np.array_equal(a, b)
    lb, ub = np.linspace(-3, -1, sdim), np.linspace(2, 7, sdim)
    assert np.array_equal(O.port_sobol_points(sdim, skip_n, 50, lb, ub), O.ref_sobol_points(sdim, skip_n, 50, lb, ub))


@need_ref
def test_sobol_has_no_generator_above_1111_dimensions():
    assert O.port_sobol_points(1112, 0, 1) is None and O.ref_sobol_points(1112, 0, 1) is None
    assert O.port_sobol_points(0, 0, 1) is None and O.ref_sobol_points(0, 0, 1) is None


def test_sobol_first_points_known_answers():
    """the classic start of the sequence: 1/2; (3/4, 1/4); (1/4, 3/4); ... in Gray-code order"""
    p = O.port_sobol_points(2, 0, 3)
    assert np.array_equal(p, np.array([[0.5, 0.5], [0.75, 0.25], [0.25, 0.75]]))


@need_ref
@pytest.mark.parametrize("obj,n,ns,seed,kw", [
    ("rastrigin", 4, 10, 42, dict(maxeval=3000)),
    ("ackley", 6, 25, 7, dict(maxeval=4000)),
    ("levy", 3, 0, 11, dict(maxeval=1500)),
    ("griewank", 30, 50, 1, dict(maxeval=6000)),
])
def test_port_mlsl_lds_matches_reference_live(obj, n, ns, seed, kw):
    """G_MLSL_LDS (39) with a live Sobol generator: every evaluation identical, and no MT word drawn"""
    a = O.run_port_mlsl(obj, n, ns, seed, lds=True, **kw)
    b = O.run_ref_mlsl(obj, n, ns, seed, alg=39, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])
    assert a["words"] == 0


# ---- ESCH -----------------------------------------------------------------------
@need_ref
@pytest.mark.parametrize("obj,n,pop,seed,kw", [
    ("rastrigin", 6, 0, 42, dict(maxeval=3000)),                    # default 40 parents / 60 offspring
    ("griewank", 10, 50, 7, dict(maxeval=4000)),
    ("ackley", 3, 7, 3, dict(maxeval=1500)),                        # no = 10, (no n)/10 = 3 mutations per generation
    ("sphere", 1, 5, 5, dict(maxeval=400)),                         # n = 1: (no n)/10 = 0 -> 1 mutation
    ("rosenbrock", 30, 200, 11, dict(maxeval=6000)),
    ("levy", 8, 30, 1, dict(stopval=0.5, maxeval=20000)),
])
def test_port_esch_matches_reference_live(obj, n, pop, seed, kw):
    a = O.run_port_esch(obj, n, pop, seed, **kw)
    b = O.run_ref_esch(obj, n, pop, seed, **kw)
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])


# ---- committed golden vectors for the rest of the path (generated from the real reference by tests/golden/make_golden.py):
# these pin the oracle where oracle/_ref is absent ---------------------------------------------------------------------
RGOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rest_golden.json")))


def _check(r, g):
    assert (r["ret"], r["nevals"]) == (g["ret"], g["nevals"])
    assert float(r["minf"]).hex() == g["minf"] and [float(v).hex() for v in r["x"]] == g["x"]
    assert hashlib.sha256(np.ascontiguousarray(r["fseq"], dtype=np.float64).tobytes()).hexdigest() == g["fseq_sha256"]
    assert hashlib.sha256(r["xhash"].tobytes()).hexdigest() == g["xhash_sha256"]


@pytest.mark.parametrize("name", sorted(RGOLD["lbfgs"]))
def test_port_lbfgs_matches_golden(name):
    g = RGOLD["lbfgs"][name]
    _check(O.run_port_lbfgs(g["obj"], g["n"], **g["kwargs"]), g)


@pytest.mark.parametrize("name", sorted(RGOLD["mlsl"]))
def test_port_mlsl_matches_golden(name):
    g = RGOLD["mlsl"][name]
    _check(O.run_port_mlsl(g["obj"], g["n"], g["ns"], g["seed"], lds=(g["alg"] == 39), **g["kwargs"]), g)


@pytest.mark.parametrize("name", sorted(RGOLD["mma"]))
def test_port_mma_matches_golden(name):
    g = RGOLD["mma"][name]
    _check(O.run_port_mma(g["obj"], g["n"], **g["kwargs"]), g)


@pytest.mark.parametrize("name", sorted(RGOLD["mlsl_mma"]))
def test_port_mlsl_with_mma_matches_golden(name):
    g = RGOLD["mlsl_mma"][name]
    kw = dict(g["kwargs"])
    tol = kw.pop("ftol_rel", None)
    if tol is not None:
        kw["local_ftol_rel"] = tol
    _check(O.run_port_mlsl(g["obj"], g["n"], g["ns"], g["seed"], local="mma", lds=(g["alg"] in (23, 39)), **kw), g)


@pytest.mark.parametrize("name", sorted(RGOLD["esch"]))
def test_port_esch_matches_golden(name):
    g = RGOLD["esch"][name]
    _check(O.run_port_esch(g["obj"], g["n"], g["pop"], g["seed"], **g["kwargs"]), g)


@pytest.mark.parametrize("name", sorted(RGOLD["sobol"]))
def test_port_sobol_matches_golden(name):
    g = RGOLD["sobol"][name]
    pts = O.port_sobol_points(g["sdim"], g["skip_n"], g["count"])
    assert hashlib.sha256(np.ascontiguousarray(pts, dtype=np.float64).tobytes()).hexdigest() == g["sha256"]
    assert [float(v).hex() for v in pts[0][:8]] == g["first"] and [float(v).hex() for v in pts[-1][:8]] == g["last"]


# ---- randomized sweep: every port against the live reference on drawn (seeded) configurations ------------------------------
def _same(a, b):
    assert (a["ret"], a["nevals"]) == (b["ret"], b["nevals"])
    assert np.array_equal(a["fseq"], b["fseq"]) and np.array_equal(a["xhash"], b["xhash"])
    assert a["minf"] == b["minf"] and np.array_equal(a["x"], b["x"])


@need_ref
@pytest.mark.parametrize("scale", [1, 5])
@pytest.mark.parametrize("draw", range(40))
def test_random_configurations_match_the_reference(draw, scale):
    """objective, dimension, population, seed and stopping rule drawn from a fixed generator: CRS2_LM, ISRES (with and without
    constraints), ESCH, LD_LBFGS, LD_MMA and MLSL with either local optimiser — evaluation by evaluation against the real reference"""
    rng = np.random.default_rng(1000 * scale + draw)
    objs = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"]
    obj = objs[int(rng.integers(len(objs)))]
    n = int(rng.integers(2, 12 * scale + 1))
    seed = int(rng.integers(1, 2 ** 31))
    # CRS2_LM: population >= n + 1
    pop = int(rng.integers(n + 1, 6 * n + 8))
    kw = dict(maxeval=int(rng.integers(pop + 50, pop + 900 * scale)))
    if rng.random() < 0.3:
        kw["ftol_rel"] = 10.0 ** -int(rng.integers(2, 7))
    _same(O.run_port_crs(obj, n, pop, seed, record=True, **kw), O.run_ref(19, obj, n, pop, seed, **kw))
    # ISRES
    ipop = int(rng.integers(8, 60))
    nineq, neq = (int(rng.integers(0, 3)), int(rng.integers(0, 2))) if n >= 4 else (0, 0)
    ikw = dict(maxeval=int(rng.integers(2 * ipop, 8 * ipop)))
    _same(O.run_port_isres(obj, n, ipop, seed, nineq=nineq, neq=neq, **ikw), O.run_ref_isres(obj, n, ipop, seed, nineq=nineq, neq=neq, **ikw))
    # ESCH
    epop = int(rng.integers(3, 40))
    ekw = dict(maxeval=int(rng.integers(3 * epop, 30 * epop)))
    _same(O.run_port_esch(obj, n, epop, seed, **ekw), O.run_ref_esch(obj, n, epop, seed, **ekw))
    # local optimisers from the golden start point
    lkw = dict(maxeval=int(rng.integers(20, 400 * scale)))
    if rng.random() < 0.5:
        lkw["ftol_rel"] = 10.0 ** -int(rng.integers(4, 11))
    else:
        lkw["xtol_rel"] = 10.0 ** -int(rng.integers(3, 9))
    mf = int(rng.integers(0, 8))
    _same(O.run_port_lbfgs(obj, n, mf=mf, **lkw), O.run_ref_lbfgs(obj, n, mf=mf, **lkw))
    mp = {}
    if rng.random() < 0.3:
        mp["inner_gradients"] = 0
    if rng.random() < 0.3:
        mp["always_improve"] = 0
    if rng.random() < 0.3:
        mp["rho_init"] = float(10.0 ** rng.uniform(-3, 1))
    if rng.random() < 0.3:
        mp["inner_maxeval"] = int(rng.integers(1, 6))
    _same(O.run_port_mma(obj, n, params=mp, **lkw), O.run_ref_mma(obj, n, params=mp, **lkw))
    # MLSL: pseudo-random or Sobol sampling, LD_LBFGS or LD_MMA
    ns = int(rng.integers(0, 30))
    lds = bool(rng.random() < 0.5)
    local = "mma" if rng.random() < 0.5 else "lbfgs"
    mkw = dict(maxeval=int(rng.integers(300, 2500 * scale)), local_ftol_rel=10.0 ** -int(rng.integers(4, 10)))
    _same(O.run_port_mlsl(obj, n, ns, seed, lds=lds, local=local, **mkw),
          O.run_ref_mlsl(obj, n, ns, seed, alg=39 if lds else 38, local=local, **mkw))
