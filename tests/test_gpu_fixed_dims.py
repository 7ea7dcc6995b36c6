"""-m gpu: coordinates with lb[i] == ub[i].  The reference eliminates them in front of CRS2_LM / ISRES / ESCH
(elimdim, optimize.c:219-445): the algorithm runs in the reduced dimension (its default population, its RNG consumption).
Same client setup against the REAL reference and against libnlopt_amd: same result, evaluation count, minimum, argmin
(bit for bit for CRS2_LM; ISRES / ESCH go through exp / tan: to rounding)."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built")]


def bounds_with_fixed(obj, n, fixed):
    xs, lo, hi = O.golden_x0(obj, n)
    lb, ub, x0 = np.full(n, lo), np.full(n, hi), np.array(xs)
    for i in fixed:
        lb[i] = ub[i] = x0[i]
    return lb, ub, x0


def run_ref(alg, obj, n, fixed, pop, seed, maxeval):
    lb, ub, x0 = bounds_with_fixed(obj, n, fixed)

    def setup(R, opt):
        R.nlopt_set_lower_bounds(opt, O.dptr(lb))
        R.nlopt_set_upper_bounds(opt, O.dptr(ub))
    return O.run_ref(alg, obj, n, pop, seed, maxeval=maxeval, x0=x0, setup=setup)


def run_amd(alg, obj, n, fixed, pop, seed, maxeval):
    lb, ub, x0 = bounds_with_fixed(obj, n, fixed)
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(lb)
    o.set_upper_bounds(ub)
    o.set_min_objective(nlopt_amd.objective(obj))
    if pop:
        o.set_population(pop)
    o.set_maxeval(maxeval)
    o.enable_trace(maxeval + 64)
    nlopt_amd.srand(seed)
    x, minf, ret = o.optimize_raw(x0)
    return dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), f=o.trace()["f"], err=o.get_errmsg())


@pytest.mark.parametrize("fixed,pop", [([0], 0), ([2, 5, 6], 40), ([0, 1, 2, 3, 4, 5, 7], 0)])
def test_crs_with_fixed_coordinates(fixed, pop):
    a = run_amd(nlopt_amd.GN_CRS2_LM, "rastrigin", 8, fixed, pop, 42, 1500)
    r = run_ref(19, "rastrigin", 8, fixed, pop, 42, 1500)
    assert a["ret"] == r["ret"] and a["nevals"] == r["nevals"], a["err"]
    assert np.array_equal(a["f"], r["fseq"]) and a["minf"] == r["minf"] and np.array_equal(a["x"], r["x"])
    lb, ub, x0 = bounds_with_fixed("rastrigin", 8, fixed)
    assert all(a["x"][i] == x0[i] for i in fixed)


@pytest.mark.parametrize("alg,refalg", [(nlopt_amd.GN_ISRES, 35), (nlopt_amd.GN_ESCH, 42)])
def test_isres_esch_with_fixed_coordinates(alg, refalg):
    a = run_amd(alg, "griewank", 7, [1, 4], 30, 5, 900)
    r = run_ref(refalg, "griewank", 7, [1, 4], 30, 5, 900)
    assert a["ret"] == r["ret"] and a["nevals"] == r["nevals"], a["err"]
    assert np.allclose(a["f"], r["fseq"], rtol=1e-9, atol=0) and abs(a["minf"] - r["minf"]) <= 1e-9 * abs(r["minf"])
    assert np.allclose(a["x"], r["x"], rtol=1e-9, atol=1e-9)


def test_all_coordinates_fixed():
    """n0 = 0 after elimination: the objective is evaluated once at the only point (optimize.c:536-539)"""
    a = run_amd(nlopt_amd.GN_CRS2_LM, "sphere", 3, [0, 1, 2], 0, 1, 100)
    r = run_ref(19, "sphere", 3, [0, 1, 2], 0, 1, 100)
    assert a["ret"] == r["ret"] == 1 and a["minf"] == r["minf"] and np.array_equal(a["x"], r["x"])
