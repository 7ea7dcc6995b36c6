"""-m gpu: user-supplied device objectives (SURVEY.md §8b "required extension (ii): an additive setter").
tests/userobj/zoo_extra.hip is written against include/nlopt_amd_device.h only and compiled into a code object outside the
library; nlopt_amd_set_min_device_objective binds it.  The objective then runs ON THE DEVICE inside every algorithm — population /
sample evaluation in one launch of the user's kernel, local searches as device coroutines stepped a batch at a time — and must
reproduce what the same function does as an ordinary host callback (the host-callback path is exact against the reference, see
tests/test_gpu_host_callbacks.py): identical candidate indices, x bit for bit where x does not depend on f (CRS2_LM), f within
1e-10.  Also here: the two n-general functions of the reference's zoo that are not compiled in (convexcosh, Shubert —
test/testfuncs.c:288-299,323-339) as such objectives."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "userobj", "zoo_extra.hip")
CO = os.path.join(HERE, "userobj", "zoo_extra.hsaco")
RTOL = 1e-10


@pytest.fixture(scope="module")
def code_object():
    if not os.path.exists(CO) or os.path.getmtime(CO) < os.path.getmtime(SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "--genco", "-I",
                        os.path.join(os.path.dirname(HERE), "include"), SRC, "-o", CO], check=True)
    return CO


# host twins (plain numpy, sequential-order sums)
def convexcosh(x, g):
    i = np.arange(len(x))
    a = (x - i) * (i + 1)
    f = 1.0
    for v in np.cosh(a):
        f *= v
    if g.size:
        g[:] = f * np.tanh(a) * (i + 1)
    return float(f)


def shubert(x, g):
    f = 0.0
    for j in range(1, 6):
        for xi in x:
            f -= j * np.sin((j + 1) * xi + j)
    if g.size:
        g[:] = -sum(j * (j + 1) * np.cos((j + 1) * x + j) for j in range(1, 6))
    return float(f)


def rastrigin(x, g):
    f = 10.0 * len(x)
    for xi in x:
        f += xi * xi - 10.0 * np.cos(6.283185307179586 * xi)
    if g.size:
        g[:] = 2 * x + 10.0 * 6.283185307179586 * np.sin(6.283185307179586 * x)
    return float(f)


TWIN = {"convexcosh": convexcosh, "shubert": shubert, "myrastrigin": rastrigin}
BOX = {"convexcosh": lambda n: (np.full(n, -1.0) + np.arange(n) * 0.5, np.arange(n) * 1.0 + 2.0),        # around the zoo's box (:301-302)
       "shubert": lambda n: (np.full(n, -10.0), np.full(n, 10.0)), "myrastrigin": lambda n: (np.full(n, -5.12), np.full(n, 5.12))}


def make(alg, name, n, co, device=True, host_eval=False):
    o = nlopt_amd.Opt(alg, n)
    lb, ub = BOX[name](n)
    o.set_lower_bounds(lb)
    o.set_upper_bounds(ub)
    if device:
        assert o.set_min_device_objective(co, name, TWIN[name]) > 0, o.get_errmsg()
        assert nlopt_amd.lib().nlopt_amd_has_device_objective(o._h) == 1
    else:
        o.set_min_objective(TWIN[name])
    if host_eval:
        o.set_param("amd_host_eval", 1)
    x0 = lb + (ub - lb) * np.modf(np.arange(1, n + 1) * 0.6180339887498949)[0]
    return o, x0


@pytest.mark.parametrize("name,n,pop,me", [("myrastrigin", 64, 2000, 9000), ("convexcosh", 10, 300, 4000), ("shubert", 7, 150, 3000),
                                           ("myrastrigin", 512, 3000, 5000)])
def test_crs_with_a_user_kernel_equals_its_host_twin(code_object, name, n, pop, me):
    runs = []
    for device in (True, False):
        o, x0 = make(nlopt_amd.GN_CRS2_LM, name, n, code_object, device)
        o.set_population(pop)
        o.set_maxeval(me)
        o.enable_trace(me + 64)
        nlopt_amd.srand(42)
        x, minf, ret = o.optimize_raw(x0)
        runs.append(dict(x=x, minf=minf, ret=ret, nev=o.get_numevals(), t=o.trace(), st=o.stats()))
    d, h = runs
    assert d["ret"] == h["ret"] and d["nev"] == h["nev"]
    assert np.array_equal(d["t"]["row"], h["t"]["row"]) and np.array_equal(d["t"]["kind"], h["t"]["kind"]) and np.array_equal(d["t"]["accepted"], h["t"]["accepted"])
    scale = np.abs(h["t"]["f"]).mean()
    assert np.all(np.abs(d["t"]["f"] - h["t"]["f"]) <= RTOL * np.maximum(np.abs(h["t"]["f"]), scale))
    assert np.array_equal(d["x"], h["x"])                                   # bit for bit
    assert d["st"]["mt_words"] == h["st"]["mt_words"]
    assert d["st"]["slots_launched"] > d["st"]["rounds"]                    # the device path ran windows of several slots, not one at a time
    if name == "myrastrigin":                                               # and equals the compiled-in objective
        o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, n)
        o.set_lower_bounds(-5.12); o.set_upper_bounds(5.12)
        o.set_min_objective(nlopt_amd.objective("rastrigin"))
        o.set_population(pop); o.set_maxeval(me); o.enable_trace(me + 64); o.set_param("amd_forward", 0)
        nlopt_amd.srand(42)
        x, minf, ret = o.optimize_raw(make(nlopt_amd.GN_CRS2_LM, name, n, code_object)[1])
        assert np.array_equal(o.trace()["row"], d["t"]["row"]) and np.array_equal(x, d["x"])


@pytest.mark.parametrize("alg", [nlopt_amd.GN_ISRES, nlopt_amd.GN_ESCH])
@pytest.mark.parametrize("name,n,pop,me", [("shubert", 6, 60, 1500), ("myrastrigin", 40, 200, 2400)])
def test_isres_and_esch_with_a_user_kernel_equal_the_host_twin(code_object, alg, name, n, pop, me):
    runs = []
    for device in (True, False):
        o, x0 = make(alg, name, n, code_object, device)
        o.set_population(pop)
        o.set_maxeval(me)
        o.enable_trace(me + 64)
        nlopt_amd.srand(7)
        x, minf, ret = o.optimize_raw(x0)
        runs.append(dict(x=x, minf=minf, ret=ret, nev=o.get_numevals(), t=o.trace(), st=o.stats()))
    d, h = runs
    assert d["ret"] == h["ret"] and d["nev"] == h["nev"] and len(d["t"]) == len(h["t"])
    scale = np.abs(h["t"]["f"]).mean()
    assert np.all(np.abs(d["t"]["f"] - h["t"]["f"]) <= RTOL * np.maximum(np.abs(h["t"]["f"]), scale))
    assert d["st"]["mt_words"] == h["st"]["mt_words"]
    assert abs(d["minf"] - h["minf"]) <= RTOL * max(abs(h["minf"]), scale)
    assert np.allclose(d["x"], h["x"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("alg", [nlopt_amd.LD_LBFGS, nlopt_amd.LD_MMA])
@pytest.mark.parametrize("name,n", [("convexcosh", 8), ("shubert", 5), ("myrastrigin", 300)])
def test_local_optimisers_with_a_user_kernel(code_object, alg, name, n):
    """the search runs as a device coroutine; every evaluation is one launch of the user's kernel (value + gradient)"""
    runs = []
    for device in (True, False):
        o, x0 = make(alg, name, n, code_object, device)
        o.set_ftol_rel(1e-10)
        o.set_maxeval(400)
        o.set_param("amd_exact_dot", 1)
        x, minf, ret = o.optimize_raw(x0)
        runs.append(dict(x=x, minf=minf, ret=ret, nev=o.get_numevals()))
    d, h = runs
    assert d["ret"] == h["ret"], (d, h)
    assert abs(d["nev"] - h["nev"]) <= 3
    assert abs(d["minf"] - h["minf"]) <= 1e-8 * max(abs(h["minf"]), 1.0)
    assert np.allclose(d["x"], h["x"], rtol=1e-6, atol=1e-7)


def test_mlsl_with_a_user_kernel_runs_batched_and_matches_the_compiled_in_objective(code_object):
    n = 12
    res = []
    for user in (True, False):
        o = nlopt_amd.Opt(nlopt_amd.G_MLSL_LDS, n)
        o.set_lower_bounds(-5.12); o.set_upper_bounds(5.12)
        if user:
            assert o.set_min_device_objective(code_object, "myrastrigin") > 0          # no host twin: single points go through the kernel
        else:
            o.set_min_objective(nlopt_amd.objective("rastrigin"))
        loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
        loc.set_ftol_rel(1e-8)
        assert nlopt_amd.lib().nlopt_set_local_optimizer(o._h, loc._h) > 0
        o.set_population(40)
        o.set_maxeval(6000)
        o.enable_trace(10000)
        nlopt_amd.srand(3)
        x, minf, ret = o.optimize_raw(np.full(n, 2.2))
        t = o.trace()
        res.append(dict(ret=ret, minf=minf, fs=t[t["kind"] == 3]["f"], nloc=int((t["kind"] == 4).sum()), st=o.stats()))
    u, c = res
    assert u["ret"] == c["ret"]
    k = min(len(u["fs"]), len(c["fs"]))
    assert k >= 40 and np.all(np.abs(u["fs"][:40] - c["fs"][:40]) <= RTOL * np.maximum(np.abs(c["fs"][:40]), 1.0))
    assert abs(u["minf"] - c["minf"]) <= 1e-6 * max(abs(c["minf"]), 1.0)
    assert u["nloc"] > 0 and u["st"]["lbfgs_launches"] >= 1


def test_binding_errors_are_loud(code_object, tmp_path):
    o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, 4)
    assert o.set_min_device_objective(str(tmp_path / "missing.hsaco"), "x") == nlopt_amd.INVALID_ARGS and "could not load" in o.get_errmsg()
    assert o.set_min_device_objective(code_object, "nosuchobjective") == nlopt_amd.INVALID_ARGS and "not found" in o.get_errmsg()
    assert nlopt_amd.lib().nlopt_amd_has_device_objective(o._h) == 0
    assert o.set_min_device_objective(code_object, "shubert") > 0
    o.set_min_objective(lambda x, g: 0.0)                                      # an ordinary objective unbinds it
    assert nlopt_amd.lib().nlopt_amd_has_device_objective(o._h) == 0
