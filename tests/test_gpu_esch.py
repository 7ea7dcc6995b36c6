"""-m gpu: NLOPT_GN_ESCH end to end through the public C API against the CPU oracle (oracle/port_esch.c,
pinned bit-exactly to the real reference).  Bar: same result code, same number of evaluations, same MT19937 stream
position (= every rejection loop took the same number of attempts and every mutation chain the same path), the fitness of
every candidate in order within 1e-10 relative (the Cauchy draws go through tan(): device libm vs glibc differ in the last
bit, so x agrees to rounding, not bit for bit), same best point."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
import nlopt_amd

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def run_amd(obj, n, pop, seed, maxeval=0, stopval=None, host_callback=None):
    assert nlopt_amd.device_count() > 0, "no HIP device: the product has no CPU fallback"
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(nlopt_amd.GN_ESCH, n)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(host_callback if host_callback is not None else nlopt_amd.objective(obj))
    if pop:
        o.set_population(pop)
    if maxeval:
        o.set_maxeval(maxeval)
    if stopval is not None:
        o.set_stopval(stopval)
    o.enable_trace((maxeval or 100000) + 64)
    nlopt_amd.srand(seed)
    x, minf, ret = o.optimize_raw(xs)
    return dict(ret=ret, minf=minf, x=x, nevals=o.get_numevals(), trace=o.trace(), stats=o.stats(), err=o.get_errmsg(),
                after=nlopt_amd.lib().nla_genrand_int32())


def assert_same(a, p):
    assert a["ret"] == p["ret"], (a["ret"], p["ret"], a["err"])
    assert a["nevals"] == p["nevals"]
    assert a["stats"]["mt_words"] == p["words"]
    fa, fp = a["trace"]["f"], p["fseq"]
    assert len(fa) == len(fp)
    scale = max(np.abs(fp).mean(), 1e-300)
    bad = np.nonzero(np.abs(fa - fp) > RTOL * np.maximum(np.abs(fp), scale))[0]
    assert len(bad) == 0, "first differing evaluation: %d of %d (%r vs %r)" % (bad[0], len(fp), fa[bad[0]], fp[bad[0]])
    assert abs(a["minf"] - p["minf"]) <= RTOL * max(abs(p["minf"]), scale)
    assert np.allclose(a["x"], p["x"], rtol=1e-9, atol=1e-9 * max(np.abs(p["x"]).max(), 1.0))


@pytest.mark.parametrize("obj,n,pop,seed,kw", [
    ("rastrigin", 6, 0, 42, dict(maxeval=3000)),                    # default 40 parents / 60 offspring
    ("griewank", 10, 50, 7, dict(maxeval=4000)),
    ("ackley", 3, 7, 3, dict(maxeval=1500)),                        # 3 mutations per generation
    ("sphere", 1, 5, 5, dict(maxeval=400)),                         # n = 1: (no n)/10 = 0 -> 1 mutation
    ("rosenbrock", 30, 200, 11, dict(maxeval=6000)),
    ("levy", 8, 30, 1, dict(stopval=0.5, maxeval=20000)),
    ("rastrigin", 64, 2000, 9, dict(maxeval=17000)),                # 19200 mutation steps per generation: many chain blocks
    ("griewank", 512, 300, 2, dict(maxeval=2100)),
])
def test_esch_matches_oracle(obj, n, pop, seed, kw):
    a = run_amd(obj, n, pop, seed, **kw)
    p = O.run_port_esch(obj, n, pop, seed, **kw)
    assert_same(a, p)


def test_esch_leaves_the_generator_where_the_reference_would():
    a = run_amd("rastrigin", 12, 40, 5, maxeval=1000)
    L = O.port()
    O.run_port_esch("rastrigin", 12, 40, 5, maxeval=1000)
    L.orc_genrand_int32.restype = C.c_uint32
    assert a["after"] == L.orc_genrand_int32()


def test_esch_host_callback_path():
    """any nlopt_func: the evolution stays on the device, the callback runs on the caller's thread in candidate order"""
    calls = []

    def f(x, grad):
        calls.append(x.copy())
        return float(np.sum(x * x) + 3.0 * np.sum(np.cos(x)))
    a = run_amd("sphere", 5, 20, 3, maxeval=600, host_callback=f)
    assert a["ret"] == nlopt_amd.MAXEVAL_REACHED and a["nevals"] == 600 and len(calls) == 600
    fs = np.array([np.sum(c * c) + 3.0 * np.sum(np.cos(c)) for c in calls])
    assert np.array_equal(fs, a["trace"]["f"]) and a["minf"] == fs.min()
    # the same draws as the device-objective run of the same seed: the candidates do not depend on who evaluates them
    b = run_amd("sphere", 5, 20, 3, maxeval=20)                      # the 20 parents: before any selection
    assert np.allclose(np.array([np.sum(c * c) for c in calls[:20]]), b["trace"]["f"], rtol=1e-12)


def test_esch_argument_errors():
    o = nlopt_amd.Opt(nlopt_amd.GN_ESCH, 3)
    o.set_min_objective(nlopt_amd.objective("sphere"))
    x, minf, ret = o.optimize_raw(np.zeros(3))
    assert ret == nlopt_amd.INVALID_ARGS and "finite domain" in o.get_errmsg()
