"""-m gpu: the two ways a CRS2_LM run is fed to the device — device-resolved windows (hip/crs_chain.hip: every slot of a window computed,
evaluated and decided inside one launch, the chain advanced by the resolver wavefront of hip/crs_chain_resolver.h; the default at every
dimension since round 5) and conservative passes (hip/crs_kernels.hip; `nlopt_set_param(opt, "amd_forward", 0)`, and what host
objectives and column-sharded multi-rank jobs run on) — must both give the oracle's run, evaluation by evaluation: golden cases,
drawn configurations with populations barely above n (every slot depends on most of its predecessors, the worst-row list is shorter
than the window), every stopping rule, several window depths.  First green on an MI355X in round 5 (profiles/r05_staged_ab.txt); the
CPU twin (tests/test_crs_windows_emulated.py) runs the same tests over the emulated device: the host's verification walk."""
import numpy as np
import pytest

import _oracle as O
import nlopt_amd  # noqa: F401
from test_gpu_crs import GOLD, assert_same_run, run_amd

pytestmark = pytest.mark.gpu
PASSES = {"amd_forward": 0}


@pytest.mark.parametrize("name", sorted(GOLD))
def test_golden_runs_on_conservative_passes(name):
    """the golden CRS2_LM cases (fixtures from the real reference) on the path that is no longer the default (tests/test_gpu_crs.py runs
    them on the windows)"""
    g = GOLD[name]
    kw = dict(g["kwargs"])
    a = run_amd(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, params=PASSES, **kw)
    p = O.run_port_crs(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, **kw)
    assert a["ret"] == g["ret"] and a["nevals"] == g["nevals"] and [float(v).hex() for v in a["x"]] == g["x"]
    assert_same_run(a, p)


def well_conditioned_prefix(p, pop, floor):
    """The parity bar is f to 1e-10 and exact indices; device and host libm differ in the last bit, so once a toy
    population has collapsed onto one point to within rounding (levy n = 3, pop = 21 after 900 evaluations: every f = 9 +- 1e-12) the
    ORDER of two rows can differ for a 1-ulp reason and the runs part ways — on every device path alike (round 5, tools/dbg_case.py).
    Returns the evaluation budget up to which an accepted value still differs from the best by more than 1e-9 relative."""
    t = p["trace"]
    f, acc = t["f"], t["accepted"] != 0
    best = np.minimum.accumulate(f)
    close = acc & (np.arange(len(f)) >= pop) & (np.abs(f - best) <= 1e-9 * np.maximum(np.abs(best), 1e-300)) & (f > best)
    idx = np.flatnonzero(close)
    return None if not idx.size else max(floor, int(idx[0]) - 4)


CUTS = []          # (draw, compared up to, drawn budget) of the drawn configurations that were cut short


@pytest.mark.parametrize("draw", range(24))
def test_drawn_configurations(draw):
    """drawn objective / dimension / population (down to n + 1 rows) / seed / stopping rule / window depth / path: the run is the
    oracle's, evaluation by evaluation — the host's verification of what each slot read (crs_driver.c) is what is exercised on the CPU
    twin of this test"""
    rng = np.random.default_rng(4100 + draw)
    obj = ["rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere"][int(rng.integers(6))]
    n = int(rng.integers(2, 97))
    pop = int(rng.integers(n + 1, 10 * n + 20))
    seed = int(rng.integers(1, 2 ** 31))
    kw = dict(maxeval=int(rng.integers(pop + 20, pop + 2500)))
    r = rng.random()
    if r < 0.25:
        kw["ftol_rel"] = 10.0 ** -int(rng.integers(2, 8))
    elif r < 0.4:
        kw["xtol_rel"] = 10.0 ** -int(rng.integers(2, 6))
    elif r < 0.5:
        kw["ftol_abs"] = 10.0 ** -int(rng.integers(1, 6))
    params = {"amd_max_spec": int(rng.choice([0, 0, 256, 3, 40]))}
    if rng.random() < 0.3:
        params["amd_forward"] = 0
    p = O.run_port_crs(obj, n, pop, seed, trace_cap=20000, **kw)
    cut = well_conditioned_prefix(p, pop, pop + 20)
    if cut is not None and cut < kw["maxeval"]:
        # (round-5 advisor: a cut must not hollow the case out silently — it is reported, and a draw whose comparison would shrink
        # to the population's evaluations and a handful of trials fails instead of passing on nothing)
        print("draw %d (%s n=%d pop=%d): compared up to evaluation %d of %d (the population has collapsed to within rounding there)" % (draw, obj, n, pop, cut, kw["maxeval"]))
        assert cut >= pop + 20, (draw, cut, pop)
        CUTS.append((draw, cut, kw["maxeval"]))
        kw["maxeval"] = cut
        p = O.run_port_crs(obj, n, pop, seed, trace_cap=20000, **kw)
    a = run_amd(obj, n, pop, seed, trace_cap=20000, params=params, **kw)
    assert_same_run(a, p)
    assert a["stats"]["slots_launched"] >= a["stats"]["slots_used"] > 0


def test_few_drawn_configurations_were_cut_short():
    """the drawn cases above are compared over their whole budget, bar a few toy populations that collapse onto one point"""
    assert len(CUTS) <= 6, CUTS


@pytest.mark.parametrize("obj,n,pop,maxeval", [("rastrigin", 512, 100000, 102500), ("rastrigin", 64, 2000, 9000), ("griewank", 4096, 4200, 5400),
                                               ("griewank", 2048, 100000, 101500), ("levy", 300, 5000, 8000), ("ackley", 96, 1501, 4000),
                                               ("rosenbrock", 1000, 3000, 5000)])
def test_windows_and_conservative_passes_give_the_same_run(obj, n, pop, maxeval):
    """same device, same gather arithmetic, same reduction of f (8 wavefronts' worth, whatever the workgroup: dev_common.h,
    nla_block_objective_as): a run on device-resolved windows is bit-identical to the run on conservative passes — f included — and
    what the host could not verify of a window and recomputed stays rare.  (maxeval counts the population's evaluations too.)"""
    a = run_amd(obj, n, pop, 42, maxeval=maxeval, trace_cap=maxeval + 1000, params=PASSES)
    b = run_amd(obj, n, pop, 42, maxeval=maxeval, trace_cap=maxeval + 1000)
    assert a["nevals"] >= maxeval > pop
    assert np.array_equal(a["trace"]["row"], b["trace"]["row"]) and np.array_equal(a["trace"]["f"], b["trace"]["f"])
    assert np.array_equal(a["x"], b["x"]) and a["minf"] == b["minf"] and a["nevals"] == b["nevals"]
    sb = b["stats"]
    if pop >= 64 * n:           # (a population of a few n: every trial reads rows its predecessors replace and new best points abound — the
        #                         driver's window then settles at 1.5 x what a pass consumes, i.e. up to a third of the slots is dropped)
        assert sb["slots_invalid"] <= 0.10 * sb["slots_launched"] + 4, sb
