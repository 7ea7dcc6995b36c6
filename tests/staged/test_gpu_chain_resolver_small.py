"""STAGED (-m gpu, NOT collected by `pytest tests/`): the device-resolved CRS2_LM windows with the resolver wavefront FORCED ON where they are
not the default — the golden cases and drawn configurations with n < 512 and populations barely above n (`amd_forward` = 1; the default
below n = 512 is conservative passes).  Written after round 4's last GPU call: the resolver's kernel test (12 shapes) and five whole-run
comparisons ran on an MI355X (profiles/r04_crs_chain_resolver.txt, tests/test_gpu_chain_resolver.py); THESE tests have only run over
the emulated device (tests/test_crs_windows_emulated.py: the host's verification walk and window plumbing), never on a device.  They
stay out of the driver's `pytest tests/ -m gpu` run (tests/conftest.py: collect_ignore_glob) until they have been green on a device
once, then they move back into tests/test_gpu_chain_resolver.py.
    python -m pytest tests/staged/test_gpu_chain_resolver_small.py -q -m gpu           (tools/r05_first_call.sh does)"""
import numpy as np
import pytest

import _oracle as O
import nlopt_amd  # noqa: F401
from test_gpu_crs import GOLD, assert_same_run, run_amd

pytestmark = pytest.mark.gpu
RES = {"amd_forward": 1, "amd_chain_resolver": 1}


@pytest.mark.parametrize("name", sorted(GOLD))
def test_golden_runs_with_the_resolver(name):
    """the golden CRS2_LM cases (fixtures from the real reference) with every window resolved on the device by the resolver wavefront"""
    g = GOLD[name]
    kw = dict(g["kwargs"])
    a = run_amd(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, params=RES, **kw)
    p = O.run_port_crs(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, **kw)
    assert a["ret"] == g["ret"] and a["nevals"] == g["nevals"] and [float(v).hex() for v in a["x"]] == g["x"]
    assert_same_run(a, p)


@pytest.mark.parametrize("draw", range(24))
def test_drawn_configurations_with_windows_resolved_on_the_device(draw):
    """drawn objective / dimension / population (down to n + 1 rows: every slot depends on most of its predecessors, the worst-row list
    is shorter than the window) / seed / stopping rule / window depth: with every window resolved inside one launch and the chain
    advanced by the resolver wavefront the run is the oracle's, evaluation by evaluation — the host's verification of what each slot
    read (crs_driver.c) is what is exercised on the CPU twin of this test (tests/test_crs_windows_emulated.py)"""
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
    params = dict(RES)
    params["amd_max_spec"] = int(rng.choice([0, 0, 256, 3, 40]))
    if rng.random() < 0.3:
        params["amd_chain_resolver"] = 0                      # the lock version through the same host path
    a = run_amd(obj, n, pop, seed, trace_cap=20000, params=params, **kw)
    p = O.run_port_crs(obj, n, pop, seed, trace_cap=20000, **kw)
    assert_same_run(a, p)
    assert a["stats"]["slots_launched"] >= a["stats"]["slots_used"] > 0
