"""STAGED (-m gpu, NOT collected by `pytest tests/`): the ISRES evolve scan with the exp taken off the lanes' serial chain
(hip/isres_evolve2.hip: ev2_scan_fast_kernel, hip/isres_scan_fast.h; launch flag NLA_EVOLVE_FAST_SCAN,
`nlopt_set_param(opt, "amd_isres_fast_scan", 1)`).

Written at the end of round 4 with no GPU minutes left: hipcc builds it (64 VGPRs, no scratch), every other kernel of the file is
unchanged instruction for instruction, its lane walk compiled by g++ returns the exact walk's counts for drawn and adversarial
individuals (tools/scan_fast_check.cpp, tests/test_host_logic.py), the host plumbing runs over the emulated device — but the kernel HAS
NOT RUN ON AN MI355X.  It is off by default; these tests stay out of the driver's `pytest tests/ -m gpu` run (tests/conftest.py:
collect_ignore_glob) until they have been green on a device once, then the file moves up into tests/ as it is.
    python -m pytest tests/staged/test_gpu_isres_fast_scan.py -q -m gpu           (tools/r05_first_call.sh does)"""
import os

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
from test_gpu_isres import IGOLD, assert_same_run, run_amd

pytestmark = pytest.mark.gpu
FAST = {"amd_isres_fast_scan": 1}
EMU = bool(os.environ.get("NLA_TEST_EMU_DEVICE"))


def _rounds(L, n, pop, phase, seed, flags, rounds, near_bounds):
    """one call of nla_k_isres_evolve_rounds_ex on a drawn population; returns everything it wrote (rows, state, statistics, workspace)"""
    from nlopt_amd import DevBuf
    rng = np.random.default_rng(seed)
    ld, surv = (n + 1) & ~1, -(-pop // 7)
    lb, ub = -5.12 * np.ones(n) - rng.random(n), 5.12 * np.ones(n) + rng.random(n)
    X = np.zeros((pop, ld)); S = np.zeros((pop, ld))
    X[:, :n] = lb + (ub - lb) * rng.random((pop, n))
    if near_bounds:                                   # parents close to a bound (many redraws), a few ON it (draws in the undecided band)
        m = rng.random((pop, n))
        X[:, :n] = np.where(m < 0.2, lb + 1e-3 * (ub - lb) * rng.random((pop, n)), X[:, :n])
        X[:, :n] = np.where(m > 0.98, ub, X[:, :n])
    S[:, :n] = (ub - lb) / np.sqrt(n) * 10.0 ** (-3 * rng.random((pop, n)))
    irank = rng.permutation(pop).astype(np.int32)
    zcount = int(pop * (1 + 2 * n) * 1.4) + 4096
    z = rng.standard_normal(zcount)
    state = np.zeros(16, np.int64); state[0] = surv if phase == 0 else 0
    wsb = L.nla_isres_evolve2_ws_bytes(n)
    dX, dS, dz, dlb, dub = (DevBuf.from_array(a) for a in (X, S, z, lb, ub))
    dirank, dinv, dstate, drho = DevBuf.from_array(irank), DevBuf(4 * pop), DevBuf.from_array(state), DevBuf.from_array(np.zeros(4))
    dx0, dws = DevBuf.from_array(X[0].copy()), DevBuf.from_array(np.zeros(wsb, np.uint8))
    assert L.nla_k_isres_inverse(pop, dirank.ptr, dinv.ptr, None) == 0
    tau, taup = 1.0 / np.sqrt(2.0 * np.sqrt(n)), 1.0 / np.sqrt(2.0 * n)
    rc = L.nla_k_isres_evolve_rounds_ex(n, ld, phase, pop, surv, zcount, taup, tau, dlb.ptr, dub.ptr, dz.ptr, dirank.ptr, dinv.ptr, dX.ptr, dS.ptr,
                                        dx0.ptr, dstate.ptr, drho.ptr, dws.ptr, rounds, flags, None)
    assert rc == 0 and L.nla_stream_sync(None) == 0
    return dict(X=dX.to_array(np.float64, pop * ld), S=dS.to_array(np.float64, pop * ld), state=dstate.to_array(np.int64, 16),
                rho=drho.to_array(np.float64, 4), ws=dws.to_array(np.uint8, wsb))


@pytest.mark.skipif(EMU, reason="the emulated device has no look-up rounds to compare")
@pytest.mark.parametrize("n,pop,phase,seed,near", [(256, 2000, 0, 1, False), (256, 2000, 0, 2, True), (64, 3000, 0, 3, True), (7, 900, 0, 4, False),
                                                   (1, 600, 0, 5, True), (300, 1500, 0, 6, True), (1150, 700, 0, 7, False),
                                                   (256, 8000, 1, 8, False), (256, 8000, 1, 9, True), (33, 6000, 1, 10, True)])
def test_fast_scan_rounds_write_what_the_exact_scan_writes(n, pop, phase, seed, near):
    """the launcher with and without NLA_EVOLVE_FAST_SCAN on the same drawn population: the children's rows and sigmas, the state words
    (individuals resolved, deviates consumed, rounds), the redraw statistics and the WHOLE workspace — E (deviates consumed from each of
    the 256 candidate starts of each individual), T (redraws in front of every coordinate chunk), window origins, exact starts — are
    bit-identical; mutation and variation phase, n from 1 to the kernels' limit, parents near and on their bounds"""
    L = nlopt_amd.lib()
    a = _rounds(L, n, pop, phase, seed, 0, 3, near)
    b = _rounds(L, n, pop, phase, seed, nlopt_amd.EVOLVE_FAST_SCAN, 3, near)
    first = -(-pop // 7) if phase == 0 else 0
    assert a["state"][11] >= 2 and a["state"][0] > first + 256, a["state"]     # several rounds resolved something
    for k in ("state", "rho", "X", "S", "ws"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name", sorted(IGOLD))
def test_golden_isres_runs_with_the_fast_scan(name):
    g = IGOLD[name]
    a = run_amd(g["obj"], g["n"], g["pop"], g["seed"], g["nineq"], g["neq"], params=FAST, **g["kwargs"])
    p = O.run_port_isres(g["obj"], g["n"], g["pop"], g["seed"], g["nineq"], g["neq"], **g["kwargs"])
    assert_same_run(a, p)
    assert a["ret"] == g["ret"] and a["nevals"] == g["nevals"]


@pytest.mark.parametrize("obj,n,pop,seed,nineq,neq,kw", [("rastrigin", 64, 1400, 42, 4, 0, dict(maxeval=7000)),
                                                         ("rastrigin", 256, 5000, 42, 4, 0, dict(maxeval=20000)),
                                                         ("griewank", 48, 3000, 2, 2, 1, dict(maxeval=12000)),
                                                         ("sphere", 40, 2000, 3, 0, 0, dict(maxeval=60000)),
                                                         ("rastrigin", 256, 50000, 42, 4, 0, dict(maxeval=150000))])
def test_fast_scan_changes_nothing(obj, n, pop, seed, nineq, neq, kw):
    """same device, same arithmetic for everything that is stored: an ISRES run with the fast scan is bit-identical to the default run —
    every f, the minimiser, the stream position — and takes the same look-up rounds (the sphere case runs 30 generations: a population
    contracting onto a point; the last case is BASELINE config 3 for three generations)"""
    a = run_amd(obj, n, pop, seed, nineq, neq, **kw)
    b = run_amd(obj, n, pop, seed, nineq, neq, params=FAST, **kw)
    assert np.array_equal(a["trace"]["f"], b["trace"]["f"]) and np.array_equal(a["x"], b["x"])
    assert a["minf"] == b["minf"] and a["nevals"] == b["nevals"] and a["stats"]["mt_words"] == b["stats"]["mt_words"]
    assert a["stats"]["evolve_rounds"] == b["stats"]["evolve_rounds"]
