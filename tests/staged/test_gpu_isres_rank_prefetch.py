"""STAGED (-m gpu, NOT collected by `pytest tests/`): the read-ahead variant of the ISRES stochastic-ranking pipeline
(hip/isres_kernels.hip: isres_stochrank_pre_kernel, launch flag NLA_STOCHRANK_PREFETCH, `nlopt_set_param(opt, "amd_isres_rank_prefetch", 1)`).

Written at the end of round 4 with no GPU minutes left: hipcc builds it (82 VGPRs, no scratch), every other kernel of the file is
unchanged instruction for instruction, the host plumbing runs over the emulated device — but the kernel HAS NOT RUN ON AN MI355X.  It is
off by default; these tests stay out of the driver's `pytest tests/ -m gpu` run (tests/conftest.py: collect_ignore_glob) until they
have been green on a device once, then the file moves up into tests/ as it is.
    python -m pytest tests/staged/test_gpu_isres_rank_prefetch.py -q -m gpu           (tools/r05_first_call.sh does)"""
import os

import numpy as np
import pytest

import _oracle as O
import nlopt_amd
from test_gpu_isres import IGOLD, _serial_stochrank, assert_same_run, run_amd

pytestmark = pytest.mark.gpu
PRE = {"amd_isres_rank_prefetch": 1}


@pytest.mark.parametrize("pop,seed", [(2, 1), (5, 2), (64, 3), (65, 4), (130, 5), (300, 6), (1000, 7), (4100, 8)])
def test_read_ahead_ranking_kernel_against_serial_loop(pop, seed):
    """test_gpu_isres.py::test_stochastic_ranking_kernels_against_serial_loop with the read-ahead kernel, and the two kernels against
    each other (same final order, same per-sweep "swapped" flags); populations of several blocks per unit and of many units"""
    from nlopt_amd import DevBuf
    L = nlopt_amd.lib()
    rng = np.random.default_rng(seed)
    f = rng.integers(0, pop // 2 + 2, pop).astype(np.float64)
    pen = np.where(rng.random(pop) < 0.4, 0.0, rng.integers(1, 6, pop).astype(np.float64))
    words = rng.integers(0, 2**32, 2 * pop * (pop - 1), dtype=np.uint64).astype(np.uint32)
    units = (pop + 63) // 64
    roww = max((pop - 1 + 63) // 64, 1)
    dF, dP, dW = DevBuf.from_array(f), DevBuf.from_array(pen), DevBuf.from_array(words)
    dstreams, dsorted = DevBuf(8 * (units + 1) * pop), DevBuf(4 * pop)
    dbits = DevBuf(8 * pop * roww)
    prog = np.zeros(units + 1, np.int32)
    prog[0] = pop
    assert L.nla_k_isres_rank_count(pop, dF.ptr, dP.ptr, dstreams.ptr, dsorted.ptr, None) == 0
    assert L.nla_k_isres_bits(dW.ptr, 0, pop, pop, dbits.ptr, None) == 0
    out = []
    for flags in (0, nlopt_amd.STOCHRANK_PREFETCH):
        dprog, dticket = DevBuf.from_array(prog), DevBuf.from_array(np.zeros(1, np.int32))
        dsw, dirank = DevBuf(pop), DevBuf(4 * pop)
        assert L.nla_k_isres_stochrank_ex(pop, pop, dstreams.ptr, dprog.ptr, dbits.ptr, dticket.ptr, dsw.ptr, dirank.ptr, None, 1, 0, flags, None) == 0
        assert L.nla_stream_sync(None) == 0
        out.append((dsw.to_array(np.uint8, pop).copy(), dirank.to_array(np.int32, pop).copy(), dprog.to_array(np.int32, units + 1).copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    if not os.environ.get("NLA_TEST_EMU_DEVICE"):                       # (the emulated device states the ranking sequentially: no counters)
        assert np.all(out[1][2] == pop)                                 # every unit published its whole stream
    if pop <= 300:                                                      # the double loop in Python
        u = ((words[0::2] >> 5).astype(np.float64) * 67108864.0 + (words[1::2] >> 6).astype(np.float64)) * (1.0 / 9007199254740992.0)
        low = (0.0 + (1.0 - 0.0) * u) < 0.45
        ref, sweeps = _serial_stochrank(f, pen, lambda i, j: bool(low[i * (pop - 1) + j]), pop, pop)
        if sweeps == pop:
            assert list(out[1][1]) == ref


@pytest.mark.parametrize("name", sorted(IGOLD))
def test_golden_isres_runs_with_read_ahead_ranking(name):
    g = IGOLD[name]
    a = run_amd(g["obj"], g["n"], g["pop"], g["seed"], g["nineq"], g["neq"], params=PRE, **g["kwargs"])
    p = O.run_port_isres(g["obj"], g["n"], g["pop"], g["seed"], g["nineq"], g["neq"], **g["kwargs"])
    assert_same_run(a, p)
    assert a["ret"] == g["ret"] and a["nevals"] == g["nevals"]


@pytest.mark.parametrize("obj,n,pop,seed,nineq,neq,kw", [("rastrigin", 64, 1400, 42, 4, 0, dict(maxeval=7000)),
                                                         ("rastrigin", 256, 5000, 42, 4, 0, dict(maxeval=10000)),
                                                         ("griewank", 48, 3000, 2, 2, 1, dict(maxeval=12000)),
                                                         ("rastrigin", 256, 50000, 42, 4, 0, dict(maxeval=100000))])
def test_read_ahead_changes_nothing(obj, n, pop, seed, nineq, neq, kw):
    """same device, same arithmetic: an ISRES run with the read-ahead ranking kernel is bit-identical to the default run (the last case is
    BASELINE config 3 for two generations)"""
    a = run_amd(obj, n, pop, seed, nineq, neq, **kw)
    b = run_amd(obj, n, pop, seed, nineq, neq, params=PRE, **kw)
    assert np.array_equal(a["trace"]["f"], b["trace"]["f"]) and np.array_equal(a["x"], b["x"])
    assert a["minf"] == b["minf"] and a["nevals"] == b["nevals"] and a["stats"]["mt_words"] == b["stats"]["mt_words"]
