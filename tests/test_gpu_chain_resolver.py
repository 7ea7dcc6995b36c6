"""-m gpu: the dedicated chain resolver of the device-resolved CRS2_LM windows (hip/crs_chain_resolver.h, launch flag
NLA_CHAIN_RESOLVER, `nlopt_set_param(opt, "amd_chain_resolver", 0 / 1)`; the default below n = 2048, where the windows are used from
n = 512 on).  Who advances the accept / reject chain inside a launch — the evaluating workgroups under a lock, or one wavefront out of
registers — must change nothing that leaves the launch: same trial points, same records of who read which row from whom, same
decisions.  First run on an MI355X: profiles/r04_crs_chain_resolver.txt (the kernel tests and the whole-run comparisons below)."""
import numpy as np
import pytest

import _oracle as O
import nlopt_amd
from test_gpu_crs import GOLD, assert_same_run, run_amd
from test_gpu_kernels import chain_kernel_case

pytestmark = pytest.mark.gpu
RES = {"amd_forward": 1, "amd_chain_resolver": 1}


@pytest.mark.parametrize("obj,n,N,K,i0", [("rastrigin", 10, 100, 7, 0), ("rastrigin", 10, 11, 9, 10), ("griewank", 64, 70, 60, 33),
                                          ("ackley", 257, 600, 40, 599), ("levy", 128, 140, 130, 17), ("rosenbrock", 512, 700, 256, 3),
                                          ("griewank", 4096, 4200, 24, 4199), ("sphere", 2, 9, 8, 4), ("griewank", 2048, 2100, 20, 77),
                                          ("ackley", 300, 320, 30, 5), ("ackley", 9000, 9100, 6, 5), ("rastrigin", 1000, 1100, 200, 1)])
def test_chain_kernel_with_the_dedicated_resolver(L, obj, n, N, K, i0):
    """the launch of test_gpu_kernels.py::test_chain_kernel_resolves_the_window_like_the_sequential_statement with the chain advanced by
    the resolver wavefront: bit-exact points, the same records of who read what from whom, the same rowstate words and counters"""
    chain_kernel_case(L, obj, n, N, K, i0, nlopt_amd.CHAIN_RESOLVER)


@pytest.mark.parametrize("obj,n,pop,maxeval", [("rastrigin", 512, 100000, 2500), ("rastrigin", 64, 2000, 9000), ("griewank", 4096, 4200, 1200),
                                               ("griewank", 2048, 100000, 1500), ("levy", 300, 5000, 3000)])
def test_the_resolver_changes_nothing_but_who_advances_the_chain(obj, n, pop, maxeval):
    """same device, same gather and evaluation kernels: a run with the resolver is bit-identical to the run with the lock version, and
    the device's predictions are as good (what the host could not verify and recomputed stays rare)"""
    a = run_amd(obj, n, pop, 42, maxeval=maxeval, trace_cap=20000, params={"amd_forward": 1})
    b = run_amd(obj, n, pop, 42, maxeval=maxeval, trace_cap=20000, params=RES)
    assert np.array_equal(a["trace"]["row"], b["trace"]["row"]) and np.array_equal(a["trace"]["f"], b["trace"]["f"])
    assert np.array_equal(a["x"], b["x"]) and a["minf"] == b["minf"] and a["nevals"] == b["nevals"]
    sa, sb = a["stats"], b["stats"]
    assert sb["slots_invalid"] <= sa["slots_invalid"] + 0.02 * sb["slots_launched"] + 4, (sa, sb)
