"""child process of tests/test_gpu_multiproc.py::test_rccl_transport_one_rank: ncclCommInitRank / ncclAllGather through the
library's dlopen()ed RCCL on a 1-rank communicator, then a whole ISRES run over that communicator against the run without one.
Runs in its own process so that a communicator creation that never returns (seen once on a GPU box, in RCCL's bootstrap) costs
one test, not the test session.  Prints RCCL1_OK on success."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import nlopt_amd  # noqa: E402
import _oracle as O  # noqa: E402
from test_gpu_multiproc import single  # noqa: E402


def main():
    L = nlopt_amd.lib()
    c = nlopt_amd.Comm.rccl(0, 1, nlopt_amd.rccl_unique_id())
    a = np.arange(1000, dtype=np.float64)
    src, dst = nlopt_amd.DevBuf.from_array(a), nlopt_amd.DevBuf(a.nbytes)
    assert L.nla_comm_allgather_dev(c._h, src.ptr, dst.ptr, a.nbytes, None) == 0
    assert L.nla_stream_sync(None) == 0
    assert np.array_equal(dst.to_array(np.float64, 1000), a)
    assert np.array_equal(c.allgather_host(a[:7]), a[None, :7])
    assert c.counters()["collectives"] == 2
    # a whole ISRES run over the RCCL communicator == the run without one
    args = dict(obj="rastrigin", n=16, pop=120, seed=9, maxeval=480, ncon=2)
    s = single("gpu_isres", args)
    xs, lo, hi = O.golden_x0("rastrigin", 16)
    o = nlopt_amd.Opt(nlopt_amd.GN_ISRES, 16)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective("rastrigin"))
    o.set_population(120)
    o.set_maxeval(480)
    o.add_blocksum_constraints(2, 1e-8)
    o.set_comm(c)
    nlopt_amd.srand(9)
    x, minf, ret = o.optimize_raw(xs)
    assert ret == s["ret"] and minf == s["minf"] and np.array_equal(x, s["x"])
    assert c.counters()["collectives"] == 2 + 4 * 4
    c.destroy()
    print("RCCL1_OK")


if __name__ == "__main__":
    main()
