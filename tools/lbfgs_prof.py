"""Where a batched L-BFGS search spends its time (hip/lbfgs_kernels.hip, -DNLA_LB_PROF): per-phase device time of every search of
one MLSL local phase at config 4 (G_MLSL_LDS + LD_LBFGS, Ackley n=4096, 1000 samples per iteration), read from the instrumented
build of the library (nlopt_amd/lib/libnlopt_amd_prof.so — built here by
    NLOPT_AMD_VARIANT="prof:-DNLA_LB_PROF" python -m nlopt_amd._build
and selected with NLOPT_AMD_LIB).  The shipped library has none of the instrumentation.
usage (GPU box): NLOPT_AMD_LIB=nlopt_amd/lib/libnlopt_amd_prof.so python tools/lbfgs_prof.py [iterations] [exact]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nlopt_amd          # noqa: E402
import _oracle as O       # noqa: E402

PH = ["top-of-loop/init", "pytrcg+pyfut1+pyrmc0", "gnorm+b dots", "Strang loops", "p dot+pytrcs", "ps1l01+x update+project", "objective+gradient",
      "dot(g,s) after eval", "pytrcd", "mxvine+add_active", "-", "-"]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    exact = len(sys.argv) > 2 and sys.argv[2] == "exact"
    n, pop = 4096, 1000
    L = nlopt_amd.lib()
    xs, lo, hi = O.golden_x0("ackley", n)
    o = nlopt_amd.Opt(nlopt_amd.G_MLSL_LDS, n)
    loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
    loc.set_ftol_rel(1e-8)
    if exact:
        loc.set_param("amd_exact_dot", 1)
    L.nlopt_set_local_optimizer(o._h, loc._h)
    o.set_lower_bounds(lo)
    o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective("ackley"))
    o.set_population(pop)
    got = {}

    def hook(gens_done, numevals):
        if gens_done == iters:
            st = o.stats()
            count = 1024
            buf = np.zeros((count, 16), dtype=np.uint64)
            rc = L.nla_lbfgs_prof_read(buf.ctypes.data_as(C.c_void_p), count)
            got["buf"], got["rc"], got["stats"] = buf, rc, st
            o.force_stop()
    o.set_progress(hook)
    nlopt_amd.srand(42)
    o.optimize_raw(xs)
    buf = got["buf"]
    live = buf[:, 13] > 0
    b = buf[live].astype(np.float64)
    print("searches in the last launch: %d   (rc %d)   stats: lbfgs launches %d, %.2f ms device time in all" %
          (live.sum(), got["rc"], got["stats"]["lbfgs_launches"], got["stats"]["t_lbfgs_ms"]))
    it, ev, cols = b[:, 12], b[:, 13], b[:, 14]
    tot = b[:, :12].sum(axis=1) / 100.0        # us
    print("per search: iterations %.1f (max %d), evaluations %.1f (max %d), history columns %.0f (max %d), time %.0f us (min %.0f, max %.0f)" %
          (it.mean(), it.max(), ev.mean(), ev.max(), cols.mean(), cols.max(), tot.mean(), tot.min(), tot.max()))
    print("%-28s %10s %8s %14s" % ("phase", "us/search", "share", "us per unit"))
    units = [it, it, it, cols * 2, it, ev, ev, ev, it, it, it, it]
    uname = ["iter", "iter", "iter", "column", "iter", "eval", "eval", "eval", "iter", "iter", "", ""]
    for i in range(10):
        us = b[:, i] / 100.0
        print("%-28s %10.1f %7.1f%% %10.2f /%s" % (PH[i], us.mean(), 100 * us.sum() / tot.sum(), us.sum() / max(units[i].sum(), 1), uname[i]))
    cu = buf[live][:, 15]
    print("distinct SMIDs used: %d" % len(set(cu.tolist())))


if __name__ == "__main__":
    main()
