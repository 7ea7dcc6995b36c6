"""re-entrancy across threads (SURVEY.md §8b "Threading": the library is re-entrant on different nlopt_opt objects, the RNG and
the seeded flag are thread-local): eight optimisations — CRS2_LM, ISRES, ESCH, MLSL twice each — run concurrently on eight
threads over the emulated device give exactly the results, evaluation counts and generator positions of the same runs one after
the other.  (ctypes releases the GIL during the C call, so the runs really overlap.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
SNIPPET = r'''
import sys, threading
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, nlopt_amd
nlopt_amd.LIB_PATH = %r
import _oracle as O
L = nlopt_amd.lib()
def job(alg, seed, out, key):
    obj, n = "rastrigin", 8
    xs, lo, hi = O.golden_x0(obj, n)
    o = nlopt_amd.Opt(alg, n)
    o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj))
    o.set_maxeval(3000)
    if alg == nlopt_amd.GN_ISRES:
        o.set_population(80); o.add_blocksum_constraints(2, 1e-8)
    if alg == nlopt_amd.G_MLSL:
        loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n); loc.set_ftol_rel(1e-8); L.nlopt_set_local_optimizer(o._h, loc._h); o.set_population(20)
    nlopt_amd.srand(seed)                # thread-local generator: seeded on the thread that optimises
    x, minf, ret = o.optimize_raw(xs)
    out[key] = (ret, minf, x.copy(), o.get_numevals(), L.nla_genrand_int32())
algs = [nlopt_amd.GN_CRS2_LM, nlopt_amd.GN_ISRES, nlopt_amd.GN_ESCH, nlopt_amd.G_MLSL] * 2
seq, par = {}, {}
for i, a in enumerate(algs):
    t = threading.Thread(target=job, args=(a, 10 + i, seq, i)); t.start(); t.join()
ths = [threading.Thread(target=job, args=(a, 10 + i, par, i)) for i, a in enumerate(algs)]
for t in ths: t.start()
for t in ths: t.join()
assert len(seq) == len(par) == len(algs)
for i in seq:
    assert seq[i][0] == par[i][0] > 0 and seq[i][1] == par[i][1] and np.array_equal(seq[i][2], par[i][2]) and seq[i][3:] == par[i][3:], i
print("THREADS_OK")
'''


@pytest.mark.skipif(not os.path.exists(EMU), reason="emulated library not built")
def test_concurrent_runs_equal_sequential_runs():
    r = subprocess.run([sys.executable, "-c", SNIPPET % (ROOT, os.path.join(ROOT, "tests"), EMU)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "THREADS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
