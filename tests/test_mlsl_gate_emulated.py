"""MLSL's gate in front of the sampling phase enqueued ahead (mlsl_driver.c; round-5 advisor): a gate that gives up — the stream it waits for
is not running beside it — says so, the run counts it (nlopt_amd_stats.mlsl_gate_timeouts), launches no further gates and ends as the
run whose gates all opened.  Host logic over the emulated device, whose gate "times out" on request (NLA_EMU_GATE_TIMEOUT); in processes
of their own: the emulated library reads its switches from the environment."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
pytestmark = pytest.mark.skipif(not os.path.exists(EMU), reason="the emulated library is not built")

SCRIPT = r"""
import json, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _emu_plugin
import numpy as np
import nlopt_amd
n = 6
o = nlopt_amd.Opt(nlopt_amd.G_MLSL_LDS, n)
lo, hi = nlopt_amd.objective_box("ackley")
o.set_lower_bounds(lo); o.set_upper_bounds(hi)
o.set_min_objective(nlopt_amd.objective("ackley"))
lopt = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n)
lopt.set_ftol_rel(1e-8)
nlopt_amd.lib().nlopt_set_local_optimizer(o._h, lopt._h)
o.set_population(40); o.set_maxeval(4000)
nlopt_amd.srand(5)
x, minf, ret = o.optimize_raw(np.full(n, 1.5))
st = o.stats()
print(json.dumps(dict(ret=ret, minf=float(minf).hex(), x=[float(v).hex() for v in x], nevals=o.get_numevals(), its=st["generations"],
                      ahead=st["mlsl_sampled_ahead"], gave_up=st["mlsl_gate_timeouts"])))
""" % (ROOT, os.path.join(ROOT, "tests"))


def run(**env):
    e = dict(os.environ, NLA_TEST_EMU_DEVICE="1", **env)
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=600, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_a_gate_that_gives_up_is_counted_and_is_the_last_one():
    a = run()
    b = run(NLA_EMU_GATE_TIMEOUT="1")
    assert a["gave_up"] == 0 and a["its"] >= 3 and a["ahead"] >= a["its"] - 1
    assert b["gave_up"] == 1                    # the first gate gave up; none was launched after it
    for k in ("ret", "minf", "x", "nevals", "its", "ahead"):
        assert a[k] == b[k], k
