"""CPU twin of tests/test_gpu_crs_windows.py: the whole-run tests of that file over the emulated device layer
(oracle/libnlopt_amd_emu.so through tests/_emu_plugin.py, in a pytest process of its own — the package holds one library per process).
The device-resolved CRS2_LM windows are the default at every dimension; what this checks is the HOST's side of them — the driver's in-order
walk that verifies, record by record, what every slot read from which producer (crs_driver.c), the engine's window plumbing
(crs_engine.c: lists as kernel arguments or uploads, ticket accounting, commits) — for the golden cases and for drawn configurations
with populations barely above n, every stopping rule and several window depths, against the oracle evaluation by evaluation.  The HIP
kernels are not run here (the emulation states them sequentially: oracle/port_kernels.c)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")


@pytest.mark.skipif(not os.path.exists(EMU), reason="the emulated library is not built")
def test_device_resolved_windows_host_side_over_the_emulated_device():
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests"), NLA_TEST_EMU_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "_emu_plugin", os.path.join(ROOT, "tests", "test_gpu_crs_windows.py"), "-m", "gpu", "-q",
                        "-p", "no:cacheprovider", "-x", "--tb=short", "-k",
                        "test_golden_runs or test_drawn_configurations or (test_windows_and_conservative_passes and not 100000 and not 4096)"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


SNIPPET = r"""
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import nlopt_amd
nlopt_amd.LIB_PATH = %r
import _oracle as O
out = {}
for label, n, pop, params in (("windows", 24, 3000, {"amd_forward": 1}), ("passes", 24, 3000, {"amd_forward": 0})):
    xs, lo, hi = O.golden_x0("rastrigin", n)
    o = nlopt_amd.Opt(nlopt_amd.GN_CRS2_LM, n)
    o.set_lower_bounds(lo); o.set_upper_bounds(hi)
    o.set_min_objective(nlopt_amd.objective("rastrigin"))
    o.set_population(pop); o.set_maxeval(pop + 4000)
    for k, v in params.items():
        o.set_param(k, v)
    nlopt_amd.srand(7)
    x, minf, ret = o.optimize_raw(xs)
    st = o.stats()
    out[label] = dict(ret=int(ret), minf=minf, rounds=int(st["rounds"]), refreshes=int(st["list_refreshes"]), beside=int(st["list_refreshes_beside_device"]))
print(json.dumps(out))
"""


@pytest.mark.skipif(not os.path.exists(EMU), reason="the emulated library is not built")
def test_the_ordered_sets_upkeep_runs_inside_the_engine_call():
    """crs_driver.c redraws its list of worst rows from the heap while the engine has a pass with the device (ops->set_idle: crs_engine.c
    calls the driver between launch and wait): over the emulated device, for windows and for conservative passes, nearly every redraw
    after the first happens there — a redraw between two launches (the list ran short because the window size jumped) stays the
    exception — and both modes end at the same minimum."""
    import json
    r = subprocess.run([sys.executable, "-c", SNIPPET % (ROOT, os.path.join(ROOT, "tests"), EMU)], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for label in ("windows", "passes"):
        v = d[label]
        assert v["ret"] == 5 and v["refreshes"] >= 3 and v["refreshes"] <= v["rounds"] + 1, v
        assert v["beside"] >= 0.7 * (v["refreshes"] - 1), v
    assert d["windows"]["minf"] == d["passes"]["minf"]
