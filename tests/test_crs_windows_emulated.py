"""CPU twin of tests/test_gpu_chain_resolver.py and tests/staged/test_gpu_chain_resolver_small.py (the windows forced on at small n: not yet
run on a device): the whole-run tests of those files over the emulated device layer
(oracle/libnlopt_amd_emu.so through tests/_emu_plugin.py, in a pytest process of its own — the package holds one library per process).
The device-resolved CRS2_LM windows are the default from n = 512 on; what this checks is the HOST's side of them — the driver's in-order
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
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "_emu_plugin", os.path.join(ROOT, "tests", "test_gpu_chain_resolver.py"),
                        os.path.join(ROOT, "tests", "staged", "test_gpu_chain_resolver_small.py"), "-m", "gpu", "-q",
                        "-p", "no:cacheprovider", "-x", "--tb=short", "-k",
                        "test_golden_runs_with_the_resolver or test_drawn_configurations or (test_the_resolver_changes_nothing and not 100000)"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
