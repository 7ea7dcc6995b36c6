"""error paths of the host drivers, on the CPU: over the emulated device layer every allocation of a run (device or pinned host
memory) is made to fail in turn (oracle/emu_device.c: orc_emu_fail_alloc_at) — the run must come back with a negative
nlopt_result and an errmsg, never crash, never report success with a failed allocation behind it, and leave no device-layer object
(buffer, stream, event) alive once the optimiser object is destroyed; the same with every kernel launch failing in turn."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(os.path.dirname(HERE), "oracle", "libnlopt_amd_emu.so")


@pytest.mark.skipif(not os.path.exists(EMU), reason="emulated library not built")
@pytest.mark.parametrize("alg", ["crs", "isres", "esch", "mlsl", "mlsl_mma", "mlsl_grow", "lbfgs", "mma", "mma_con", "auglag"])
def test_every_failing_allocation_is_reported(alg):
    r = subprocess.run([sys.executable, os.path.join(HERE, "_emu_fault_worker.py"), alg], capture_output=True, text=True, timeout=900)
    last = (r.stdout.strip().splitlines() or ["<no output>"])[-1]
    assert r.returncode == 0, "stopped at: %s\n%s" % (last, r.stderr[-2000:])
    assert last.startswith("done") and last.endswith("[]"), last
