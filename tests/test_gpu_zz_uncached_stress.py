"""The uncached-memory pool under load (the workaround has no root cause, so the suite bounds it): a loop of CRS2_LM
runs of the golden configurations — the trial points and the control block live in pooled uncached device memory (hip/devrt.hip) —
with small MLSL / ISRES runs and raw uncached allocations of drawn sizes written and freed in between (tools/stress_crs.py
--golden-only --churn --uc-churn), every run compared with the oracle over its full trace.  It runs in a process of its own: a device
fault there is a failed test, not a dead suite."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_crs_runs_stay_on_the_oracles_path_while_uncached_memory_churns():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_crs.py"), "--seconds", "45", "--tag", "suite", "--golden-only", "1",
                        "--churn", "1", "--uc-churn", "48"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("SUMMARY ")]
    assert line, r.stdout[-2000:]
    s = json.loads(line[-1][len("SUMMARY "):])
    assert s["runs"] >= 20, s
    assert s["bad"] == 0, (s, [ln for ln in r.stdout.splitlines() if ln.startswith("DIVERGENCE")][:3])
