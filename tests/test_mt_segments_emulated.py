"""CPU twin of tests/test_gpu_mlsl_short_segments.py (an MT19937 device stream cut into shorter segments than the default; MLSL on such
a stream — on the device since round 5: tests/test_gpu_mlsl_short_segments.py): the same tests over the emulated device layer (oracle/libnlopt_amd_emu.so through tests/_emu_plugin.py,
in a pytest process of its own).  The emulated jump-ahead is the host's GF(2) arithmetic and the emulated generator regenerates block by
block, so what is checked here is everything of the feature that is not the 30-line kernel: mtstream.c's segment arithmetic (which
polynomial a doubling round applies, which state a fill starts from, where the host generator is left) and MLSL's plumbing of the option."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")


@pytest.mark.skipif(not os.path.exists(EMU), reason="the emulated library is not built")
def test_short_segment_streams_over_the_emulated_device():
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "tests"), NLA_TEST_EMU_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-p", "_emu_plugin", os.path.join(ROOT, "tests", "test_gpu_mlsl_short_segments.py"),
                        "-m", "gpu", "-q", "-p", "no:cacheprovider", "-x", "--tb=short", "-k",
                        "serial_generator or not_served or (same_run and 64 and not kw2)"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
