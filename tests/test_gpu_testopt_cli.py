"""-m gpu: the reference's command-line driver (test/testopt.c + its objective zoo test/testfuncs.c, built unmodified by
`make -C oracle cpptest`) linked once against the real reference and once against libnlopt_amd.so: same command line,
same printout (minus the wall-clock line).  Objectives are the zoo's C callbacks, i.e. the exact host-callback path."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
AMD, REF = os.path.join(REFDIR, "testopt_amd"), os.path.join(REFDIR, "testopt_ref")
need = pytest.mark.skipif(not (os.path.exists(AMD) and os.path.exists(REF)), reason="oracle/_ref/testopt_* not built")


def run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return [l for l in r.stdout.splitlines() if not l.startswith("finished after")]


@need
@pytest.mark.parametrize("obj,seed,maxeval", [(0, 0, 1000), (1, 3, 2000), (5, 7, 3000), (11, 1, 1500), (17, 2, 4000)])
def test_testopt_crs_identical_printout(obj, seed, maxeval):
    """NLOPT_GN_CRS2_LM (-a 19): bit-identical run, hence identical text (SURVEY.md §8c pins `-r 0 -a 19 -o 0`)"""
    a = run(AMD, "-r", seed, "-a", 19, "-o", obj, "-e", maxeval)
    r = run(REF, "-r", seed, "-a", 19, "-o", obj, "-e", maxeval)
    assert a == r
    if (obj, seed, maxeval) == (0, 0, 1000):
        assert any("Found minimum f = 1.45289e-09 after 1001 evaluations" in l for l in a)


@need
@pytest.mark.parametrize("alg", [20, 22, 25])
@pytest.mark.parametrize("obj,seed,maxeval", [(0, 0, 1000), (1, 3, 1000), (17, 2, 2000)])
def test_testopt_gn_mlsl_and_cobyla_identical_printout(alg, obj, seed, maxeval):
    """NLOPT_GN_MLSL / GN_MLSL_LDS with their default local optimiser LN_COBYLA, and LN_COBYLA itself: the exact host path (samples
    from the device's stream / Sobol kernels, distances on the device, COBYLA on the host) — identical text, among it SURVEY.md
    §8c's pin `testopt -r 0 -a 22 -o 1`"""
    a = run(AMD, "-r", seed, "-a", alg, "-o", obj, "-e", maxeval)
    r = run(REF, "-r", seed, "-a", alg, "-o", obj, "-e", maxeval)
    assert a == r
    if (alg, obj, seed, maxeval) == (22, 1, 0, 1000):
        assert any("f = -1.91322" in l for l in a)


@need
@pytest.mark.parametrize("alg", [19, 35, 42])
def test_testopt_fixed_dimension(alg):
    """-b 1: the driver equates the bounds of dimension 1 — the library eliminates it (optimize.c:1038-1060) and runs in one dimension less"""
    a = run(AMD, "-r", 5, "-a", alg, "-o", 5, "-e", 800, "-b", 1)
    r = run(REF, "-r", 5, "-a", alg, "-o", 5, "-e", 800, "-b", 1)
    if alg == 19:
        assert a == r
    else:
        cnt = lambda lines: [re.search(r"after (\d+) evaluations \(numevals = (\d+)\)", l).groups() for l in lines if "evaluations (numevals" in l]
        assert cnt(a) == cnt(r) and len(cnt(a)) == 1
        assert [l for l in a if l.startswith("return code")] == [l for l in r if l.startswith("return code")]


@need
@pytest.mark.parametrize("alg", [35, 42])
@pytest.mark.parametrize("obj,seed", [(0, 0), (5, 4), (17, 9)])
def test_testopt_isres_esch_same_result(alg, obj, seed):
    """ISRES (-a 35) / ESCH (-a 42): exp / tan on the device vs glibc — same evaluation count and return code, minimum to rounding"""
    a = run(AMD, "-r", seed, "-a", alg, "-o", obj, "-e", 1500)
    r = run(REF, "-r", seed, "-a", alg, "-o", obj, "-e", 1500)

    def found(lines):
        m = [re.search(r"Found minimum f = (\S+) after (\d+) evaluations \(numevals = (\d+)\)", l) for l in lines]
        m = [x for x in m if x][0]
        return float(m.group(1)), int(m.group(2)), int(m.group(3))
    fa, fr = found(a), found(r)
    assert fa[1:] == fr[1:]
    assert abs(fa[0] - fr[0]) <= 1e-5 * max(abs(fr[0]), 1e-12)
    assert [l for l in a if l.startswith("return code")] == [l for l in r if l.startswith("return code")]
    assert a[:6] == r[:6]                       # same problem statement and starting point (drawn from the library's generator)
