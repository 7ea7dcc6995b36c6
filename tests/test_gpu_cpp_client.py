"""-m gpu: the reference's own C++ client test (test/t_bounded.cxx, built unmodified through the generated nlopt.hpp by
`make -C oracle cpptest` where /root/reference exists) linked against libnlopt_amd.so — SURVEY.md §8f.4.  The reference's
ctest runs it for algorithms 19 (CRS2_LM), 35 (ISRES) and 42 (ESCH) (test/CMakeLists.txt:23); it maximises x0^2 + x1^2 on
[0,1]^2 through a std::vector functor (trampoline + munge hooks of nlopt.hpp) and exits 0 iff |f - 2| < 2e-2."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REFDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
EXE = os.path.join(REFDIR, "t_bounded_amd")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/t_bounded_amd not built (no /root/reference at build time)")
@pytest.mark.parametrize("alg", [19, 35, 42])
def test_reference_cpp_client_passes_against_libnlopt_amd(alg):
    r = subprocess.run([EXE, str(alg)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "found minimum at f(" in r.stdout


TUT, TUT_REF = os.path.join(REFDIR, "t_tutorial_amd"), os.path.join(REFDIR, "t_tutorial_ref")


@pytest.mark.skipif(not (os.path.exists(TUT) and os.path.exists(TUT_REF)), reason="oracle/_ref/t_tutorial_* not built (no /root/reference at build time)")
def test_reference_tutorial_program_with_cobyla_against_libnlopt_amd():
    """test/t_tutorial.cxx as ctest runs it for LN_COBYLA (test/CMakeLists.txt:19): same line as the reference build prints"""
    r = subprocess.run([TUT, "25"], capture_output=True, text=True, timeout=300)
    q = subprocess.run([TUT_REF, "25"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and q.returncode == 0, r.stdout + r.stderr
    assert r.stdout == q.stdout
