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
