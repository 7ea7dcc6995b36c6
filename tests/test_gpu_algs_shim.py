"""-m gpu: the secondary drop-in boundary on the MI355X — nlopt_amd/lib/libnlopt_algs_amd.so (crs_minimize / isres_minimize / mlsl_minimize
with the reference's signatures over the HIP kernels) put in front of the REAL, unmodified reference library with LD_PRELOAD: the reference's
testopt prints the same text.  See tests/_algs_shim_cases.py; INTEGRATION.md B."""
import os

import pytest

import _algs_shim_cases as S

pytestmark = pytest.mark.gpu
SHIM = os.path.join(S.ROOT, "nlopt_amd", "lib", "libnlopt_algs_amd.so")
need = pytest.mark.skipif(not os.path.exists(S.TESTOPT), reason="oracle/_ref/testopt_ref not built")


@need
@pytest.mark.parametrize("alg,obj,seed,maxeval,extra", S.CASES)
def test_reference_api_shell_with_the_three_entry_points_preloaded(alg, obj, seed, maxeval, extra):
    assert os.path.exists(SHIM), "libnlopt_algs_amd.so not built (nlopt_amd/_build.py)"
    if alg == 35:
        # ISRES coordinates go through exp / log on the device (one ulp from glibc's): the counts and the return
        # code must be the reference's, the printed minimum to 6 digits
        args = ["-r", seed, "-a", alg, "-o", obj, "-e", maxeval] + list(extra)
        _, ref, _ = S.run(S.TESTOPT, args)
        rc, got, err = S.run(S.TESTOPT, args, preload=SHIM)
        assert rc == 0, err[-2000:]
        pick = lambda ls: [l for l in ls if "evaluations (numevals" in l or l.startswith("return code")]
        assert [l.split(" after ")[-1] for l in pick(got)] == [l.split(" after ")[-1] for l in pick(ref)]
        return
    S.check_case(SHIM, alg, obj, seed, maxeval, extra)


@need
def test_the_shim_exports_the_reference_names_and_nothing_else():
    assert S.shim_exports(SHIM) == S.EXPECTED_EXPORTS
