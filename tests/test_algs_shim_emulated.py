"""the secondary drop-in boundary on a machine without a GPU: oracle/libnlopt_algs_amd_emu.so (the shim over the emulated device layer,
`make -C oracle algsemu`) in front of the real reference library — see tests/_algs_shim_cases.py"""
import os

import pytest

import _algs_shim_cases as S

SHIM = os.path.join(S.ROOT, "oracle", "libnlopt_algs_amd_emu.so")
need = pytest.mark.skipif(not (os.path.exists(S.TESTOPT) and os.path.exists(SHIM)), reason="oracle/_ref/testopt_ref or the emulated shim not built")
ENV = dict(NLA_TEST_EMU_DEVICE="1")


@need
@pytest.mark.parametrize("alg,obj,seed,maxeval,extra", S.CASES)
def test_reference_api_shell_with_the_three_entry_points_preloaded(alg, obj, seed, maxeval, extra):
    S.check_case(SHIM, alg, obj, seed, maxeval, extra, ENV)


@need
def test_the_shim_exports_the_reference_names_and_nothing_else():
    assert S.shim_exports(SHIM) == S.EXPECTED_EXPORTS


@need
@pytest.mark.skipif(not os.path.exists(S.TBOUNDED), reason="t_bounded_ref not built")
@pytest.mark.parametrize("alg", [19, 35])
def test_reference_cpp_client_with_the_shim(alg):
    """the reference's C++ test (functor trampolines, exceptions) over its own library, algorithms taken over"""
    rc, out, err = S.run(S.TBOUNDED, [alg], preload=SHIM, env_extra=ENV)
    assert rc == 0, "\n".join(out) + err
