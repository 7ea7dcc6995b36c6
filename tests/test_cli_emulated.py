"""CPU twins of tests/test_gpu_testopt_cli.py and tests/test_gpu_cpp_client.py: the reference's own command-line driver
(test/testopt.c + test/testfuncs.c) and its own C++ client test (test/t_bounded.cxx through the generated nlopt.hpp), the
binaries `make -C oracle cpptest` links against libnlopt_amd.so, run with the emulated library (oracle/libnlopt_amd_emu.so)
preloaded — its nlopt_* symbols take precedence, so the product's API shell and host drivers serve the program over the CPU
stand-in for the device layer.  Everything on that layer follows the reference's operation order with the host's libm: the
printouts must be IDENTICAL to the reference build's for every algorithm of the path, not just to rounding."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
EMU = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
need = pytest.mark.skipif(not all(os.path.exists(p) for p in (EMU, os.path.join(REFDIR, "testopt_amd"), os.path.join(REFDIR, "testopt_ref"),
                                                             os.path.join(REFDIR, "t_bounded_amd"))), reason="oracle/_ref binaries or the emulated library not built")


def run(exe, *args, preload=None):
    env = dict(os.environ)
    if preload:
        env["LD_PRELOAD"] = preload
    r = subprocess.run([os.path.join(REFDIR, exe)] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    return r.returncode, [l for l in r.stdout.splitlines() if not l.startswith("finished after")], r.stderr


@need
@pytest.mark.parametrize("alg", [19, 35, 42])
@pytest.mark.parametrize("obj,seed,maxeval", [(0, 0, 1000), (1, 3, 2000), (5, 7, 1500), (11, 1, 1500), (17, 2, 2500)])
def test_testopt_identical_printout(alg, obj, seed, maxeval):
    rc_e, out_e, err_e = run("testopt_amd", "-r", seed, "-a", alg, "-o", obj, "-e", maxeval, preload=EMU)
    rc_r, out_r, _ = run("testopt_ref", "-r", seed, "-a", alg, "-o", obj, "-e", maxeval)
    assert rc_e == rc_r == 0, err_e
    assert out_e == out_r


@need
@pytest.mark.parametrize("alg", [20, 22, 25])
@pytest.mark.parametrize("obj,seed,maxeval", [(0, 0, 1000), (1, 3, 1000), (5, 7, 1500), (17, 2, 2000)])
def test_testopt_gn_mlsl_and_cobyla_identical_printout(alg, obj, seed, maxeval):
    """GN_MLSL / GN_MLSL_LDS with their default local optimiser, and LN_COBYLA itself, on the reference's zoo — among them
    SURVEY.md section 8(c)'s pin `testopt -r 0 -a 22 -o 1` (f = -1.91322 after 1000 evaluations)"""
    rc_e, out_e, err_e = run("testopt_amd", "-r", seed, "-a", alg, "-o", obj, "-e", maxeval, preload=EMU)
    rc_r, out_r, _ = run("testopt_ref", "-r", seed, "-a", alg, "-o", obj, "-e", maxeval)
    assert rc_e == rc_r == 0, err_e
    assert out_e == out_r


@need
@pytest.mark.parametrize("alg", [19, 35, 42])
def test_testopt_fixed_dimension_identical_printout(alg):
    rc_e, out_e, err_e = run("testopt_amd", "-r", 5, "-a", alg, "-o", 5, "-e", 800, "-b", 1, preload=EMU)
    rc_r, out_r, _ = run("testopt_ref", "-r", 5, "-a", alg, "-o", 5, "-e", 800, "-b", 1)
    assert rc_e == rc_r == 0, err_e
    assert out_e == out_r


@need
@pytest.mark.parametrize("alg", [19, 35, 42])
def test_the_references_cpp_client_test_passes(alg):
    """t_bounded.cxx: functor trampolines, munge hooks, maximisation through nlopt.hpp — exit code 0 as under ctest"""
    rc, out, err = run("t_bounded_amd", alg, preload=EMU)
    assert rc == 0, "\n".join(out) + err


@pytest.mark.skipif(not all(os.path.exists(p) for p in (EMU, os.path.join(REFDIR, "t_tutorial_amd"), os.path.join(REFDIR, "t_tutorial_ref"))),
                    reason="oracle/_ref/t_tutorial_* or the emulated library not built")
def test_the_references_tutorial_program_with_cobyla():
    """t_tutorial.cxx as ctest runs it for LN_COBYLA (test/CMakeLists.txt:19: `t_tutorial 25`): the constrained tutorial problem
    (two cubic inequality constraints with per-constraint data, a half-open box replaced by [1e-6, 10]^2, stopval, initial
    step 0.1) and the string-keyed parameters (set_param / get_param / num_params / nth_param) — exit code 0 and the same
    line, evaluation count included, as the reference build prints"""
    rc_e, out_e, err_e = run("t_tutorial_amd", 25, preload=EMU)
    rc_r, out_r, _ = run("t_tutorial_ref", 25)
    assert rc_e == rc_r == 0, "\n".join(out_e) + err_e
    assert out_e == out_r and "found minimum at f(" in out_e[0]


@pytest.mark.skipif(not all(os.path.exists(p) for p in (EMU, os.path.join(REFDIR, "t_tutorial_amd"), os.path.join(REFDIR, "t_tutorial_ref"))),
                    reason="oracle/_ref/t_tutorial_* or the emulated library not built")
@pytest.mark.parametrize("args", [(), (24,), (31,), (30,)])
def test_the_references_tutorial_program_with_constrained_mma(args):
    """t_tutorial.cxx with its default algorithm, LD_MMA under two nonlinear inequality constraints (the tutorial of the
    reference's documentation; ctest: `t_tutorial 24`): mma_host.c with the dual problems solved through the library's own
    LD_MMA — same line as the reference build prints, evaluation count included"""
    rc_e, out_e, err_e = run("t_tutorial_amd", *args, preload=EMU)
    rc_r, out_r, _ = run("t_tutorial_ref", *args)
    assert rc_e == rc_r == 0, "\n".join(out_e) + err_e
    assert out_e == out_r and ("Method of Moving Asymptotes" in out_e[0] or "Augmented Lagrangian" in out_e[0])   # 31 / 30: LD_ / LN_AUGLAG (auglag_host.c)


@pytest.mark.skipif(not all(os.path.exists(p) for p in (EMU, os.path.join(REFDIR, "cpp_functor_amd"), os.path.join(REFDIR, "cpp_functor_ref"))),
                    reason="oracle/_ref/cpp_functor_* or the emulated library not built")
def test_the_references_functor_program():
    """cpp_functor.cxx (ctest: `cpp_functor 0`): objectives given as std::function objects, nlopt::opt("LD_MMA", 3) by name,
    no bounds at all, the same optimiser reused for a second objective — same printout as the reference build"""
    rc_e, out_e, err_e = run("cpp_functor_amd", preload=EMU)
    rc_r, out_r, _ = run("cpp_functor_ref")
    assert rc_e == rc_r == 0, "\n".join(out_e) + err_e
    assert out_e == out_r and any("Sine regression" in l for l in out_e)


@need
@pytest.mark.parametrize("obj", [0, 1])
@pytest.mark.parametrize("alg", [11, 19, 20, 21, 22, 23, 24, 25])
def test_testopt_as_ctest_runs_it(alg, obj):
    """the reference's ctest matrix `testopt -r 0 -a <0..28> -o <0,1>` (test/CMakeLists.txt:39-66), for the algorithms of it
    that this library serves: LD_LBFGS, CRS2_LM, the four MLSL enums with their default local optimisers, LD_MMA, LN_COBYLA"""
    rc_e, out_e, err_e = run("testopt_amd", "-r", 0, "-a", alg, "-o", obj, preload=EMU)
    rc_r, out_r, _ = run("testopt_ref", "-r", 0, "-a", alg, "-o", obj)
    assert rc_e == rc_r == 0, err_e
    assert out_e == out_r
