"""nlopt_amd/nlopt.py — the reference's Python module (`import nlopt`, the SWIG wrapping of nlopt::opt: src/swig/nlopt.i,
nlopt-python.i, src/api/nlopt-in.hpp) over libnlopt_amd.  Client scripts written against `import nlopt` only are run twice,
once with the module bound to the REAL reference library (oracle/_ref/libnlopt_ref.so) and once bound to the product (here: its
build over the emulated device layer), and must print the same text:

  * the reference's own test/t_python.py as ctest runs it for LN_COBYLA and LD_MMA (test/CMakeLists.txt:76-79), unmodified;
  * the reference's own test/t_memoize.py (all algorithms in one process, unmodified): it must pass, and the blocks of the
    algorithms of the path that draw no random numbers must be identical (the others start from a generator position that
    depends on algorithms outside the path having run before them);
  * tests/pyapi/seeded_runs.py: every algorithm of the path from fixed seeds, constraints (scalar and vector), maximisation,
    copies, every getter / setter, the error mapping, exceptions and wrong return types inside callbacks, force_stop."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libnlopt_ref.so")
EMU = os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so")
SHIM = os.path.join(ROOT, "tests", "pyapi")
REFTEST = "/root/reference/test"
need = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(EMU)), reason="oracle/_ref or the emulated library not built")
need_reftest = pytest.mark.skipif(not os.path.isdir(REFTEST), reason="the reference's test scripts are not on this machine")


def run(script, *args, library=None, timeout=900):
    env = dict(os.environ, PYTHONPATH=SHIM + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("NLOPT_AMD_PYAPI_LIBRARY", None)
    if library:
        env["NLOPT_AMD_PYAPI_LIBRARY"] = library
    r = subprocess.run([sys.executable, script] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env,
                       cwd=SHIM)
    return r.returncode, r.stdout, r.stderr


@need
@need_reftest
@pytest.mark.parametrize("args", [(25,), (24,), (31,), ()])
def test_the_references_t_python_script(args):
    """LN_COBYLA, (constrained) LD_MMA — the script's default — and LD_AUGLAG as ctest runs it (test/CMakeLists.txt:76-79)"""
    rc_r, out_r, err_r = run(os.path.join(REFTEST, "t_python.py"), *args, library=REF)
    rc_a, out_a, err_a = run(os.path.join(REFTEST, "t_python.py"), *args, library=EMU)
    assert rc_r == 0 and rc_a == 0, err_r + err_a
    assert out_a == out_r and "result code:" in out_a


def blocks(text):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"Algo: .* (\d+)$", line)
        if m:
            cur = int(m.group(1))
            out[cur] = []
        elif cur is not None and not line.startswith("-----"):
            out[cur].append(line)
    return out


@need
@need_reftest
def test_the_references_t_memoize_script():
    rc_r, out_r, err_r = run(os.path.join(REFTEST, "t_memoize.py"), library=REF)
    rc_a, out_a, err_a = run(os.path.join(REFTEST, "t_memoize.py"), library=EMU)
    assert rc_r == 0 and rc_a == 0, err_r + err_a
    b_r, b_a = blocks(out_r), blocks(out_a)
    assert sorted(b_r) == sorted(b_a)
    served = [a for a in b_a if any(l.startswith("minimum value") for l in b_a[a])]
    assert sorted(served) == [19, 20, 21, 22, 23, 24, 25, 30, 31, 32, 33, 35, 36, 37, 38, 39, 42]
    for a in (22, 23, 24, 25, 30, 31, 32, 33, 36, 37, 39):   # quasi-random MLSL, MMA, COBYLA, the AUGLAG family: no draws from the shared generator
        assert b_a[a] == b_r[a], a


@need
def test_a_seeded_client_script_prints_the_same_over_both_libraries():
    rc_r, out_r, err_r = run(os.path.join(SHIM, "seeded_runs.py"), library=REF)
    rc_a, out_a, err_a = run(os.path.join(SHIM, "seeded_runs.py"), library=EMU)
    assert rc_r == 0 and rc_a == 0, err_r + err_a
    assert out_a.splitlines() == out_r.splitlines()
    assert out_a.count("\nrun ") == 24 and "ForcedStop NLopt forced stop 60" in out_a and "raised Boom" in out_a


@need
def test_mlsl_with_explicit_local_optimisers_prints_the_same_over_both_libraries():
    rc_r, out_r, err_r = run(os.path.join(SHIM, "mlsl_locals.py"), library=REF)
    rc_a, out_a, err_a = run(os.path.join(SHIM, "mlsl_locals.py"), library=EMU)
    assert rc_r == 0 and rc_a == 0, err_r + err_a
    assert out_a == out_r and len(out_a.splitlines()) == 12 and "Error" not in out_a and "invalid" not in out_a


def test_the_constants_are_the_headers():
    """names and values of nlopt_algorithm / nlopt_result as include/nlopt.h (= src/api/nlopt.h:71-177) declares them"""
    sys.path.insert(0, ROOT)
    import nlopt_amd.nlopt as N
    text = open(os.path.join(ROOT, "include", "nlopt.h")).read()
    body = text[text.index("typedef enum {"):text.index("} nlopt_algorithm;")]
    names = re.findall(r"\bNLOPT_([A-Z][A-Z0-9_]*)", re.sub(r"/\*.*?\*/", "", body, flags=re.S))
    assert names[-1] == "NUM_ALGORITHMS"
    assert names[:-1] == N._ALGORITHMS and N.NUM_ALGORITHMS == len(names) - 1
    for i, nm in enumerate(N._ALGORITHMS):
        assert getattr(N, nm) == i
    res = text[text.index("} nlopt_algorithm;"):text.index("} nlopt_result;")]
    for nm, val in re.findall(r"NLOPT_([A-Z_]+) = (-?\d+)", res):
        assert getattr(N, nm) == int(val), nm


def test_without_a_device_the_module_fails_loudly():
    """bound to the real libnlopt_amd.so on a machine without a GPU: optimize must raise with the library's message, not
    compute anything"""
    code = ("import nlopt_amd, nlopt\n"
            "if nlopt_amd.device_count() > 0:\n    print('has device'); raise SystemExit(0)\n"
            "o = nlopt.opt(nlopt.GN_CRS2_LM, 2)\no.set_min_objective(lambda x, g: float(x[0] ** 2 + x[1] ** 2))\n"
            "o.set_lower_bounds(-1.0); o.set_upper_bounds(1.0); o.set_maxeval(50)\n"
            "try:\n    o.optimize([0.5, 0.5]); print('computed')\n"
            "except Exception as e:\n    print(type(e).__name__, e)\n")
    env = dict(os.environ, PYTHONPATH=SHIM + os.pathsep + ROOT)
    env.pop("NLOPT_AMD_PYAPI_LIBRARY", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "has device" in r.stdout or ("no HIP device" in r.stdout and "computed" not in r.stdout), r.stdout


@need
def test_a_registered_device_objective_through_the_module():
    """device_objective(): pointer identity selects the device evaluator (here: the emulated one); same run as through the
    C API directly"""
    code = ("import ctypes as C, numpy as np, nlopt\n"
            "n = 12; x0 = np.linspace(-300, 300, n)\n"
            "nlopt.srand(11); a = nlopt.opt(nlopt.GN_CRS2_LM, n); a.set_min_objective(nlopt.device_objective('griewank'))\n"
            "a.set_lower_bounds(-600.0); a.set_upper_bounds(600.0); a.set_maxeval(3000); xa = a.optimize(x0)\n"
            "L = C.CDLL(%r); L.nlopt_create.restype = C.c_void_p; L.nlopt_amd_objective.restype = C.c_void_p\n"
            "vp = C.c_void_p; dp = C.POINTER(C.c_double)\n"
            "L.nlopt_srand(C.c_ulong(11)); o = vp(L.nlopt_create(19, n))\n"
            "L.nlopt_set_min_objective(o, vp(L.nlopt_amd_objective(2)), None)\n"
            "L.nlopt_set_lower_bounds1(o, C.c_double(-600.0)); L.nlopt_set_upper_bounds1(o, C.c_double(600.0)); L.nlopt_set_maxeval(o, 3000)\n"
            "xb = x0.copy(); mf = C.c_double(); r = L.nlopt_optimize(o, xb.ctypes.data_as(dp), C.byref(mf))\n"
            "print(r == a.last_optimize_result() == 5, np.array_equal(xa, xb), mf.value == a.last_optimum_value(), mf.value < 30)\n" % EMU)
    env = dict(os.environ, PYTHONPATH=SHIM + os.pathsep + ROOT, NLOPT_AMD_PYAPI_LIBRARY=EMU)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["True"] * 4, r.stdout
