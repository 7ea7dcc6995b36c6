"""shared by tests/test_algs_shim_emulated.py (CPU) and tests/test_gpu_algs_shim.py (MI355X): the REAL reference library with ITS OWN API
shell — object, setters, dispatcher (src/api/*.c, unmodified, oracle/_ref/libnlopt_ref.so) — running the reference's command-line driver
test/testopt.c, once as it is and once with the three algorithm entry points of this path taken over by LD_PRELOAD:

    crs_minimize (crs.h:34-40), isres_minimize (isres.h:34-41), mlsl_minimize (mlsl.h:34-41)   <-   libnlopt_algs_amd*.so

(nlopt_amd/csrc/shim/algs_shim.c).  Same command line -> the same text, minus the wall-clock line."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
TESTOPT = os.path.join(REFDIR, "testopt_ref")
TBOUNDED = os.path.join(REFDIR, "t_bounded_ref")

# (algorithm, objective, seed, maxeval, extra args): the ctest matrix's algorithms of this path (test/CMakeLists.txt:39-66) and more
CASES = [
    (19, 0, 0, 1000, ()),        # SURVEY.md section 8c pin: "Found minimum f = 1.45289e-09 after 1001 evaluations"
    (19, 5, 7, 3000, ()),
    (19, 17, 2, 4000, ()),
    (19, 5, 5, 800, ("-b", "1")),    # a fixed dimension: the reference's elimdim wrapper sits between its dispatcher and crs_minimize
    (35, 1, 3, 2000, ()),        # ISRES
    (35, 11, 1, 1500, ()),
    (20, 0, 1, 1000, ()),        # GN_MLSL: the dispatcher's default local optimiser LN_COBYLA (an object of the REFERENCE) handed to mlsl_minimize
    (22, 1, 0, 1000, ()),        # SURVEY.md section 8c pin: f = -1.91322
    (21, 5, 2, 1500, ()),        # GD_MLSL: default LD_MMA
    (23, 17, 4, 2000, ()),       # GD_MLSL_LDS
]


def run(exe, args, preload=None, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    if preload:
        env["LD_PRELOAD"] = preload
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    return r.returncode, [l for l in r.stdout.splitlines() if not l.startswith("finished after")], r.stderr


def check_case(shim, alg, obj, seed, maxeval, extra, env_extra=None):
    args = ["-r", seed, "-a", alg, "-o", obj, "-e", maxeval] + list(extra)
    rc0, ref, _ = run(TESTOPT, args)
    rc1, got, err = run(TESTOPT, args, preload=shim, env_extra=env_extra)
    assert rc0 == 0 and rc1 == 0, err[-2000:]
    assert got == ref, "\n".join(got[-6:]) + "\n--- reference:\n" + "\n".join(ref[-6:])
    if (alg, obj, seed, maxeval) == (19, 0, 0, 1000):
        assert any("Found minimum f = 1.45289e-09 after 1001 evaluations" in l for l in got)
    if (alg, obj, seed, maxeval) == (22, 1, 0, 1000):
        assert any("f = -1.91322" in l for l in got)


def shim_exports(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    return sorted(l.split()[-1] for l in out.splitlines() if " T " in l)


EXPECTED_EXPORTS = sorted(["crs_minimize", "isres_minimize", "mlsl_minimize", "nlopt_srand", "nlopt_srand_time", "nlopt_srand_time_default",
                           "nlopt_init_genrand", "nlopt_urand", "nlopt_iurand", "nlopt_nrand"])
