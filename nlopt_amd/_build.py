"""Build libnlopt_amd.so in-tree: HIP kernels with hipcc for gfx950, C host code with gcc, one shared
library exporting the NLopt C API (include/nlopt.h) + the extension / kernel-level C-ABI
(include/nlopt_amd.h).  hipcc cross-compiles without a GPU, so this runs on the build container;
the .so travels to the GPU box with the snapshot (git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# NLOPT_AMD_VARIANT=<name>[:-Dflag ...] builds an instrumented library of its own beside the product (lib/libnlopt_amd_<name>.so,
# objects in lib/obj_<name>) — tools/lbfgs_prof.py uses "prof:-DNLA_LB_PROF"; nothing loads it unless NLOPT_AMD_LIB points there
_VAR = os.environ.get("NLOPT_AMD_VARIANT", "")
_VNAME, _VFLAGS = (_VAR.split(":", 1) + [""])[:2] if _VAR else ("", "")
OBJ = os.path.join(HERE, "lib", "obj" + ("_" + _VNAME if _VNAME else ""))
LIB = os.path.join(HERE, "lib", "libnlopt_amd" + ("_" + _VNAME if _VNAME else "") + ".so")
SHIM = os.path.join(HERE, "lib", "libnlopt_algs_amd" + ("_" + _VNAME if _VNAME else "") + ".so")

HIP_SRC = ["hip/devrt.hip", "hip/mt_kernels.hip", "hip/crs_kernels.hip", "hip/crs_chain.hip", "hip/crs_shard.hip", "hip/isres_kernels.hip", "hip/isres_evolve2.hip", "hip/lbfgs_kernels.hip", "hip/lbfgs_resident.hip", "hip/lbfgs_resident32.hip", "hip/mma_kernels.hip", "hip/cobyla_kernels.hip", "hip/mlsl_kernels.hip", "hip/esch_kernels.hip"]
C_SRC = ["mt_host.c", "mtstream.c", "stopping.c", "objfuncs.c", "api_general.c", "api_options.c", "api_optimize.c",
         "comm.c", "sobol.c", "userobj.c", "crs_driver.c", "crs_engine.c", "isres_driver.c", "lbfgs_driver.c", "mma_driver.c", "mlsl_driver.c", "esch_driver.c", "cobyla_host.c", "mma_host.c", "auglag_host.c"]
# per-source flags.  lbfgs_resident.hip: without machine-level loop-invariant code motion — the pass hoists ~40 VGPRs of literal
# constants (the sin / cos polynomials of the objectives, small integers) out of the search loop and keeps them alive across the
# Strang recurrences, which then spill; with it off the kernel needs 226 VGPRs and no scratch (tools/kernel_resources.py)
EXTRA_FLAGS = {"hip/lbfgs_resident.hip": ["-mllvm", "-disable-machine-licm"], "hip/lbfgs_resident32.hip": ["-mllvm", "-disable-machine-licm"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off everywhere: population rows and trial points must be bit-identical to the
# reference, which is built with it (CMakeLists.txt:280-284).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall"] + _VFLAGS.split()
C_FLAGS = ["-O2", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra", "-fvisibility=default"] + _VFLAGS.split()


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src,) + tuple(extra))


def _headers():
    hs = []
    for root in (CSRC, os.path.join(CSRC, "hip"), os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith(".h"):
                hs.append(os.path.join(root, f))
    return tuple(hs)


def build(force=False, verbose=False):
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()
    objs, jobs = [], []
    for rel in HIP_SRC + C_SRC:
        src = os.path.join(CSRC, rel)
        obj = os.path.join(OBJ, os.path.basename(rel).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            jobs.append(([HIPCC] + HIP_FLAGS if rel.endswith(".hip") else ["gcc"] + C_FLAGS) + EXTRA_FLAGS.get(rel, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    if jobs:                                     # translation units are independent: compile them side by side
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-lm", "-ldl"])
    # the secondary boundary (csrc/shim/algs_shim.c): crs_minimize / isres_minimize / mlsl_minimize with the reference's signatures over the
    # same objects — a library of its own, -Bsymbolic, exporting those names (and the generator's entry points) only: what goes in front of
    # an unmodified libnlopt.so with LD_PRELOAD
    shim_src, shim_map = os.path.join(CSRC, "shim", "algs_shim.c"), os.path.join(CSRC, "shim", "shim.map")
    shim_obj = os.path.join(OBJ, "algs_shim.o")
    if force or _newer(shim_src, shim_obj, hdrs):
        run(["gcc"] + C_FLAGS + ["-c", shim_src, "-o", shim_obj])
    if jobs or force or not os.path.exists(SHIM) or _newer(shim_obj, SHIM, (shim_map,)):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-Wl,--version-script=" + shim_map, "-o", SHIM] + objs + [shim_obj] + ["-lpthread", "-lm", "-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
