"""CPU: the drop-in boundary is a C-ABI shared library — libnlopt_amd.so loads without a GPU and exports every function that
include/nlopt.h (the reference's public API, src/api/nlopt.h) and include/nlopt_amd.h (extension + kernel-level launchers)
declare; enum values that are part of the reference's ABI are what the reference uses; without a HIP device the
optimisers fail loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nlopt_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    """function names declared in a header: `NLOPT_EXTERN(type) name(` (nlopt.h) or `type name(` (nlopt_amd.h)"""
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*(\\\n[^\n]*)*", " ", src, flags=re.M)
    names = set(re.findall(r"NLOPT_EXTERN\([^)]*\)\s*(\w+)\s*\(", src))
    src = re.sub(r"NLOPT_EXTERN\([^)]*\)", "int", src)
    for stmt in src.split(";"):
        stmt = " ".join(stmt.replace('extern "C" {', " ").split())
        if not stmt or stmt.startswith("typedef") or "(" not in stmt or "{" in stmt or "}" in stmt:
            continue
        m = re.match(r"^(?:const\s+)?[A-Za-z_]\w*(?:\s+[A-Za-z_]\w*)*[\s\*]+(\w+)\s*\(", stmt)
        if m:
            names.add(m.group(1))
    return names


@pytest.mark.parametrize("header", ["nlopt.h", "nlopt_amd.h"])
def test_every_declared_function_is_exported(header):
    L = C.CDLL(nlopt_amd.LIB_PATH)
    names = declared_functions(header)
    assert len(names) > (60 if header == "nlopt.h" else 50), sorted(names)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, "declared in include/%s but not exported: %s" % (header, missing)


def test_reference_abi_constants():
    L = nlopt_amd.lib()
    # nlopt_algorithm values of the hot path (src/api/nlopt.h:85-153) and result codes (:162-176)
    for name, val in (("GN_CRS2_LM", 19), ("GN_MLSL", 20), ("GD_MLSL", 21), ("GN_MLSL_LDS", 22), ("GD_MLSL_LDS", 23), ("LD_LBFGS", 11),
                      ("LD_MMA", 24), ("GN_ISRES", 35), ("G_MLSL", 38), ("G_MLSL_LDS", 39), ("GN_ESCH", 42)):
        assert L.nlopt_algorithm_from_string(name.encode()) == val
        assert L.nlopt_algorithm_to_string(val).decode() == name
    for name, val in (("FAILURE", -1), ("INVALID_ARGS", -2), ("OUT_OF_MEMORY", -3), ("ROUNDOFF_LIMITED", -4), ("FORCED_STOP", -5),
                      ("SUCCESS", 1), ("STOPVAL_REACHED", 2), ("FTOL_REACHED", 3), ("XTOL_REACHED", 4), ("MAXEVAL_REACHED", 5),
                      ("MAXTIME_REACHED", 6)):
        assert L.nlopt_result_from_string(name.encode()) == val
    major, minor, bugfix = C.c_int(), C.c_int(), C.c_int()
    L.nlopt_version(C.byref(major), C.byref(minor), C.byref(bugfix))
    assert (major.value, minor.value) == (2, 11)


@pytest.mark.skipif(nlopt_amd.device_count() > 0, reason="a HIP device is visible")
@pytest.mark.parametrize("alg", [nlopt_amd.GN_CRS2_LM, nlopt_amd.GN_ISRES, nlopt_amd.GN_ESCH, nlopt_amd.LD_LBFGS, nlopt_amd.LD_MMA, nlopt_amd.G_MLSL,
                                 nlopt_amd.GD_MLSL, nlopt_amd.GD_MLSL_LDS])
def test_no_device_fails_loudly(alg):
    o = nlopt_amd.Opt(alg, 4)
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    o.set_min_objective(nlopt_amd.objective("sphere"))
    if alg == nlopt_amd.G_MLSL:
        loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, 4)
        nlopt_amd.lib().nlopt_set_local_optimizer(o._h, loc._h)
    o.set_maxeval(100)
    x, minf, ret = o.optimize_raw(np.zeros(4))
    assert ret == nlopt_amd.FAILURE and "no HIP device" in o.get_errmsg()


def test_unprovided_algorithms_refuse_with_a_message():
    o = nlopt_amd.Opt(0, 2)                    # NLOPT_GN_DIRECT
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    o.set_min_objective(nlopt_amd.objective("sphere"))
    x, minf, ret = o.optimize_raw(np.zeros(2))
    assert ret == nlopt_amd.INVALID_ARGS and "not provided" in o.get_errmsg()


def test_argument_errors_of_the_local_optimisers_come_before_any_device_work():
    """LD_MMA's parameters are validated as the reference's dispatcher does (optimize.c:807-815); what the device path does
    not provide is refused by name; what it provides needs a device (constrained LD_MMA, LN_COBYLA, GN_MLSL with its default)"""
    L = nlopt_amd.lib()

    def mk(alg):
        o = nlopt_amd.Opt(alg, 3)
        o.set_lower_bounds(-1.0)
        o.set_upper_bounds(1.0)
        o.set_min_objective(nlopt_amd.objective("sphere"))
        o.set_maxeval(50)
        return o
    for name, val, msg in (("rho_init", -1.0, "rho_init must be positive and finite"), ("inner_gradients", 2, "inner_gradients must be 0 or 1"),
                           ("always_improve", -1, "always_improve must be 0 or 1"), ("sigma_min", -0.5, "sigma_min must be non-negative")):
        o = mk(nlopt_amd.LD_MMA)
        o.set_param(name, val)
        x, minf, ret = o.optimize_raw(np.full(3, 0.5))
        assert ret == nlopt_amd.INVALID_ARGS and msg in o.get_errmsg()
        g = mk(nlopt_amd.G_MLSL)                       # the same through MLSL's local optimiser
        loc = mk(nlopt_amd.LD_MMA)
        loc.set_param(name, val)
        assert L.nlopt_set_local_optimizer(g._h, loc._h) > 0
        x, minf, ret = g.optimize_raw(np.full(3, 0.5))
        assert ret == nlopt_amd.INVALID_ARGS and msg in g.get_errmsg()
    o = mk(nlopt_amd.LD_MMA)
    con = nlopt_amd.NLOPT_FUNC(lambda n, x, g, d: x[0] - 0.25)
    L.nlopt_add_inequality_constraint.argtypes = [C.c_void_p, nlopt_amd.NLOPT_FUNC, C.c_void_p, C.c_double]
    assert L.nlopt_add_inequality_constraint(o._h, con, None, 1e-8) > 0
    x, minf, ret = o.optimize_raw(np.full(3, 0.5))          # constrained LD_MMA is served (mma_host.c) — on a machine with a device
    assert ret == nlopt_amd.FAILURE and "no HIP device" in o.get_errmsg()
    o = mk(nlopt_amd.LN_COBYLA) if hasattr(nlopt_amd, "LN_COBYLA") else None
    if o is not None:                                  # a host algorithm, but not a CPU NLopt either: no device, no run
        x, minf, ret = o.optimize_raw(np.zeros(3))
        assert ret == nlopt_amd.FAILURE and "no HIP device" in o.get_errmsg()
    o = mk(nlopt_amd.GN_MLSL)                          # its default local optimiser LN_COBYLA is served (cobyla_host.c): a valid run,
    x, minf, ret = o.optimize_raw(np.zeros(3))         # which on a machine without a GPU fails loudly like every other one
    assert ret == nlopt_amd.FAILURE and "no HIP device" in o.get_errmsg()


def test_mma_with_maxtime_only_is_accepted_and_needs_a_device():
    """maxtime is observed inside a device-resident search (the kernel polls an abort flag, mma.c:258-260), so LD_MMA with maxtime
    as its only criterion is a valid run; on a machine without a GPU it fails loudly instead of falling back to a CPU path
    (tests/test_gpu_stops.py runs it on the device)"""
    o = nlopt_amd.Opt(nlopt_amd.LD_MMA, 3)
    o.set_lower_bounds(-1.0)
    o.set_upper_bounds(1.0)
    o.set_min_objective(nlopt_amd.objective("sphere"))
    o.set_maxtime(0.5)
    x, minf, ret = o.optimize_raw(np.full(3, 0.5))
    if nlopt_amd.device_count() <= 0:
        assert ret == nlopt_amd.FAILURE and "no HIP device" in o.get_errmsg()
    else:
        assert ret in (nlopt_amd.MAXTIME_REACHED, nlopt_amd.SUCCESS, nlopt_amd.XTOL_REACHED, nlopt_amd.FTOL_REACHED)


def test_hot_kernels_use_no_scratch_memory():
    """the code objects' own metadata (tools/kernel_resources.py: llvm-readelf --notes on the gfx950 ELF of every built object): the
    kernels on the benchmarked paths keep their state in registers / LDS — `.private_segment_fixed_size` is 0 for the CRS2_LM gather /
    chain / finish kernels, the resident L-BFGS kernel in both summation modes, the MLSL, MMA and ISRES evolve kernels.  (A by-value
    kernel argument indexed at run time, or a lambda captured by another lambda, silently moves state to scratch: this is the guard.)"""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.path.join(root, "nlopt_amd", "lib", "obj")) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        import pytest
        pytest.skip("no object files / no llvm-readelf here")
    sys.path.insert(0, os.path.join(root, "tools"))
    import kernel_resources
    ks = kernel_resources.kernels()
    assert len(ks) > 100
    must = ("crs_advance_kernel", "crs_chain_kernel", "crs_finish_kernel", "crs_vitter_kernel", "lbfgs_resident_kernel", "mlsl_dist2_kernel",
            "mma_batch_kernel", "ev2_scan_kernel", "ev2_chain_seg_kernel", "ev2_write_kernel", "isres_stochrank_kernel",
            "mt_rankbits_kernel")
    seen = set()
    for k in ks:
        for m in must:
            if m in k["name"]:
                seen.add(m)
                assert k["scratch"] == "0" and k["vgpr_spill"] == "0", (k["name"], k["scratch"], k["vgpr_spill"])
    assert seen == set(must), set(must) - seen
    # two workgroups of the resident L-BFGS kernel share a CU: at most 256 registers and 80 KB of LDS each
    for k in ks:
        if "lbfgs_resident_kernel" in k["name"]:
            assert int(k["vgpr"]) + int(k["agpr"]) <= 256 and int(k["lds"]) <= 80 * 1024, k
