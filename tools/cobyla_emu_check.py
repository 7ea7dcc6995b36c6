"""Development check, CPU only: the batched COBYLA kernel (hip/cobyla_kernels.hip, one wavefront per search, lane-parallel) compiled by
g++ over tools/simt_emu (one std::thread per lane, barriers where the kernel has them) against the product's HOST COBYLA
(cobyla_host.c over cobyla_core.h, reached through the emulated device library's nla_k_cobyla_batch: every start through
nlopt_optimize(LN_COBYLA) with the objective in the host callback's summation order).  In exact-order mode every objective value is
the host's bit for bit (sphere / Rosenbrock: no transcendental), so every decision, the evaluation count, the result code and the
minimiser must be IDENTICAL.  The GPU twin is tests/test_gpu_cobyla.py (against the real reference).
       usage: python tools/cobyla_emu_check.py [quick | tiny]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O          # noqa: E402

HIP = os.path.join(ROOT, "nlopt_amd", "csrc", "hip")
OUT = os.path.join(ROOT, "tools", "_build", "libcobyla_emu.so")


class Params(C.Structure):
    _fields_ = [("minf_max", C.c_double), ("ftol_rel", C.c_double), ("ftol_abs", C.c_double), ("xtol_rel", C.c_double),
                ("maxeval", C.c_int32), ("exact", C.c_int32), ("sign", C.c_double), ("xtol_abs", C.c_void_p), ("abort", C.c_void_p), ("done", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("f", C.c_double), ("ret", C.c_int32), ("nevals", C.c_int32), ("iterm", C.c_int32), ("cols", C.c_int32)]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(HIP, "cobyla_kernels.hip")]
    deps = srcs + [os.path.join(HIP, "local_common.h"), os.path.join(HIP, "dev_common.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(s) for s in deps):
        return
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-x", "c++", "-w", "-I", os.path.join(ROOT, "tools", "simt_emu"),
                    "-o", OUT] + srcs + ["-lpthread"], check=True)


def run(L, obj, n, starts, lo, hi, xtol_rel=1e-6, maxeval=0, ftol_rel=0.0, dx=None, exact=1, sign=1.0, minf_max=-np.inf):
    count, ld = starts.shape[0], (n + 1) & ~1
    X = np.zeros((count, ld)); X[:, :n] = starts
    lb = np.ascontiguousarray(lo, dtype=np.float64); ub = np.ascontiguousarray(hi, dtype=np.float64)
    work = np.zeros(max(8, count * 8)); iwork = np.zeros(max(8, count * 8), dtype=np.int32)
    res = (Result * count)()
    P = Params(minf_max, ftol_rel, 0.0, xtol_rel, maxeval, exact, sign, None, None, None)
    vp = C.c_void_p
    L.nla_k_cobyla_batch.argtypes = [C.c_int] * 4 + [vp] * 6 + [C.POINTER(Params), vp, vp]
    L.nla_k_cobyla_batch.restype = C.c_int
    rc = L.nla_k_cobyla_batch(O.OBJ[obj], n, ld, count, lb.ctypes.data, ub.ctypes.data, dx.ctypes.data if dx is not None else None, X.ctypes.data,
                              work.ctypes.data, iwork.ctypes.data, C.byref(P), C.cast(res, vp), None)
    assert rc == 0, rc
    return dict(x=X[:, :n].copy(), f=np.array([r.f for r in res]), ret=[r.ret for r in res], nevals=[r.nevals for r in res])


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    build()
    K = C.CDLL(OUT)                                                        # the kernel on 64 lockstep CPU threads
    H = C.CDLL(os.path.join(ROOT, "oracle", "libnlopt_amd_emu.so"))        # the host algorithm
    rng = np.random.default_rng(11)
    #        obj, n, count, maxeval, xtol_rel, kind of box
    cases = [("sphere", 2, 3, 0, 1e-6, "plain"), ("rosenbrock", 3, 3, 0, 1e-6, "plain"), ("rosenbrock", 6, 2, 600, 1e-7, "plain"),
             ("sphere", 5, 3, 0, 1e-6, "onbound"), ("rosenbrock", 4, 2, 0, 1e-5, "halfinf"), ("sphere", 7, 2, 0, 1e-6, "steps"),
             ("rosenbrock", 12, 2, 1500, 1e-6, "plain"), ("sphere", 24, 1, 0, 1e-4, "plain"), ("rosenbrock", 33, 1, 1200, 1e-4, "onbound")]
    if quick:
        cases = cases[:4]
    if len(sys.argv) > 1 and sys.argv[1] == "tiny":           # seconds: tests/test_host_logic.py runs this
        cases = [("sphere", 2, 2, 0, 1e-6, "plain"), ("sphere", 5, 2, 0, 1e-6, "onbound"), ("rosenbrock", 6, 1, 300, 1e-7, "plain"), ("rosenbrock", 4, 1, 400, 1e-5, "halfinf"),
                 ("sphere", 7, 1, 0, 1e-5, "steps")]
    bad = 0
    for obj, n, count, maxeval, xtol, kind in cases:
        _, lo, hi = O.golden_x0(obj, n)
        lov, hiv = np.full(n, float(lo)), np.full(n, float(hi))
        starts = rng.uniform(lo, hi, (count, n))
        dx = None
        if kind == "onbound":
            starts[0, : max(1, n // 3)] = hi
            starts[-1, -1] = lo
        if kind == "halfinf":
            hiv[0] = np.inf; lov[1] = -np.inf
            if n > 2:
                lov[2] = -np.inf; hiv[2] = np.inf
        if kind == "steps":
            dx = np.linspace(0.3, 1.7, n) * 0.1 * (hi - lo)
        a = run(K, obj, n, starts, lov, hiv, xtol_rel=xtol, maxeval=maxeval, dx=dx)
        b = run(H, obj, n, starts, lov, hiv, xtol_rel=xtol, maxeval=maxeval, dx=dx)
        same = a["ret"] == b["ret"] and a["nevals"] == b["nevals"] and np.array_equal(a["f"], b["f"]) and np.array_equal(a["x"], b["x"])
        print("%-10s n=%-3d %-8s kernel ret %s nevals %s f %s | host ret %s nevals %s f %s  %s"
              % (obj, n, kind, a["ret"], a["nevals"], a["f"], b["ret"], b["nevals"], b["f"], "IDENTICAL" if same else "DIFFERENT"), flush=True)
        bad += 0 if same else 1
    print("cobyla emu check:", "ok" if bad == 0 else "%d case(s) differ" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(main())
