"""Development check, CPU only: the two batched L-BFGS kernels (hip/lbfgs_kernels.hip = streaming, hip/lbfgs_resident.hip = resident)
compiled by g++ over tools/simt_emu (one std::thread per work-item) and run on the same starts — the resident kernel must
reproduce the streaming kernel's tree-sum mode BIT FOR BIT (f of every evaluation, minimiser, result, counts), and both must
agree with the oracle's sequential-order search (oracle/port_lbfgs.c) to rounding.  The GPU twin of this check is
tests/test_gpu_lbfgs.py::test_resident_kernel_is_the_streaming_kernel; this one exists so that kernel LOGIC can be debugged
without a GPU.       usage: python tools/lbfgs_emu_check.py [quick | wide]
`wide`: workgroups of 64 threads, so that dimensions 1024 < n <= 2048 take the 32-coordinates-per-thread build of the resident kernel
(hip/lbfgs_resident32.hip: 4096 < n <= 8192 on the device) — the same comparison there."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O          # noqa: E402

HIP = os.path.join(ROOT, "nlopt_amd", "csrc", "hip")
WIDE = len(sys.argv) > 1 and sys.argv[1] == "wide"
LB_T = "64" if WIDE else os.environ.get("EMU_LB_T", "128")
OUT = os.path.join(ROOT, "tools", "_build", "liblbfgs_emu%s.so" % ("_wide" if WIDE else ""))


class Params(C.Structure):
    _fields_ = [("minf_max", C.c_double), ("ftol_rel", C.c_double), ("ftol_abs", C.c_double), ("xtol_rel", C.c_double), ("tolg", C.c_double),
                ("maxeval", C.c_int32), ("exact", C.c_int32), ("sign", C.c_double), ("xtol_abs", C.c_void_p), ("x_weights", C.c_void_p),
                ("abort", C.c_void_p), ("ftrace", C.c_void_p), ("ftrace_cap", C.c_int64), ("done", C.c_void_p)]


class Result(C.Structure):
    _fields_ = [("f", C.c_double), ("ret", C.c_int32), ("nevals", C.c_int32), ("iterm", C.c_int32), ("cols", C.c_int32)]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    srcs = [os.path.join(HIP, "lbfgs_kernels.hip"), os.path.join(HIP, "lbfgs_resident.hip"), os.path.join(HIP, "lbfgs_resident32.hip")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(s) for s in srcs + [os.path.join(HIP, "local_common.h")]):
        return
    subprocess.run(["g++", "-O1", "-std=c++17", "-DLB_T=%s" % LB_T, "-ffp-contract=off", "-fPIC", "-shared", "-x", "c++", "-w", "-I", os.path.join(ROOT, "tools", "simt_emu"),
                    "-o", OUT] + srcs + ["-lpthread"], check=True)


def run(L, streaming, obj, n, starts, lo, hi, mf, ftol_rel=1e-8, maxeval=0, sign=1.0, xtol_abs=None, weights=None, trace_cap=1400, exact=False):
    count, ld = starts.shape[0], (n + 1) & ~1
    X = np.zeros((count, ld)); X[:, :n] = starts
    lb = np.ascontiguousarray(lo, dtype=np.float64); ub = np.ascontiguousarray(hi, dtype=np.float64)
    work = np.zeros(count * (4 * ld + 2 * mf)); iwork = np.zeros(count * ld, dtype=np.int32); hist = np.full(count * 2 * mf * ld, np.nan)
    ft = np.full((count, trace_cap), np.nan)
    res = (Result * count)()
    P = Params(-np.inf, ftol_rel, 0.0, 0.0, 0.0, maxeval, (3 if streaming else 1) if exact else (2 if streaming else 0), sign,
               xtol_abs.ctypes.data if xtol_abs is not None else None, weights.ctypes.data if weights is not None else None, None,
               ft.ctypes.data, trace_cap)
    vp = C.c_void_p
    L.nla_k_lbfgs_batch.argtypes = [C.c_int] * 5 + [vp] * 6 + [C.POINTER(Params), vp, vp, vp]
    rc = L.nla_k_lbfgs_batch(O.OBJ[obj], n, ld, mf, count, lb.ctypes.data, ub.ctypes.data, X.ctypes.data, work.ctypes.data, iwork.ctypes.data,
                             hist.ctypes.data, C.byref(P), C.cast(res, vp), None, None)
    assert rc == 0, rc
    return dict(x=X[:, :n].copy(), f=np.array([r.f for r in res]), ret=[r.ret for r in res], nevals=[r.nevals for r in res],
                iterm=[r.iterm for r in res], cols=[r.cols for r in res], ftrace=ft)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    build()
    L = C.CDLL(OUT)
    rng = np.random.default_rng(7)
    cases = [("ackley", 300, 3, None, 0), ("rastrigin", 40, 2, 5, 0), ("rosenbrock", 10, 2, None, 0), ("griewank", 257, 2, 3, 0), ("levy", 33, 2, None, 0),
             ("sphere", 5, 2, None, 0), ("ackley", 600, 2, None, 25), ("rastrigin", 513, 2, 4, 0)]
    if quick:
        cases = cases[:3]
    if WIDE:
        cases = [("sphere", 1100, 1, 3, 0), ("rastrigin", 1500, 2, 4, 14), ("ackley", 2048, 1, 3, 10), ("griewank", 1025, 1, 2, 8)]
    bad = 0
    for obj, n, count, mf, maxeval in cases:
        _, lo, hi = O.golden_x0(obj, n)
        lov, hiv = np.full(n, lo), np.full(n, hi)
        starts = rng.uniform(lo, hi, (count, n))
        starts[0, : max(1, n // 7)] = hi            # coordinates on a bound from the start: the active-set paths
        if n > 8:
            lov[3] = hiv[3] = 0.5 * (lo + hi)         # a fixed coordinate
            starts[:, 3] = lov[3]
        m = mf or min(max(1310720 // n, 10), 400)
        a = run(L, True, obj, n, starts, lov, hiv, m, maxeval=maxeval)
        b = run(L, False, obj, n, starts, lov, hiv, m, maxeval=maxeval)

        def identical(a, b):
            return (np.array_equal(a["x"], b["x"]) and np.array_equal(a["f"], b["f"]) and a["ret"] == b["ret"] and a["nevals"] == b["nevals"] and
                    a["iterm"] == b["iterm"] and a["cols"] == b["cols"] and np.array_equal(a["ftrace"], b["ftrace"], equal_nan=True))
        same = identical(a, b)
        # the reference's summation order: streaming kernel (exact = 3) against the resident kernel (exact = 1), and — the emulated device's
        # libm being the host's — against the oracle's search evaluation by evaluation, bit for bit
        xa = run(L, True, obj, n, starts, lov, hiv, m, maxeval=maxeval, exact=True)
        xb = run(L, False, obj, n, starts, lov, hiv, m, maxeval=maxeval, exact=True)
        xsame = identical(xa, xb)
        oracle_same = True
        # the oracle's sequential-order search from the same starts (tolerance: the summation order differs)
        orc = []
        for s in range(count):
            p = O.run_port_lbfgs(obj, n, x0=starts[s], ftol_rel=1e-8, maxeval=maxeval, mf=mf or 0, lb=lov, ub=hiv)
            orc.append((p["minf"], p["nevals"], p["ret"]))
            ft = xb["ftrace"][s][: p["nevals"]]
            if not (xb["nevals"][s] == p["nevals"] and np.array_equal(ft, p["fseq"][: len(ft)]) and np.array_equal(xb["x"][s], p["x"])):
                oracle_same = False          # (Levy: the device's gradient expression groups its four terms differently from the host loop's accumulation)
        near = all(abs(orc[s][0] - b["f"][s]) <= 1e-7 * max(1.0, abs(orc[s][0])) or abs(orc[s][1] - b["nevals"][s]) > 4 for s in range(count))
        print("%-10s n=%-4d mf=%-4d: streaming f=%s evals=%s ret=%s | resident identical: %s | exact order: resident == streaming: %s, == the oracle bit for bit: %s | oracle (minf, evals, ret) %s %s" %
              (obj, n, m, np.array2string(a["f"], precision=12), a["nevals"], a["ret"], same, xsame, oracle_same, orc, "" if near else "  <-- differs from the oracle"))
        if not xsame:
            bad += 1
        if not same:
            bad += 1
            for s in range(count):
                fa, fb = a["ftrace"][s], b["ftrace"][s]
                d = np.nonzero(~((fa == fb) | (np.isnan(fa) & np.isnan(fb))))[0]
                if len(d):
                    print("   search %d: first differing evaluation %d: %r vs %r" % (s, d[0], fa[d[0]], fb[d[0]]))
    print("FAILED" if bad else "ok: resident == streaming bit for bit on %d cases" % len(cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
