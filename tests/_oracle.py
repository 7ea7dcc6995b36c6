"""ctypes bindings for the CPU oracle (oracle/liboracle.so = the port, oracle/_ref/libnlopt_ref.so =
the real reference compiled by oracle/Makefile).  TEST INFRASTRUCTURE: imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")

FUNC = C.CFUNCTYPE(C.c_double, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p)

OBJ = {"rastrigin": 0, "ackley": 1, "griewank": 2, "rosenbrock": 3, "levy": 4, "sphere": 5}


class OrcStop(C.Structure):
    _fields_ = [("n", C.c_uint), ("minf_max", C.c_double), ("ftol_rel", C.c_double), ("ftol_abs", C.c_double),
                ("xtol_rel", C.c_double), ("xtol_abs", C.POINTER(C.c_double)), ("x_weights", C.POINTER(C.c_double)),
                ("nevals", C.c_long), ("maxeval", C.c_long), ("maxtime", C.c_double), ("start", C.c_double),
                ("force_stop", C.c_int)]


class TraceRec(C.Structure):
    _fields_ = [("f", C.c_double), ("row", C.c_int64), ("kind", C.c_int32), ("accepted", C.c_int32)]


class Trace(C.Structure):
    _fields_ = [("rec", C.POINTER(TraceRec)), ("cap", C.c_size_t), ("len", C.c_size_t)]


class Recorder(C.Structure):
    _fields_ = [("inner", C.c_void_p), ("inner_data", C.c_void_p), ("fbuf", C.POINTER(C.c_double)),
                ("xhash", C.POINTER(C.c_uint64)), ("cap", C.c_size_t), ("len", C.c_size_t)]


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORC_DIR, "port"], check=True)
    if os.path.isdir("/root/reference/src"):
        if not os.path.exists(os.path.join(ORC_DIR, "_ref", "libnlopt_ref.so")):
            subprocess.run(["make", "-s", "-C", ORC_DIR, "ref"], check=True)


_port = None
_ref = None


def port():
    global _port
    if _port is None:
        path = os.path.join(ORC_DIR, "liboracle.so")
        if not os.path.exists(path):
            build_oracle()
        L = C.CDLL(path)
        L.orc_urand.restype = C.c_double
        L.orc_urand.argtypes = [C.c_double, C.c_double]
        L.orc_nrand.restype = C.c_double
        L.orc_nrand.argtypes = [C.c_double, C.c_double]
        L.orc_iurand.argtypes = [C.c_int]
        L.orc_genrand_int32.restype = C.c_uint32
        L.orc_srand.argtypes = [C.c_ulong]
        L.orc_mt_words_drawn.restype = C.c_uint64
        L.orc_objective.restype = C.c_void_p
        L.orc_objective.argtypes = [C.c_int]
        L.orc_stop_default.argtypes = [C.POINTER(OrcStop), C.c_uint]
        L.orc_crs_minimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(OrcStop), C.c_long,
                                       C.POINTER(Trace)]
        L.orc_hash_doubles.restype = C.c_uint64
        L.orc_hash_doubles.argtypes = [C.POINTER(C.c_double), C.c_uint]
        L.orc_obj_box.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _port = L
    return _port


def have_ref():
    return os.path.exists(os.path.join(ORC_DIR, "_ref", "libnlopt_ref.so"))


def ref():
    """the real reference library (NLopt 2.11.0 compiled from /root/reference by oracle/Makefile)"""
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(ORC_DIR, "_ref", "libnlopt_ref.so"))
        L.nlopt_create.restype = C.c_void_p
        L.nlopt_create.argtypes = [C.c_int, C.c_uint]
        L.nlopt_destroy.argtypes = [C.c_void_p]
        L.nlopt_srand.argtypes = [C.c_ulong]
        L.nlopt_optimize.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.nlopt_set_min_objective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nlopt_set_max_objective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        for nm in ("nlopt_set_lower_bounds", "nlopt_set_upper_bounds", "nlopt_set_xtol_abs"):
            getattr(L, nm).argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        for nm in ("nlopt_set_stopval", "nlopt_set_ftol_rel", "nlopt_set_ftol_abs", "nlopt_set_xtol_rel",
                   "nlopt_set_maxtime", "nlopt_set_xtol_abs1"):
            getattr(L, nm).argtypes = [C.c_void_p, C.c_double]
        L.nlopt_set_maxeval.argtypes = [C.c_void_p, C.c_int]
        L.nlopt_set_population.argtypes = [C.c_void_p, C.c_uint]
        L.nlopt_get_numevals.argtypes = [C.c_void_p]
        L.nlopt_add_inequality_constraint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.nlopt_add_equality_constraint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.nlopt_set_local_optimizer.argtypes = [C.c_void_p, C.c_void_p]
        L.nlopt_urand.restype = C.c_double
        L.nlopt_urand.argtypes = [C.c_double, C.c_double]
        L.nlopt_nrand.restype = C.c_double
        L.nlopt_nrand.argtypes = [C.c_double, C.c_double]
        L.nlopt_iurand.argtypes = [C.c_int]
        _ref = L
    return _ref


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def golden_x0(obj, n):
    """non-special start point: x0_i = lb + (ub-lb)*frac((i+1)*phi)"""
    lo, hi = C.c_double(), C.c_double()
    port().orc_obj_box(OBJ[obj], C.byref(lo), C.byref(hi))
    i = np.arange(1, n + 1, dtype=np.float64)
    fr = np.modf(i * 0.6180339887498949)[0]
    return lo.value + (hi.value - lo.value) * fr, lo.value, hi.value


def run_port_crs(obj, n, pop, seed, maxeval=0, x0=None, stopval=None, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0,
                 trace_cap=0, record=False):
    """run the port's CRS2_LM; returns dict(ret, minf, x, nevals, trace, words[, fseq, xhash])"""
    L = port()
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lb = np.full(n, lo)
    ub = np.full(n, hi)
    st = OrcStop()
    L.orc_stop_default(C.byref(st), n)
    st.maxeval = maxeval
    st.ftol_rel, st.ftol_abs, st.xtol_rel = ftol_rel, ftol_abs, xtol_rel
    if stopval is not None:
        st.minf_max = stopval
    tr = Trace()
    recs = (TraceRec * max(trace_cap, 1))()
    tr.rec, tr.cap, tr.len = recs, trace_cap, 0
    f = L.orc_objective(OBJ[obj])
    fdata = None
    rec = None
    if record:
        cap = (maxeval or 350000) + 2 * n + 50000
        fbuf = np.zeros(cap)
        hbuf = np.zeros(cap, dtype=np.uint64)
        rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
        f = C.cast(L.orc_recording_callback, C.c_void_p).value
        fdata = C.cast(C.pointer(rec), C.c_void_p)
    minf = C.c_double()
    L.orc_srand(seed)
    ret = L.orc_crs_minimize(n, f, fdata, dptr(lb), dptr(ub), dptr(x), C.byref(minf), C.byref(st), pop, C.byref(tr))
    out = dict(ret=ret, minf=minf.value, x=x, nevals=st.nevals, words=L.orc_mt_words_drawn())
    k = min(tr.len, trace_cap)
    out["trace"] = np.array([(recs[i].f, recs[i].row, recs[i].kind, recs[i].accepted) for i in range(k)],
                            dtype=[("f", "f8"), ("row", "i8"), ("kind", "i4"), ("accepted", "i4")])
    if record:
        out["fseq"] = fbuf[:rec.len].copy()
        out["xhash"] = hbuf[:rec.len].copy()
    return out


def run_ref(alg, obj, n, pop, seed, maxeval=0, x0=None, stopval=None, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0,
            record=True, setup=None, cap=None):
    """run the REAL reference through its public C API with the zoo objective as host callback"""
    R, L = ref(), port()
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lb = np.full(n, lo)
    ub = np.full(n, hi)
    opt = R.nlopt_create(alg, n)
    R.nlopt_set_lower_bounds(opt, dptr(lb))
    R.nlopt_set_upper_bounds(opt, dptr(ub))
    f = L.orc_objective(OBJ[obj])
    cap = cap or (maxeval or 100000) + 2 * n + 50000
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    R.nlopt_set_min_objective(opt, C.cast(L.orc_recording_callback, C.c_void_p).value,
                              C.cast(C.pointer(rec), C.c_void_p))
    if pop:
        R.nlopt_set_population(opt, pop)
    if maxeval:
        R.nlopt_set_maxeval(opt, maxeval)
    if stopval is not None:
        R.nlopt_set_stopval(opt, stopval)
    if ftol_rel:
        R.nlopt_set_ftol_rel(opt, ftol_rel)
    if ftol_abs:
        R.nlopt_set_ftol_abs(opt, ftol_abs)
    if xtol_rel:
        R.nlopt_set_xtol_rel(opt, xtol_rel)
    keep = setup(R, opt) if setup else None
    minf = C.c_double()
    R.nlopt_srand(seed)
    ret = R.nlopt_optimize(opt, dptr(x), C.byref(minf))
    out = dict(ret=ret, minf=minf.value, x=x, nevals=R.nlopt_get_numevals(opt), fseq=fbuf[:rec.len].copy(),
               xhash=hbuf[:rec.len].copy())
    R.nlopt_destroy(opt)
    del keep
    return out


class TraceRecP(C.Structure):
    _fields_ = [("f", C.c_double), ("row", C.c_int64), ("kind", C.c_int32), ("accepted", C.c_int32)]


def _stats_type():
    import nlopt_amd
    return nlopt_amd.Stats          # one definition of nlopt_amd_stats' layout (include/nlopt_amd.h)


class _StatsProxy:
    def __call__(self):
        return _stats_type()()


Stats = _StatsProxy()

TRACE_DT = [("f", "f8"), ("row", "i8"), ("kind", "i4"), ("accepted", "i4")]

_emu = None


def emu():
    """the product's host-side CRS driver over the CPU emulation of the device engine (oracle/libemu.so)"""
    global _emu
    if _emu is None:
        path = os.path.join(ORC_DIR, "libemu.so")
        subprocess.run(["make", "-s", "-C", ORC_DIR, "port", "emu"], check=True)
        port()
        L = C.CDLL(path)
        L.orc_emu_crs.argtypes = [C.c_int, C.c_int, C.c_long, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_long, C.c_double, C.c_double,
                                  C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(_stats_type()),
                                  C.POINTER(C.c_int), C.POINTER(C.c_ulonglong), C.c_double]
        _emu = L
    return _emu


def run_emu_crs(obj, n, pop, seed, maxeval=0, x0=None, stopval=None, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0,
                trace_cap=0, max_slots=0, max_spec=0, host_eval=False, window_factor=0.0):
    E, L = emu(), port()
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lb = np.full(n, lo)
    ub = np.full(n, hi)
    tr = np.zeros(max(trace_cap, 1), dtype=TRACE_DT)
    tlen = C.c_size_t(0)
    st = Stats()
    minf = C.c_double()
    nev = C.c_int()
    words = C.c_ulonglong()
    L.orc_srand(seed)
    ret = E.orc_emu_crs(OBJ[obj], n, pop, dptr(lb), dptr(ub), dptr(x), C.byref(minf), maxeval,
                        -np.inf if stopval is None else stopval, ftol_rel, ftol_abs, xtol_rel, None,
                        max_slots, max_spec, int(host_eval), tr.ctypes.data, trace_cap, C.byref(tlen), C.byref(st),
                        C.byref(nev), C.byref(words), float(window_factor))
    return dict(ret=ret, minf=minf.value, x=x, nevals=nev.value, words=words.value,
                trace=tr[:min(tlen.value, trace_cap)].copy(), trace_len=tlen.value, stats=st.asdict())


# ---- ISRES ------------------------------------------------------------------------------------
class OrcConstraint(C.Structure):
    _fields_ = [("f", C.c_void_p), ("f_data", C.c_void_p), ("tol", C.c_double)]


class IsresTrace(C.Structure):
    _fields_ = [("f", C.POINTER(C.c_double)), ("pen", C.POINTER(C.c_double)), ("cap", C.c_size_t), ("len", C.c_size_t),
                ("generations", C.c_long)]


def blocksum_data(ncon, q0=0):
    """func_data of the block-sum constraints q0..q0+ncon-1 of Q = ncon blocks: unsigned[2] = {q, Q} each"""
    arr = (C.c_uint * (2 * max(ncon, 1)))()
    for q in range(ncon):
        arr[2 * q], arr[2 * q + 1] = q, ncon
    return arr


def run_port_isres(obj, n, pop, seed, nineq=0, neq=0, tol=1e-8, maxeval=0, x0=None, stopval=None, ftol_rel=0.0,
                   ftol_abs=0.0, xtol_rel=0.0, record=True):
    """the port's ISRES with `nineq` block-sum inequality and `neq` block-sum equality constraints"""
    L = port()
    L.orc_isres_minimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OrcConstraint), C.c_int,
                                     C.POINTER(OrcConstraint), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(OrcStop), C.c_long,
                                     C.POINTER(IsresTrace)]
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lb, ub = np.full(n, lo), np.full(n, hi)
    st = OrcStop()
    L.orc_stop_default(C.byref(st), n)
    st.maxeval = maxeval
    st.ftol_rel, st.ftol_abs, st.xtol_rel = ftol_rel, ftol_abs, xtol_rel
    if stopval is not None:
        st.minf_max = stopval
    con = C.cast(L.orc_con_blocksum, C.c_void_p).value
    di, de = blocksum_data(nineq), blocksum_data(neq)
    fc = (OrcConstraint * max(nineq, 1))(*[OrcConstraint(con, C.addressof(di) + 8 * q, tol) for q in range(nineq)])
    hc = (OrcConstraint * max(neq, 1))(*[OrcConstraint(con, C.addressof(de) + 8 * q, tol) for q in range(neq)])
    f = L.orc_objective(OBJ[obj])
    cap = (maxeval or 200000) + 16
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    fcb = C.cast(L.orc_recording_callback, C.c_void_p).value
    tf, tp = np.zeros(cap), np.zeros(cap)
    tr = IsresTrace(dptr(tf), dptr(tp), cap, 0, 0)
    minf = C.c_double()
    L.orc_srand(seed)
    ret = L.orc_isres_minimize(n, fcb, C.cast(C.pointer(rec), C.c_void_p), nineq, fc, neq, hc, dptr(lb), dptr(ub), dptr(x),
                               C.byref(minf), C.byref(st), pop, C.byref(tr))
    k = min(tr.len, cap)
    return dict(ret=ret, minf=minf.value, x=x, nevals=st.nevals, words=L.orc_mt_words_drawn(), fseq=fbuf[:rec.len].copy(),
                xhash=hbuf[:rec.len].copy(), ftrace=tf[:k].copy(), pentrace=tp[:k].copy(), generations=tr.generations)


def run_ref_isres(obj, n, pop, seed, nineq=0, neq=0, tol=1e-8, **kw):
    """the REAL reference's NLOPT_GN_ISRES (35) with the same block-sum constraints"""
    L = port()
    con = C.cast(L.orc_con_blocksum, C.c_void_p).value
    di, de = blocksum_data(nineq), blocksum_data(neq)

    def setup(R, opt):
        for q in range(nineq):
            assert R.nlopt_add_inequality_constraint(opt, con, C.addressof(di) + 8 * q, tol) > 0
        for q in range(neq):
            assert R.nlopt_add_equality_constraint(opt, con, C.addressof(de) + 8 * q, tol) > 0
        return (di, de)
    return run_ref(35, obj, n, pop, seed, setup=setup, **kw)


# ---- LD_LBFGS ---------------------------------------------------------------------------------
def run_port_lbfgs(obj, n, seed=None, x0=None, maxeval=0, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0, stopval=None, mf=0, tolg=0.0,
                   lb=None, ub=None):
    L = port()
    L.orc_lbfgs_minimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(OrcStop), C.c_int, C.c_double]
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lbv = np.full(n, lo) if lb is None else np.array(lb, dtype=np.float64)
    ubv = np.full(n, hi) if ub is None else np.array(ub, dtype=np.float64)
    st = OrcStop()
    L.orc_stop_default(C.byref(st), n)
    st.maxeval = maxeval
    st.ftol_rel, st.ftol_abs, st.xtol_rel = ftol_rel, ftol_abs, xtol_rel
    if stopval is not None:
        st.minf_max = stopval
    f = L.orc_objective(OBJ[obj])
    cap = (maxeval or 100000) + 16
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    minf = C.c_double()
    ret = L.orc_lbfgs_minimize(n, C.cast(L.orc_recording_callback, C.c_void_p).value, C.cast(C.pointer(rec), C.c_void_p),
                               dptr(lbv), dptr(ubv), dptr(x), C.byref(minf), C.byref(st), mf, tolg)
    return dict(ret=ret, minf=minf.value, x=x, nevals=st.nevals, fseq=fbuf[:rec.len].copy(), xhash=hbuf[:rec.len].copy())


def run_ref_lbfgs(obj, n, x0=None, mf=0, lb=None, ub=None, **kw):
    def setup(R, opt):
        R.nlopt_set_vector_storage.argtypes = [C.c_void_p, C.c_uint]
        if mf:
            R.nlopt_set_vector_storage(opt, mf)
        if lb is not None:
            R.nlopt_set_lower_bounds(opt, dptr(np.array(lb, dtype=np.float64)))
        if ub is not None:
            R.nlopt_set_upper_bounds(opt, dptr(np.array(ub, dtype=np.float64)))
    return run_ref(11, obj, n, 0, 0, x0=x0, setup=setup, **kw)


# ---- LD_MMA (no nonlinear constraints) -----------------------------------------------------------
class OrcMma(C.Structure):
    _fields_ = [("rho_init", C.c_double), ("sigma_min", C.c_double), ("inner_maxeval", C.c_int), ("inner_gradients", C.c_int),
                ("always_improve", C.c_int), ("pad", C.c_int), ("sigma_init", C.c_void_p)]


MMA_PARAM_NAMES = ("rho_init", "sigma_min", "inner_maxeval", "inner_gradients", "always_improve")


def mma_params(p=None):
    """the dispatcher's defaults (optimize.c:798-803) overridden by p"""
    d = dict(rho_init=1.0, sigma_min=0.0, inner_maxeval=0, inner_gradients=1, always_improve=1)
    d.update(p or {})
    return OrcMma(d["rho_init"], d["sigma_min"], d["inner_maxeval"], d["inner_gradients"], d["always_improve"], 0, None)


def run_port_mma(obj, n, x0=None, maxeval=0, ftol_rel=0.0, ftol_abs=0.0, xtol_rel=0.0, stopval=None, lb=None, ub=None, params=None,
                 step=None):
    L = port()
    L.orc_mma_minimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(OrcStop), C.POINTER(OrcMma)]
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lbv = np.full(n, lo) if lb is None else np.array(lb, dtype=np.float64)
    ubv = np.full(n, hi) if ub is None else np.array(ub, dtype=np.float64)
    st = OrcStop()
    L.orc_stop_default(C.byref(st), n)
    st.maxeval = maxeval
    st.ftol_rel, st.ftol_abs, st.xtol_rel = ftol_rel, ftol_abs, xtol_rel
    if stopval is not None:
        st.minf_max = stopval
    f = L.orc_objective(OBJ[obj])
    cap = 2 * (maxeval or 200000) + 16          # inner_gradients = 0: up to two calls per counted evaluation
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    minf = C.c_double()
    prm = mma_params(params)
    dx = None if step is None else np.full(n, float(step))
    if dx is not None:
        prm.sigma_init = dx.ctypes.data
    ret = L.orc_mma_minimize(n, C.cast(L.orc_recording_callback, C.c_void_p).value, C.cast(C.pointer(rec), C.c_void_p),
                             dptr(lbv), dptr(ubv), dptr(x), C.byref(minf), C.byref(st), C.byref(prm))
    return dict(ret=ret, minf=minf.value, x=x, nevals=st.nevals, fseq=fbuf[:rec.len].copy(), xhash=hbuf[:rec.len].copy())


def run_ref_mma(obj, n, x0=None, lb=None, ub=None, params=None, step=None, **kw):
    """the REAL reference's NLOPT_LD_MMA (24)"""
    def setup(R, opt):
        R.nlopt_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        R.nlopt_set_initial_step1.argtypes = [C.c_void_p, C.c_double]
        if step is not None:
            assert R.nlopt_set_initial_step1(opt, float(step)) > 0
        for k, v in (params or {}).items():
            assert R.nlopt_set_param(opt, k.encode(), float(v)) > 0
        if lb is not None:
            R.nlopt_set_lower_bounds(opt, dptr(np.array(lb, dtype=np.float64)))
        if ub is not None:
            R.nlopt_set_upper_bounds(opt, dptr(np.array(ub, dtype=np.float64)))
    return run_ref(24, obj, n, 0, 0, x0=x0, setup=setup, **kw)


# ---- MLSL + LD_LBFGS / LD_MMA ---------------------------------------------------------------------
class OrcLocal(C.Structure):
    _fields_ = [("ftol_rel", C.c_double), ("ftol_abs", C.c_double), ("xtol_rel", C.c_double), ("tolg", C.c_double),
                ("maxeval", C.c_long), ("mf", C.c_int), ("alg", C.c_int), ("mma", OrcMma)]


class MlslTrace(C.Structure):
    _fields_ = [("fsamp", C.POINTER(C.c_double)), ("floc", C.POINTER(C.c_double)), ("eloc", C.POINTER(C.c_int)),
                ("cap", C.c_size_t), ("nsamp", C.c_size_t), ("nloc", C.c_size_t), ("iterations", C.c_long),
                ("sloc", C.POINTER(C.c_int)), ("it_nloc", C.POINTER(C.c_long)), ("it_nevals", C.POINTER(C.c_long)),
                ("it_words", C.POINTER(C.c_ulonglong)), ("it_cap", C.c_size_t)]


def run_port_mlsl(obj, n, nsamples, seed, maxeval=0, stopval=None, local_ftol_rel=1e-8, local_xtol_rel=0.0, local_ftol_abs=0.0,
                  local_maxeval=0, mf=0, x0=None, record=True, lds=False, local="lbfgs", local_params=None):
    L = port()
    L.orc_mlsl_set_lds(int(lds))
    L.orc_mlsl_minimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(OrcStop), C.c_int,
                                    C.POINTER(OrcLocal), C.POINTER(MlslTrace)]
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lb, ub = np.full(n, lo), np.full(n, hi)
    st = OrcStop()
    L.orc_stop_default(C.byref(st), n)
    st.maxeval = maxeval
    if stopval is not None:
        st.minf_max = stopval
    loc = OrcLocal(local_ftol_rel, local_ftol_abs, local_xtol_rel, 0.0, local_maxeval, mf, 1 if local == "mma" else 0, mma_params(local_params))
    f = L.orc_objective(OBJ[obj])
    cap = (maxeval or 400000) + 4096
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    fs, fl, el = np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int32)
    sl, itl, ite, itw = np.zeros(cap, dtype=np.int32), np.zeros(4096, dtype=np.int64), np.zeros(4096, dtype=np.int64), np.zeros(4096, dtype=np.uint64)
    tr = MlslTrace(dptr(fs), dptr(fl), el.ctypes.data_as(C.POINTER(C.c_int)), cap, 0, 0, 0, sl.ctypes.data_as(C.POINTER(C.c_int)),
                   itl.ctypes.data_as(C.POINTER(C.c_long)), ite.ctypes.data_as(C.POINTER(C.c_long)), itw.ctypes.data_as(C.POINTER(C.c_ulonglong)), 4096)
    minf = C.c_double()
    L.orc_srand(seed)
    ret = L.orc_mlsl_minimize(n, C.cast(L.orc_recording_callback, C.c_void_p).value, C.cast(C.pointer(rec), C.c_void_p),
                              dptr(lb), dptr(ub), dptr(x), C.byref(minf), C.byref(st), nsamples, C.byref(loc), C.byref(tr))
    L.orc_mlsl_set_lds(0)
    return dict(ret=ret, minf=minf.value, x=x, nevals=st.nevals, words=L.orc_mt_words_drawn(), fseq=fbuf[:rec.len].copy(),
                xhash=hbuf[:rec.len].copy(), fsamp=fs[:tr.nsamp].copy(), floc=fl[:tr.nloc].copy(), eloc=el[:tr.nloc].copy(),
                iterations=tr.iterations, sloc=sl[:tr.nloc].copy(), it_nloc=itl[:min(tr.iterations, 4096)].copy(),
                it_nevals=ite[:min(tr.iterations, 4096)].copy(), it_words=itw[:min(tr.iterations, 4096)].copy())


def port_sobol_points(sdim, skip_n, count, lb=None, ub=None):
    """`count` points of the oracle's stateful Sobol generator after nlopt_sobol_skip(s, skip_n, .) (skip_n = 0: no skip);
    None if there is no generator for this dimension"""
    L = port()
    L.orc_sobol_create.restype = C.c_void_p
    L.orc_sobol_create.argtypes = [C.c_uint]
    L.orc_sobol_destroy.argtypes = [C.c_void_p]
    L.orc_sobol_next01.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.orc_sobol_next.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_sobol_skip.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_double)]
    s = L.orc_sobol_create(sdim)
    if not s:
        return None
    out = np.zeros((count, sdim))
    if skip_n:
        L.orc_sobol_skip(s, skip_n, dptr(out[0]))
    for i in range(count):
        if lb is None:
            L.orc_sobol_next01(s, dptr(out[i]))
        else:
            L.orc_sobol_next(s, dptr(out[i]), dptr(lb), dptr(ub))
    L.orc_sobol_destroy(s)
    return out


def ref_sobol_points(sdim, skip_n, count, lb=None, ub=None):
    """the same from the REAL reference (src/util/sobolseq.c through oracle/_ref)"""
    R = ref()
    R.nlopt_sobol_create.restype = C.c_void_p
    R.nlopt_sobol_create.argtypes = [C.c_uint]
    R.nlopt_sobol_destroy.argtypes = [C.c_void_p]
    R.nlopt_sobol_next01.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    R.nlopt_sobol_next.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    R.nlopt_sobol_skip.argtypes = [C.c_void_p, C.c_uint, C.POINTER(C.c_double)]
    s = R.nlopt_sobol_create(sdim)
    if not s:
        return None
    out = np.zeros((count, sdim))
    if skip_n:
        R.nlopt_sobol_skip(s, skip_n, dptr(out[0]))
    for i in range(count):
        if lb is None:
            R.nlopt_sobol_next01(s, dptr(out[i]))
        else:
            R.nlopt_sobol_next(s, dptr(out[i]), dptr(lb), dptr(ub))
    R.nlopt_sobol_destroy(s)
    return out


def run_ref_mlsl(obj, n, nsamples, seed, alg=38, local_ftol_rel=1e-8, local_xtol_rel=0.0, local_ftol_abs=0.0, local_maxeval=0, mf=0,
                 local="lbfgs", local_params=None, **kw):
    """the REAL reference's G_MLSL (38) with an LD_LBFGS (11) or LD_MMA (24) local optimiser; local=None: no local optimiser is
    set and the tolerances go on the global object, from which the dispatcher builds its default (optimize.c:763-777)"""
    def setup(R, opt):
        R.nlopt_set_vector_storage.argtypes = [C.c_void_p, C.c_uint]
        R.nlopt_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        if local is None:
            return
        loc = R.nlopt_create(24 if local == "mma" else 11, n)
        for k, v in (local_params or {}).items():
            assert R.nlopt_set_param(loc, k.encode(), float(v)) > 0
        if local_ftol_rel:
            R.nlopt_set_ftol_rel(loc, local_ftol_rel)
        if local_ftol_abs:
            R.nlopt_set_ftol_abs(loc, local_ftol_abs)
        if local_xtol_rel:
            R.nlopt_set_xtol_rel(loc, local_xtol_rel)
        if local_maxeval:
            R.nlopt_set_maxeval(loc, local_maxeval)
        if mf:
            R.nlopt_set_vector_storage(loc, mf)
        assert R.nlopt_set_local_optimizer(opt, loc) > 0
        R.nlopt_destroy(loc)           # set_local_optimizer copies it (options.c:824-846)
    return run_ref(alg, obj, n, nsamples, seed, setup=setup, **kw)


# ---- ESCH ---------------------------------------------------------------------------------------
def run_port_esch(obj, n, pop, seed, maxeval=0, stopval=None, x0=None, trace_cap=0):
    """oracle/port_esch.c with np = pop, no = int(pop * 1.5) (0 -> 40 / 60), as the reference's dispatcher passes them"""
    L = port()
    L.orc_esch_minimize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(OrcStop), C.c_long, C.c_long, C.POINTER(Trace)]
    xs, lo, hi = golden_x0(obj, n)
    x = np.array(xs if x0 is None else x0, dtype=np.float64)
    lb, ub = np.full(n, lo), np.full(n, hi)
    st = OrcStop()
    L.orc_stop_default(C.byref(st), n)
    st.maxeval = maxeval
    if stopval is not None:
        st.minf_max = stopval
    f = L.orc_objective(OBJ[obj])
    cap = (maxeval or 100000) + 4096
    fbuf = np.zeros(cap)
    hbuf = np.zeros(cap, dtype=np.uint64)
    rec = Recorder(f, None, dptr(fbuf), hbuf.ctypes.data_as(C.POINTER(C.c_uint64)), cap, 0)
    tr = np.zeros(max(trace_cap, 1), dtype=TRACE_DT)
    t = Trace(tr.ctypes.data_as(C.POINTER(TraceRec)), trace_cap, 0)
    minf = C.c_double()
    L.orc_srand(seed)
    ret = L.orc_esch_minimize(n, C.cast(L.orc_recording_callback, C.c_void_p).value, C.cast(C.pointer(rec), C.c_void_p), dptr(lb), dptr(ub),
                              dptr(x), C.byref(minf), C.byref(st), pop, int(pop * 1.5), C.byref(t) if trace_cap else None)
    return dict(ret=ret, minf=minf.value, x=x, nevals=st.nevals, words=L.orc_mt_words_drawn(), fseq=fbuf[:rec.len].copy(),
                xhash=hbuf[:rec.len].copy(), trace=tr[:min(t.len, trace_cap)].copy())


def run_ref_esch(obj, n, pop, seed, **kw):
    """the REAL reference's NLOPT_GN_ESCH (42)"""
    return run_ref(42, obj, n, pop, seed, **kw)
