"""`import nlopt_amd.nlopt as nlopt` — the reference's Python module, over libnlopt_amd.so.

The reference's Python interface is the SWIG wrapping of its C++ class `nlopt::opt` (src/swig/nlopt.i, nlopt-python.i,
nlopt-exceptions.i over src/api/nlopt-in.hpp).  This module offers the same names with the same argument meaning and error
behaviour on top of the C-ABI, so that a script written for the reference runs with only its import line changed
(test/t_python.py and test/t_memoize.py run unmodified in tests/test_python_module.py):

  * the algorithm / result constants of nlopt.h, NUM_ALGORITHMS, `__version__`, srand / srand_time / version_* /
    algorithm_name (nlopt-in.hpp:618-640);
  * `opt(algorithm, n)`, `opt("name", n)`, `opt(other)` (copy), `opt()` (uninitialised) and every method of nlopt::opt
    that SWIG exports (nlopt-in.hpp:296-612; get_initial_step is the value-returning variant, nlopt.i:40-47);
  * objective and constraint callbacks f(x, grad) / fc(result, x, grad) on numpy arrays: x is a read-only view of the
    library's buffer, grad a writable view, or an array of size 0 when no gradient is wanted (nlopt-python.i:142-213);
    the value must be a Python float or int (nlopt-python.i:166-177);
  * an exception raised inside a callback stops the run and is re-raised by optimize (nlopt-python.i:162-165); negative
    results become exceptions as nlopt::opt::mythrow maps them (nlopt-in.hpp:87-95): RuntimeError (FAILURE), ValueError
    (INVALID_ARGS; raised as `nlopt.invalid_argument`, which is a ValueError), MemoryError, nlopt.RoundoffLimited,
    nlopt.ForcedStop — unless set_exceptions_enabled(False).

Additions (not in the reference): `device_objective(name_or_id)` returns a handle for one of the library's registered
device objectives; passing it to set_min_objective / set_max_objective selects the HIP evaluator (include/nlopt_amd.h).

The library behind the module is nlopt_amd.lib(); the environment variable NLOPT_AMD_PYAPI_LIBRARY names another shared
library with the NLopt C API instead (the tests use it to run the same scripts over the real reference build)."""
import ctypes as _C
import os as _os
import threading as _threading

import numpy as _np

# ---- nlopt_algorithm (src/api/nlopt.h:71-152; ABI values) and nlopt_result (nlopt.h:163-177) -------------------------
_ALGORITHMS = ("GN_DIRECT GN_DIRECT_L GN_DIRECT_L_RAND GN_DIRECT_NOSCAL GN_DIRECT_L_NOSCAL GN_DIRECT_L_RAND_NOSCAL "
               "GN_ORIG_DIRECT GN_ORIG_DIRECT_L GD_STOGO GD_STOGO_RAND LD_LBFGS_NOCEDAL LD_LBFGS LN_PRAXIS LD_VAR1 LD_VAR2 "
               "LD_TNEWTON LD_TNEWTON_RESTART LD_TNEWTON_PRECOND LD_TNEWTON_PRECOND_RESTART GN_CRS2_LM GN_MLSL GD_MLSL "
               "GN_MLSL_LDS GD_MLSL_LDS LD_MMA LN_COBYLA LN_NEWUOA LN_NEWUOA_BOUND LN_NELDERMEAD LN_SBPLX LN_AUGLAG "
               "LD_AUGLAG LN_AUGLAG_EQ LD_AUGLAG_EQ LN_BOBYQA GN_ISRES AUGLAG AUGLAG_EQ G_MLSL G_MLSL_LDS LD_SLSQP "
               "LD_CCSAQ GN_ESCH GN_AGS").split()
for _i, _name in enumerate(_ALGORITHMS):
    globals()[_name] = _i
NUM_ALGORITHMS = len(_ALGORITHMS)
FAILURE, INVALID_ARGS, OUT_OF_MEMORY, ROUNDOFF_LIMITED, FORCED_STOP, NUM_FAILURES = -1, -2, -3, -4, -5, -6
SUCCESS, STOPVAL_REACHED, FTOL_REACHED, XTOL_REACHED, MAXEVAL_REACHED, MAXTIME_REACHED, NUM_RESULTS = 1, 2, 3, 4, 5, 6, 7


# ---- exceptions (nlopt-python.i:6-50, nlopt-in.hpp:71-95) ------------------------------------------------------------
class ForcedStop(Exception):
    """Python version of nlopt::forced_stop exception."""


class RoundoffLimited(Exception):
    """Python version of nlopt::roundoff_limited exception."""


class invalid_argument(ValueError):
    """std::invalid_argument as the SWIG module exposes it; also a ValueError"""


class runtime_error(RuntimeError):
    """std::runtime_error as the SWIG module exposes it; also a RuntimeError"""


# ---- the library ------------------------------------------------------------------------------------------------------
_vp, _dbl, _dpp = _C.c_void_p, _C.c_double, _C.POINTER(_C.c_double)
_FUNC = _C.CFUNCTYPE(_C.c_double, _C.c_uint, _dpp, _dpp, _vp)
_MFUNC = _C.CFUNCTYPE(None, _C.c_uint, _dpp, _C.c_uint, _dpp, _dpp, _vp)
_lib = None


def _bind(L):
    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, list(args)
    ci, cu = _C.c_int, _C.c_uint
    sig("nlopt_create", _vp, ci, cu)
    sig("nlopt_copy", _vp, _vp)
    sig("nlopt_destroy", None, _vp)
    sig("nlopt_optimize", ci, _vp, _dpp, _dpp)
    for nm in ("min_objective", "max_objective"):
        sig("nlopt_set_" + nm, ci, _vp, _vp, _vp)
    for nm in ("lower_bounds", "upper_bounds", "xtol_abs", "x_weights"):
        sig("nlopt_set_" + nm, ci, _vp, _dpp)
        sig("nlopt_get_" + nm, ci, _vp, _dpp)
        sig("nlopt_set_" + nm + "1", ci, _vp, _dbl)
    sig("nlopt_set_initial_step", ci, _vp, _dpp)
    sig("nlopt_set_initial_step1", ci, _vp, _dbl)
    sig("nlopt_get_initial_step", ci, _vp, _dpp, _dpp)
    sig("nlopt_set_default_initial_step", ci, _vp, _dpp)
    for nm in ("stopval", "ftol_rel", "ftol_abs", "xtol_rel", "maxtime"):
        sig("nlopt_set_" + nm, ci, _vp, _dbl)
        sig("nlopt_get_" + nm, _dbl, _vp)
    for nm in ("maxeval", "force_stop"):
        sig("nlopt_set_" + nm, ci, _vp, ci)
        sig("nlopt_get_" + nm, ci, _vp)
    for nm in ("population", "vector_storage"):
        sig("nlopt_set_" + nm, ci, _vp, cu)
        sig("nlopt_get_" + nm, cu, _vp)
    sig("nlopt_get_numevals", ci, _vp)
    sig("nlopt_get_algorithm", ci, _vp)
    sig("nlopt_get_dimension", cu, _vp)
    sig("nlopt_get_errmsg", _C.c_char_p, _vp)
    sig("nlopt_set_local_optimizer", ci, _vp, _vp)
    sig("nlopt_set_param", ci, _vp, _C.c_char_p, _dbl)
    sig("nlopt_get_param", _dbl, _vp, _C.c_char_p, _dbl)
    sig("nlopt_has_param", ci, _vp, _C.c_char_p)
    sig("nlopt_num_params", cu, _vp)
    sig("nlopt_nth_param", _C.c_char_p, _vp, cu)
    for nm in ("inequality", "equality"):
        sig("nlopt_add_%s_constraint" % nm, ci, _vp, _vp, _vp, _dbl)
        sig("nlopt_add_%s_mconstraint" % nm, ci, _vp, cu, _vp, _vp, _dpp)
        sig("nlopt_remove_%s_constraints" % nm, ci, _vp)
    sig("nlopt_srand", None, _C.c_ulong)
    sig("nlopt_srand_time", None)
    sig("nlopt_version", None, _C.POINTER(ci), _C.POINTER(ci), _C.POINTER(ci))
    sig("nlopt_algorithm_name", _C.c_char_p, ci)
    sig("nlopt_algorithm_from_string", ci, _C.c_char_p)
    return L


def _library():
    global _lib
    if _lib is None:
        path = _os.environ.get("NLOPT_AMD_PYAPI_LIBRARY")
        if path:
            _lib = _bind(_C.CDLL(path))
        else:
            import nlopt_amd
            nlopt_amd.lib()                  # raises if the library has not been built
            _lib = _bind(_C.CDLL(nlopt_amd.LIB_PATH))   # a handle of its own: the prototypes set here stay private
    return _lib


def srand(seed):
    _library().nlopt_srand(int(seed))


def srand_time():
    _library().nlopt_srand_time()


def _version():
    v = [_C.c_int(), _C.c_int(), _C.c_int()]
    _library().nlopt_version(*[_C.byref(c) for c in v])
    return tuple(c.value for c in v)


def version_major():
    return _version()[0]


def version_minor():
    return _version()[1]


def version_bugfix():
    return _version()[2]


def algorithm_name(a):
    s = _library().nlopt_algorithm_name(int(a))
    return s.decode() if s is not None else None


def device_objective(name_or_id):
    """handle of a registered device objective of libnlopt_amd (include/nlopt_amd.h:28-41) for set_min/max_objective"""
    from nlopt_amd import OBJECTIVES
    L = _library()
    if not hasattr(L, "nlopt_amd_objective"):
        raise runtime_error("the library behind this module has no device objectives")
    L.nlopt_amd_objective.restype, L.nlopt_amd_objective.argtypes = _vp, [_C.c_int]
    p = L.nlopt_amd_objective(OBJECTIVES[name_or_id] if isinstance(name_or_id, str) else int(name_or_id))
    if not p:
        raise invalid_argument("no such device objective: %r" % (name_or_id,))
    return _DeviceObjective(p)


class _DeviceObjective:
    def __init__(self, pointer):
        self.pointer = pointer


def __getattr__(name):                       # nlopt.__version__ needs the library: resolve it on first use
    if name == "__version__":
        return "%d.%d.%d" % _version()
    raise AttributeError("module 'nlopt' has no attribute %r" % name)


# ---- callbacks --------------------------------------------------------------------------------------------------------
_running = _threading.local()                # the stack of opt objects inside optimize() on this thread


def _abort_run(exc):
    """an exception left a callback: remember it for optimize() to re-raise and stop the run gracefully
    (nlopt-in.hpp:160-172 — forced_stop_reason + force_stop; nlopt-python.i:162-165 keeps the Python error pending)"""
    stack = getattr(_running, "stack", None)
    if stack:
        o = stack[-1]
        if o._pending is None:
            o._pending = exc
        _library().nlopt_set_force_stop(o._o, 1)


def _as_value(r):
    if isinstance(r, float):
        return float(r)
    if isinstance(r, int):
        if r < 0:
            raise OverflowError("can't convert negative value to unsigned int")
        return float(r)
    raise _InvalidResult("invalid result passed to nlopt")


class _InvalidResult(Exception):
    """the callback returned something that is neither float nor int (nlopt-python.i:174-177)"""


def _func_thunk(f):
    def call(n, x, grad, _data):
        try:
            xa = _np.ctypeslib.as_array(x, shape=(n,)) if n else _np.empty(0)
            xa.setflags(write=False)
            ga = _np.ctypeslib.as_array(grad, shape=(n,)) if (grad and n) else _np.empty(0)
            return _as_value(f(xa, ga))
        except BaseException as e:           # noqa: BLE001 - everything must stay on this side of the C frames
            _abort_run(e)
            return float("inf")
    return _FUNC(call)


def _mfunc_thunk(f):
    def call(m, result, n, x, grad, _data):
        try:
            ra = _np.ctypeslib.as_array(result, shape=(m,)) if m else _np.empty(0)
            xa = _np.ctypeslib.as_array(x, shape=(n,)) if n else _np.empty(0)
            xa.setflags(write=False)
            ga = _np.ctypeslib.as_array(grad, shape=(m, n)) if (grad and m and n) else _np.empty(0)
            f(ra, xa, ga)
        except BaseException as e:           # noqa: BLE001
            _abort_run(e)
            for i in range(m):
                result[i] = float("inf")
    return _MFUNC(call)


def _vector(v, what="argument"):
    try:
        a = _np.ascontiguousarray(v, dtype=_np.float64)
    except (TypeError, ValueError):
        raise TypeError("%s must be a sequence of numbers" % what) from None
    if a.ndim != 1:
        raise TypeError("%s must be one-dimensional" % what)
    return a


def _ptr(a):
    return a.ctypes.data_as(_dpp) if a.size else None


class opt:
    """nlopt::opt (src/api/nlopt-in.hpp:83-612) as its SWIG Python proxy presents it"""

    def __init__(self, *args):
        self._L = _library()
        self._o = None
        self._objective = None               # (thunk, callable): keeps the ctypes callback alive
        self._ineq, self._eq = [], []
        self._local = None
        self._exceptions_enabled = True
        self._last_result, self._last_optf = FAILURE, float("inf")
        self._pending = None
        if not args:
            return
        if len(args) == 1 and isinstance(args[0], opt):
            src = args[0]
            if src._o is not None:
                self._o = self._L.nlopt_copy(src._o)
                if not self._o:
                    raise MemoryError("std::bad_alloc")
            self._objective, self._ineq, self._eq, self._local = src._objective, list(src._ineq), list(src._eq), src._local
            self._exceptions_enabled = src._exceptions_enabled
            self._last_result, self._last_optf = src._last_result, src._last_optf
            return
        if len(args) != 2:
            raise TypeError("opt(algorithm, n), opt(name, n), opt(other) or opt()")
        a, n = args
        if isinstance(a, str):
            a = self._L.nlopt_algorithm_from_string(a.encode())
            if a < 0:
                raise invalid_argument("wrong algorithm string")
        if int(n) < 0:
            raise OverflowError("can't convert negative value to unsigned int")
        self._o = self._L.nlopt_create(int(a), int(n))
        if not self._o:
            raise MemoryError("std::bad_alloc")

    def __del__(self):
        o, self._o = getattr(self, "_o", None), None
        if o:
            self._L.nlopt_destroy(o)

    # -- error mapping (nlopt-in.hpp:87-95) --
    def _need(self):
        if self._o is None:
            raise runtime_error("uninitialized nlopt::opt")
        return self._o

    def _throw(self, ret):
        if ret == FAILURE:
            raise runtime_error(self.get_errmsg() or "nlopt failure")
        if ret == OUT_OF_MEMORY:
            raise MemoryError("std::bad_alloc")
        if ret == INVALID_ARGS:
            raise invalid_argument(self.get_errmsg() or "nlopt invalid argument")
        if ret == ROUNDOFF_LIMITED:
            raise RoundoffLimited("NLopt roundoff-limited")
        if ret == FORCED_STOP:
            raise ForcedStop("NLopt forced stop")

    def _dim_check(self, a):
        if self._o is not None and self._L.nlopt_get_dimension(self._o) != a.size:
            raise invalid_argument("dimension mismatch")

    # -- run --
    def optimize(self, x0):
        x = _vector(x0, "x0").copy()
        self._dim_check(x)
        minf = _C.c_double(self._last_optf)
        self._pending = None
        stack = _running.__dict__.setdefault("stack", [])
        stack.append(self)
        try:
            ret = self._L.nlopt_optimize(self._o, _ptr(x), _C.byref(minf))
        finally:
            stack.pop()
        self._last_result, self._last_optf = ret, minf.value
        pending, self._pending = self._pending, None
        if isinstance(pending, _InvalidResult):
            # a C++ exception of func_python, not a Python error: mythrow(forced_stop_reason) (nlopt-in.hpp:323-325)
            if self._exceptions_enabled and ret == FORCED_STOP:
                raise invalid_argument(self.get_errmsg() or "nlopt invalid argument")
        elif pending is not None:
            raise pending
        if self._exceptions_enabled:
            self._throw(ret)
        return x

    def last_optimize_result(self):
        return self._last_result

    def last_optimum_value(self):
        return self._last_optf

    # -- accessors --
    def get_algorithm(self):
        return self._L.nlopt_get_algorithm(self._need())

    def get_algorithm_name(self):
        return algorithm_name(self._L.nlopt_get_algorithm(self._need()))

    def get_dimension(self):
        return self._L.nlopt_get_dimension(self._need())

    def get_numevals(self):
        return self._L.nlopt_get_numevals(self._need())

    def get_errmsg(self):
        m = self._L.nlopt_get_errmsg(self._need())
        return m.decode() if m is not None else None

    # -- objective --
    def _set_objective(self, setter, f):
        if isinstance(f, _DeviceObjective):
            self._throw(setter(self._o, f.pointer, None))
            self._objective = (None, f)
            return
        if not callable(f):
            raise TypeError("the objective must be callable")
        thunk = _func_thunk(f)
        self._throw(setter(self._o, _C.cast(thunk, _vp), None))
        self._objective = (thunk, f)

    def set_min_objective(self, f):
        self._set_objective(self._L.nlopt_set_min_objective, f)

    def set_max_objective(self, f):
        self._set_objective(self._L.nlopt_set_max_objective, f)

    # -- nonlinear constraints --
    def _add_constraint(self, adder, keep, fc, tol):
        if not callable(fc):
            raise TypeError("the constraint must be callable")
        thunk = _func_thunk(fc)
        self._throw(adder(self._o, _C.cast(thunk, _vp), None, float(tol)))
        keep.append((thunk, fc))

    def _add_mconstraint(self, adder, keep, fc, tol):
        if not callable(fc):
            raise TypeError("the constraint must be callable")
        t = _vector(tol, "tol")
        thunk = _mfunc_thunk(fc)
        self._throw(adder(self._o, t.size, _C.cast(thunk, _vp), None, _ptr(t)))
        keep.append((thunk, fc))

    def add_inequality_constraint(self, fc, tol=0.0):
        self._add_constraint(self._L.nlopt_add_inequality_constraint, self._ineq, fc, tol)

    def add_equality_constraint(self, h, tol=0.0):
        self._add_constraint(self._L.nlopt_add_equality_constraint, self._eq, h, tol)

    def add_inequality_mconstraint(self, fc, tol):
        self._add_mconstraint(self._L.nlopt_add_inequality_mconstraint, self._ineq, fc, tol)

    def add_equality_mconstraint(self, h, tol):
        self._add_mconstraint(self._L.nlopt_add_equality_mconstraint, self._eq, h, tol)

    def remove_inequality_constraints(self):
        self._throw(self._L.nlopt_remove_inequality_constraints(self._o))
        self._ineq = []

    def remove_equality_constraints(self):
        self._throw(self._L.nlopt_remove_equality_constraints(self._o))
        self._eq = []

    # -- string-keyed parameters --
    def set_param(self, name, val):
        self._throw(self._L.nlopt_set_param(self._o, name.encode(), float(val)))

    def get_param(self, name, defaultval):
        return self._L.nlopt_get_param(self._o, name.encode(), float(defaultval))

    def has_param(self, name):
        return bool(self._L.nlopt_has_param(self._o, name.encode()))

    def nth_param(self, n):
        s = self._L.nlopt_nth_param(self._o, int(n))
        return s.decode() if s is not None else None

    def num_params(self):
        return self._L.nlopt_num_params(self._o)

    # -- vectors: set_X(scalar | sequence), get_X() (NLOPT_GETSET_VEC, nlopt-in.hpp:531-553) --
    def _set_vec(self, name, v):
        if isinstance(v, (int, float, _np.floating, _np.integer)) and not isinstance(v, bool):
            self._throw(getattr(self._L, "nlopt_set_%s1" % name)(self._o, float(v)))
            return
        a = _vector(v, name)
        self._dim_check(a)
        self._throw(getattr(self._L, "nlopt_set_" + name)(self._o, _ptr(a)))

    def _get_vec(self, name):
        a = _np.empty(self._L.nlopt_get_dimension(self._need()))
        self._throw(getattr(self._L, "nlopt_get_" + name)(self._o, _ptr(a)))
        return a

    def set_lower_bounds(self, v): self._set_vec("lower_bounds", v)
    def get_lower_bounds(self): return self._get_vec("lower_bounds")
    def set_upper_bounds(self, v): self._set_vec("upper_bounds", v)
    def get_upper_bounds(self): return self._get_vec("upper_bounds")
    def set_xtol_abs(self, v): self._set_vec("xtol_abs", v)
    def get_xtol_abs(self): return self._get_vec("xtol_abs")
    def set_x_weights(self, v): self._set_vec("x_weights", v)
    def get_x_weights(self): return self._get_vec("x_weights")
    def set_initial_step(self, v): self._set_vec("initial_step", v)

    def get_initial_step(self, x):
        """the initial step the library would take from x (get_initial_step_, nlopt-in.hpp:600-605)"""
        xa = _vector(x, "x")
        dx = _np.empty(self._L.nlopt_get_dimension(self._need()))
        if xa.size != dx.size:
            raise invalid_argument("dimension mismatch")
        self._throw(self._L.nlopt_get_initial_step(self._o, _ptr(xa), _ptr(dx)))
        return dx

    def set_default_initial_step(self, x):
        xa = _vector(x, "x")
        self._throw(self._L.nlopt_set_default_initial_step(self._o, _ptr(xa)))

    # -- scalars (NLOPT_GETSET, nlopt-in.hpp:560-568) --
    def set_stopval(self, v): self._throw(self._L.nlopt_set_stopval(self._o, float(v)))
    def get_stopval(self): return self._L.nlopt_get_stopval(self._need())
    def set_ftol_rel(self, v): self._throw(self._L.nlopt_set_ftol_rel(self._o, float(v)))
    def get_ftol_rel(self): return self._L.nlopt_get_ftol_rel(self._need())
    def set_ftol_abs(self, v): self._throw(self._L.nlopt_set_ftol_abs(self._o, float(v)))
    def get_ftol_abs(self): return self._L.nlopt_get_ftol_abs(self._need())
    def set_xtol_rel(self, v): self._throw(self._L.nlopt_set_xtol_rel(self._o, float(v)))
    def get_xtol_rel(self): return self._L.nlopt_get_xtol_rel(self._need())
    def set_maxeval(self, v): self._throw(self._L.nlopt_set_maxeval(self._o, int(v)))
    def get_maxeval(self): return self._L.nlopt_get_maxeval(self._need())
    def set_maxtime(self, v): self._throw(self._L.nlopt_set_maxtime(self._o, float(v)))
    def get_maxtime(self): return self._L.nlopt_get_maxtime(self._need())
    def set_force_stop(self, v): self._throw(self._L.nlopt_set_force_stop(self._o, int(v)))
    def get_force_stop(self): return self._L.nlopt_get_force_stop(self._need())
    def force_stop(self): self.set_force_stop(1)
    def set_population(self, v): self._throw(self._L.nlopt_set_population(self._o, int(v)))
    def get_population(self): return self._L.nlopt_get_population(self._need())
    def set_vector_storage(self, v): self._throw(self._L.nlopt_set_vector_storage(self._o, int(v)))
    def get_vector_storage(self): return self._L.nlopt_get_vector_storage(self._need())

    def set_local_optimizer(self, lo):
        if not isinstance(lo, opt):
            raise TypeError("set_local_optimizer takes an nlopt.opt")
        self._throw(self._L.nlopt_set_local_optimizer(self._o, lo._o))

    # -- exceptions in optimize (nlopt-in.hpp:608-609) --
    def get_exceptions_enabled(self): return self._exceptions_enabled
    def set_exceptions_enabled(self, enable): self._exceptions_enabled = bool(enable)
