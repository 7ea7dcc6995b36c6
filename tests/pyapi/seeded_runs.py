"""A client of the reference's Python module, written against `import nlopt` only: every algorithm of the path, each from a
fixed seed, plus the interface's corner cases.  tests/test_python_module.py runs it over the real reference library and over
libnlopt_amd and requires the two printouts to be identical."""
import math
import sys

import numpy as np

import nlopt

algs = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else \
    [nlopt.GN_CRS2_LM, nlopt.GN_ISRES, nlopt.GN_ESCH, nlopt.GN_MLSL, nlopt.GD_MLSL, nlopt.GN_MLSL_LDS, nlopt.GD_MLSL_LDS,
     nlopt.G_MLSL, nlopt.G_MLSL_LDS, nlopt.LD_LBFGS, nlopt.LD_MMA, nlopt.LN_COBYLA]
count = [0]


def powell(x, grad):
    count[0] += 1
    x1, x2 = x[0], x[1]
    f1 = 10000. * x1 * x2 - 1.
    f2 = math.exp(-x1) + math.exp(-x2) - 1.0001
    if grad.size > 0:
        grad[0] = 2.0 * f1 * 10000. * x2 - 2.0 * f2 * math.exp(-x1)
        grad[1] = 2.0 * f1 * 10000. * x1 - 2.0 * f2 * math.exp(-x2)
    return f1 * f1 + f2 * f2


def bowl(x, grad):
    count[0] += 1
    c = np.arange(1, x.size + 1) * 0.3
    if grad.size > 0:
        grad[:] = 2 * (x - c) - 0.4 * np.sin(4 * x)
    return float(np.sum((x - c) ** 2) + 0.1 * np.sum(np.cos(4 * x)))


def show(tag, opt, x):
    print(tag, [repr(float(v)) for v in x], repr(opt.last_optimum_value()), opt.last_optimize_result(), opt.get_numevals(),
          count[0])


print("version", nlopt.version_major(), nlopt.version_minor(), nlopt.version_bugfix(), nlopt.__version__)
print("constants", nlopt.NUM_ALGORITHMS, nlopt.GN_CRS2_LM, nlopt.GN_ISRES, nlopt.GN_ESCH, nlopt.G_MLSL_LDS, nlopt.LN_COBYLA,
      nlopt.FORCED_STOP, nlopt.MAXTIME_REACHED)
for a in algs:
    print("name", a, nlopt.algorithm_name(a))

# 1. every algorithm of the path on a smooth 3-d problem and on Powell's badly scaled function, each from a fixed seed
for a in algs:
    for fn, n, lo, hi, x0 in ((bowl, 3, -2.0, 3.0, [0.5, -1.0, 2.5]), (powell, 2, -10.0, 10.0, [0.0, 0.0])):
        nlopt.srand(1000 + a)
        count[0] = 0
        o = nlopt.opt(a, n)
        o.set_min_objective(fn)
        o.set_lower_bounds(lo)
        o.set_upper_bounds([hi] * n)
        o.set_maxeval(600)
        o.set_xtol_rel(1e-6)
        if a in (nlopt.G_MLSL, nlopt.G_MLSL_LDS):
            lo_ = nlopt.opt(nlopt.LD_LBFGS, n)
            lo_.set_ftol_rel(1e-9)
            o.set_local_optimizer(lo_)
        try:
            x = o.optimize(x0)
            show("run %d %s" % (a, fn.__name__), o, x)
        except Exception as e:                      # noqa: BLE001
            print("run %d %s raised" % (a, fn.__name__), type(e).__name__, e)

# 2. constraints: ISRES with scalar constraints, COBYLA with a vector constraint; maximisation; a copy runs like the original
for a in (nlopt.GN_ISRES, nlopt.LN_COBYLA):
    if a not in algs:
        continue
    nlopt.srand(77)
    count[0] = 0
    o = nlopt.opt(a, 3)
    o.set_max_objective(lambda x, g: -bowl(x, g))
    o.set_lower_bounds([-2.0, -2.0, -2.0])
    o.set_upper_bounds(3.0)
    o.add_inequality_constraint(lambda x, g: float(x[0] + x[1] - 1.0), 1e-8)
    o.add_equality_constraint(lambda x, g: float(x[2] - 0.5 * x[0] - 0.25), 1e-6)
    if a == nlopt.LN_COBYLA:
        def vc(result, x, grad):
            result[0] = x[0] * x[0] - 1.5
            result[1] = -x[1] - 1.0
        o.add_inequality_mconstraint(vc, [1e-8, 1e-8])
        o.set_initial_step([0.3, 0.2, 0.1])
    o.set_maxeval(700)
    o.set_population(24)
    dup = nlopt.opt(o)
    show("constrained %d" % a, o, o.optimize([0.1, 0.2, 0.3]))
    nlopt.srand(77)
    count[0] = 0
    show("constrained %d copy" % a, dup, dup.optimize([0.1, 0.2, 0.3]))
    o.remove_inequality_constraints()
    o.remove_equality_constraints()
    nlopt.srand(78)
    show("unconstrained %d" % a, o, o.optimize([0.1, 0.2, 0.3]))

# 3. getters, setters, parameters
o = nlopt.opt("GN_CRS2_LM", 4)
o.set_lower_bounds([-1, -2, -3, -4])
o.set_upper_bounds(5)
o.set_xtol_abs([1e-3, 1e-4, 1e-5, 1e-6])
o.set_x_weights(2.0)
o.set_stopval(-3.5)
o.set_ftol_rel(1e-3)
o.set_ftol_abs(1e-5)
o.set_xtol_rel(1e-7)
o.set_maxeval(123)
o.set_maxtime(4.5)
o.set_population(55)
o.set_vector_storage(7)
o.set_param("inner_maxeval", 9)
print("get", o.get_algorithm(), o.get_algorithm_name(), o.get_dimension(), list(o.get_lower_bounds()), list(o.get_upper_bounds()),
      list(o.get_xtol_abs()), list(o.get_x_weights()), o.get_stopval(), o.get_ftol_rel(), o.get_ftol_abs(), o.get_xtol_rel(),
      o.get_maxeval(), o.get_maxtime(), o.get_population(), o.get_vector_storage(), o.get_force_stop(), o.get_numevals(),
      o.get_param("inner_maxeval", 1), o.get_param("nothing", 1.5), o.has_param("inner_maxeval"), o.has_param("x"),
      o.num_params(), o.nth_param(0), o.get_errmsg(), o.get_exceptions_enabled())
print("initial step", list(o.get_initial_step([0.0, 0.0, 0.0, 0.0])))
o.set_initial_step(0.25)
print("initial step", list(o.get_initial_step([0.0, 0.0, 0.0, 0.0])))
o.set_default_initial_step([1.0, 1.0, 1.0, 1.0])
print("initial step", list(o.get_initial_step([1.0, 1.0, 1.0, 1.0])))

# 4. errors
for what, call in (("dimension", lambda: o.set_lower_bounds([0.0, 1.0])),
                   ("negative tolerance", lambda: o.set_xtol_abs([-1.0, 0, 0, 0]) or o.get_errmsg()),
                   ("wrong name", lambda: nlopt.opt("NO_SUCH_ALGORITHM", 2)),
                   ("x0 size", lambda: o.optimize([0.0, 0.0])),
                   ("no objective", lambda: o.optimize([0.0, 0.0, 0.0, 0.0])),
                   ("uninitialised get", lambda: nlopt.opt().get_dimension()),
                   ("constraint refused", lambda: o.add_inequality_constraint(lambda x, g: 0.0)),
                   ("bad param", lambda: o.set_param(None, 1.0) if False else o.set_maxeval("many"))):
    try:
        print("error", what, "->", call())
    except Exception as e:                          # noqa: BLE001
        print("error", what, "->", type(e).__name__, "|", isinstance(e, ValueError), isinstance(e, RuntimeError))


class Boom(Exception):
    pass


def explode(x, grad):
    count[0] += 1
    if count[0] == 40:
        raise Boom("evaluation 40")
    return float(np.sum(x * x))


def wrong_type(x, grad):
    count[0] += 1
    return "seven" if count[0] == 25 else float(np.sum(x * x))


def integer_valued(x, grad):
    count[0] += 1
    return int(round(float(np.sum(x * x)) * 100))


for a in (nlopt.GN_CRS2_LM, nlopt.GN_ISRES, nlopt.LN_COBYLA):
    if a not in algs:
        continue
    for fn in (explode, wrong_type, integer_valued):
        for enabled in (True, False):
            nlopt.srand(5)
            count[0] = 0
            q = nlopt.opt(a, 2)
            q.set_exceptions_enabled(enabled)
            q.set_min_objective(fn)
            q.set_lower_bounds(-1.0)
            q.set_upper_bounds(2.0)
            q.set_maxeval(150)
            try:
                x = q.optimize([0.5, 0.5])
                show("callback %d %s %s" % (a, fn.__name__, enabled), q, x)
            except Exception as e:                  # noqa: BLE001
                print("callback %d %s %s raised" % (a, fn.__name__, enabled), type(e).__name__, e, count[0], q.last_optimize_result(),
                      q.get_force_stop())

# 5. force_stop from inside the objective, and read-only x
nlopt.srand(9)
count[0] = 0
q = nlopt.opt(nlopt.GN_CRS2_LM, 2)


def stopper(x, grad):
    count[0] += 1
    if count[0] == 1:
        try:
            x[0] = 0.0
            print("x is writable")
        except ValueError:
            print("x is read-only", grad.size)
    if count[0] == 60:
        q.force_stop()
    return float(np.sum(x * x))


q.set_min_objective(stopper)
q.set_lower_bounds(-1.0)
q.set_upper_bounds(2.0)
try:
    q.optimize([0.5, 0.5])
    print("no exception")
except nlopt.ForcedStop as e:
    print("ForcedStop", e, count[0], q.last_optimize_result(), repr(q.last_optimum_value()))
q.set_exceptions_enabled(False)
nlopt.srand(9)
count[0] = 0
show("forced stop without exceptions", q, q.optimize([0.5, 0.5]))
