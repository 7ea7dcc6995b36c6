"""development aid: LD_MMA on the device next to the CPU oracle, case by case (prints, asserts nothing)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import _oracle as O
import test_gpu_mma as T

CASES = [("sphere", 8, dict(ftol_rel=1e-10)), ("rosenbrock", 10, dict(maxeval=500)), ("rosenbrock", 10, dict(maxeval=60)), ("ackley", 30, dict(ftol_rel=1e-8)),
         ("rastrigin", 20, dict(ftol_rel=1e-8)), ("griewank", 12, dict(xtol_rel=1e-6)), ("levy", 7, dict(ftol_abs=1e-12)),
         ("ackley", 200, dict(ftol_rel=1e-8)), ("rastrigin", 64, dict(maxeval=37)), ("sphere", 6, dict(stopval=1e-3)),
         ("rastrigin", 16, dict(ftol_rel=1e-9, params=dict(inner_gradients=0))),
         ("ackley", 10, dict(ftol_rel=1e-9, params=dict(inner_maxeval=2, rho_init=0.01))),
         ("griewank", 10, dict(xtol_rel=1e-8, params=dict(sigma_min=0.5))), ("rastrigin", 12, dict(ftol_rel=1e-9, step=0.3)),
         ("ackley", 4096, dict(ftol_rel=1e-8)), ("rastrigin", 1000, dict(ftol_rel=1e-9)),
         ("rosenbrock", 6, dict(maxeval=400, params=dict(always_improve=0)))]
for obj, n, kw in CASES:
    kw = dict(kw)
    kw.setdefault("maxeval", 20000)
    a = T.run_amd(obj, n, **kw)
    p = O.run_port_mma(obj, n, **kw)
    print(obj, n, kw, "ret", a["ret"], p["ret"], "nev", a["nevals"], p["nevals"], "minf", a["minf"], p["minf"],
          "dx", float(np.abs(a["x"] - p["x"]).max()), flush=True)
