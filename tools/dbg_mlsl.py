import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, nlopt_amd, _oracle as O
import test_gpu_mlsl as T
obj,n,ns,seed,kw = "griewank", 5, 0, 3, dict(stopval=1e-7, maxeval=100000)
L = nlopt_amd.lib()
xs, lo, hi = O.golden_x0(obj, n)
o = nlopt_amd.Opt(nlopt_amd.G_MLSL, n); o.set_lower_bounds(lo); o.set_upper_bounds(hi); o.set_min_objective(nlopt_amd.objective(obj))
loc = nlopt_amd.Opt(nlopt_amd.LD_LBFGS, n); loc.set_ftol_rel(1e-8); L.nlopt_set_local_optimizer(o._h, loc._h)
o.set_maxeval(100000); o.set_stopval(1e-7); o.enable_trace(200000)
nlopt_amd.srand(seed); x, minf, ret = o.optimize_raw(xs)
t = o.trace()
p = O.run_port_mlsl(obj, n, ns, seed, **kw)
fl = t[t["kind"] == 4]
print(len(fl), len(p["floc"]))
for i in range(max(len(fl), len(p["floc"]))):
    a = fl[i] if i < len(fl) else None
    b = (p["floc"][i], p["eloc"][i]) if i < len(p["floc"]) else None
    flag = "" if (a is not None and b is not None and abs(a["f"] - b[0]) <= 1e-7 * max(1, abs(b[0]))) else "  <<<<"
    print(i, None if a is None else (a["f"], a["row"], a["accepted"]), b, flag)
