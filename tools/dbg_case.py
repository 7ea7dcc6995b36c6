"""one CRS2_LM configuration on the GPU under several parameter sets against the port: first divergence of the traces (development aid)
   python tools/dbg_case.py levy 3 21 944243182 2046 [loops]"""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import nlopt_amd
import test_gpu_crs as T
obj, n, pop, seed, me = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
loops = int(sys.argv[6]) if len(sys.argv) > 6 else 3
p = O.run_port_crs(obj, n, pop, seed, trace_cap=me + 5000, maxeval=me)
tp = p["trace"]
SETS = [("default", {}), ("windows", {"amd_forward": 1}), ("windows K=8", {"amd_forward": 1, "amd_max_spec": 8}), ("conservative", {"amd_forward": 0})]
for name, params in SETS:
    for it in range(loops):
        a = T.run_amd(obj, n, pop, seed, trace_cap=me + 5000, params=params, maxeval=me)
        ta = a["trace"]
        m = min(len(ta), len(tp))
        d = np.flatnonzero((ta["row"][:m] != tp["row"][:m]) | (ta["kind"][:m] != tp["kind"][:m]) | (ta["accepted"][:m] != tp["accepted"][:m]))
        df = np.flatnonzero(np.abs(ta["f"][:m] - tp["f"][:m]) > 1e-10 * np.maximum(np.abs(tp["f"][:m]), np.abs(tp["f"]).mean()))
        st = {k: a["stats"][k] for k in ("rounds", "slots_launched", "slots_used", "slots_invalid", "slots_newbest", "slots_role", "accepted")}
        if a["nevals"] == p["nevals"] and not d.size and not df.size:
            print(name, it, "OK nevals", a["nevals"], "minf", repr(a["minf"]), st)
            continue
        i = int(min(d[0] if d.size else m, df[0] if df.size else m))
        print(name, it, "DIFF nevals", a["nevals"], p["nevals"], "first index diff", int(d[0]) if d.size else None, "first f diff", int(df[0]) if df.size else None, "of", m, st)
        for j in range(max(0, i - 3), min(m, i + 4)):
            print("   ", j, "amd", repr(float(ta["f"][j])), ta["row"][j], ta["kind"][j], ta["accepted"][j], "| port", repr(float(tp["f"][j])), tp["row"][j], tp["kind"][j], tp["accepted"][j],
                  "  df=%.3g" % (ta["f"][j] - tp["f"][j]))
