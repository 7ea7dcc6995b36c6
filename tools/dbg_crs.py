"""loop the golden CRS cases on the GPU and report every run that differs from the oracle (development aid)"""
import sys, json, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _oracle as O
import nlopt_amd
import test_gpu_crs as T
G = json.load(open("tests/golden/crs_golden.json"))
loops = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ports = {}
bad = 0
for it in range(loops):
    for name in sorted(G):
        g = G[name]
        kw = dict(g["kwargs"])
        a = T.run_amd(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, **kw)
        if name not in ports:
            ports[name] = O.run_port_crs(g["obj"], g["n"], g["pop"], g["seed"], trace_cap=200000, **kw)
        p = ports[name]
        ta, tp = a["trace"], p["trace"]
        m = min(len(ta["f"]), len(tp["f"]))
        d = np.flatnonzero((ta["row"][:m] != tp["row"][:m]) | (ta["kind"][:m] != tp["kind"][:m]) | (ta["accepted"][:m] != tp["accepted"][:m]))
        if a["nevals"] != p["nevals"] or d.size:
            bad += 1
            i = int(d[0]) if d.size else m
            print("loop", it, name, "nevals", a["nevals"], p["nevals"], "first diff at", i, "of", m)
            for j in range(max(0, i - 2), min(m, i + 3)):
                print("   ", j, "amd", repr(ta["f"][j]), ta["row"][j], ta["kind"][j], ta["accepted"][j], "| port", repr(tp["f"][j]), tp["row"][j], tp["kind"][j], tp["accepted"][j])
            print("    stats", {k: a["stats"][k] for k in ("rounds", "slots_launched", "slots_used", "slots_newbest", "slots_role", "accepted")})
print("loops", loops, "bad runs", bad)
