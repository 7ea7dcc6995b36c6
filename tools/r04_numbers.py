"""Format the round's measured numbers (a `python bench.py` line) as the table of DESIGN.md section 5.
usage: python tools/r04_numbers.py <bench.json>"""
import json
import sys


def main():
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    r, cb = d["roofline"], d["cpu_baseline"]
    ow = d.get("other_workloads", {})
    osz = d.get("other_sizes", {})
    e2e = d.get("nlopt_optimize_end_to_end", {})
    g = d.get("gens_to_ftol", {})
    out = []
    out.append("| workload | GPU | CPU reference (1 thread, same box) | ratio | dominant kernel vs its roofline |")
    out.append("|---|---|---|---|---|")
    out.append("| CRS2_LM Griewank n=4096 pop=1e5 (metric) | **%.1f k evals/s**, %.1f ms per 2000-eval step%s | %.0f evals/s | **%.0f×** (target ≥ 50×) | `crs_chain_kernel` %.2f GB per launch in %.3f ms = %.2f TB/s = **%.3f** of the 8 TB/s peak; **%.3f counting only consumed trials** (%.1f of 128 slots per launch); %.3f of the ≈ 6.3 TB/s a plain stream reaches |"
               % (d["value"] / 1e3, d["ms_per_step"], ("; one `nlopt_optimize` call incl. init (%d evals): %.2f s" % (e2e.get("numevals", 0), e2e.get("wall_s", 0))) if e2e.get("wall_s") else "",
                  cb["value"], d["value"] / cb["value"], r["avg_algorithmic_bytes_per_launch"] / 1e9, r["avg_launch_ms"], r["achieved"] / 1e3, r["frac"],
                  r.get("frac_useful", 0), r.get("avg_trials_consumed_per_launch", 0), r.get("frac_of_achievable", 0)))
    for key, label in (("n=512", "CRS2_LM Rastrigin n=512 pop=1e5 (config 2)"), ("n=64", "CRS2_LM Rastrigin n=64 pop=1e5")):
        v = osz.get(key)
        if isinstance(v, dict) and "value" in v:
            c = v.get("cpu_baseline", {})
            out.append("| %s | **%.0f k evals/s** | %.1f k | %.0f× | %.3f (conservative passes: pass latency) |" % (label, v["value"] / 1e3, c.get("value", 0) / 1e3, v["value"] / max(c.get("value", 1), 1), v.get("roofline_frac") or 0))
    i = ow.get("isres")
    if i and "value" in i:
        c = i.get("cpu_baseline", {})
        ph = i.get("phases", {})
        out.append("| ISRES Rastrigin n=256 pop=5e4, 4 constraints (config 3) | **%.0f k evals/s = %.1f ms per generation** (rank %.1f, evolve %.1f, eval %.2f ms) | **%.0f evals/s at the benchmark's own population**: one whole generation of the real reference in %.1f s, timed in the bench run | **%.0f×** | `isres_stochrank_kernel` (latency-bound): %.0f ns per serial tick vs the 43 ns floor of its 26 instructions |"
                   % (i["value"] / 1e3, i["ms_per_step"], 1e3 * ph.get("rank_s_per_gen", 0), 1e3 * ph.get("evolve_s_per_gen", 0), 1e3 * ph.get("eval_s_per_gen", 0), c.get("value", 0), c.get("seconds_per_generation", 0),
                      i["value"] / max(c.get("value", 1), 1e-9), i["roofline"].get("achieved") or 0))
    for key, label in (("mlsl", "default mode: workgroup tree sums"), ("mlsl_exact_order", "`amd_exact_dot` = 1: the reference's summation order")):
        m = ow.get(key)
        if m and "value" in m:
            c = (ow.get("mlsl") or {}).get("cpu_baseline", {})
            ph, rr = m.get("phases", {}), m["roofline"]
            out.append("| G_MLSL_LDS+LD_LBFGS Ackley n=4096, 1000 samples (config 4), %s | **%.0f k evals/s**, %.1f ms per iteration (sampling %.1f, local phase %.1f) | %.1f k | %.0f× | `%s` %.2f GB per launch in %.2f ms = %.2f TB/s = **%.3f**%s |"
                       % (label, m["value"] / 1e3, m["ms_per_step"], 1e3 * ph.get("sampling_s_per_iter", 0), 1e3 * ph.get("local_phase_s_per_iter", 0), c.get("value", 0) / 1e3,
                          m["value"] / max(c.get("value", 1), 1e-9), rr["kernel"], rr["avg_algorithmic_bytes_per_launch"] / 1e9, rr["avg_launch_ms"], rr["achieved"] / 1e3, rr["frac"],
                          ("; PMC traffic %.2f GB per launch = %.2f × algorithmic" % (rr["traffic"] / 1e9, rr["traffic"] / rr["avg_algorithmic_bytes_per_launch"])) if rr.get("traffic") else ""))
    if g:
        s2 = g.get("second_pin", {})
        out.append("| gens-to-ftol, both of BASELINE.md's pins | n=10 pop=100 ftol 1e-4: %.2f; n=64 pop=2000 ftol 1e-6: %.2f (%d evals, %.2f s) | %.2f; %.2f | identical: %s / %s | — |"
                   % (g.get("value", 0), s2.get("value", 0), s2.get("numevals", 0), s2.get("wall_s", 0), g.get("reference_value", 0), s2.get("reference_value", 0), g.get("identical_to_reference"), s2.get("identical_to_reference")))
    print("\n".join(out))


if __name__ == "__main__":
    main()
