#!/bin/bash
# Round 5: the default bench line once more at the last commit (another box of the pool)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_bench2; mkdir -p $O
timeout -k 5 600 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line bytes $(tail -1 $O/bench.json | wc -c)"
python - $O/bench_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline %.0f evals/s frac %.3f useful %.3f pinned %s speedup %.0f" % (d["value"], d["roofline"]["frac"], d["roofline"]["frac_useful"], (d.get("pinned_run") or {}).get("identical_to_reference"), d.get("speedup_vs_cpu_single_thread", 0)))
for k, v in (d.get("other_sizes") or {}).items():
    print(k, "%.0f evals/s" % v["value"], "frac %.3f" % (v.get("roofline_frac") or 0), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
for k, v in (d.get("other_workloads") or {}).items():
    r = v.get("roofline") or {}
    print(k, "%.0f evals/s  %.2f ms/step" % (v.get("value", 0), v.get("ms_per_step", 0)), r.get("bound"), "frac %.3f" % (r.get("frac") or 0), "avg launch %.2f ms" % (r.get("avg_launch_ms") or 0), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
PY
