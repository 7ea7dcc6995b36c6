#!/bin/bash
# Round 5, after tools/r05_final.sh (suite 629 passed / smoke ok at 4ca6f90 + bench note): the one change since — the sampling phase is not
# enqueued ahead when the local searches sum in the reference's order (amd_exact_dot = 1: 103 -> 117 ms per launch beside the distance
# pass) — so: the MLSL / local-optimiser device tests again, and the default bench line at the final code.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final2; mkdir -p $O
date +%s > $O/t0
timeout -k 5 600 python -X faulthandler -m pytest tests/test_gpu_mlsl.py tests/test_gpu_mlsl_short_segments.py tests/test_gpu_exact_local.py tests/test_gpu_lbfgs.py tests/test_gpu_fullsize.py tests/test_gpu_stops.py -x -q -m gpu -k "not crs and not isres" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
timeout -k 5 600 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line bytes $(tail -1 $O/bench.json | wc -c)"
python - $O/bench_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline %.0f evals/s frac %.3f useful %.3f pinned %s speedup %.0f" % (d["value"], d["roofline"]["frac"], d["roofline"]["frac_useful"], (d.get("pinned_run") or {}).get("identical_to_reference"), d.get("speedup_vs_cpu_single_thread", 0)))
for k, v in (d.get("other_sizes") or {}).items():
    print(k, "%.0f evals/s" % v["value"], "frac %.3f" % (v.get("roofline_frac") or 0), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
for k, v in (d.get("other_workloads") or {}).items():
    r = v.get("roofline") or {}
    print(k, "%.0f evals/s  %.2f ms/step" % (v.get("value", 0), v.get("ms_per_step", 0)), r.get("bound"), "frac %.3f" % (r.get("frac") or 0), "achieved %.2f %s" % (r.get("achieved") or 0, r.get("unit")), "avg launch %.2f ms" % (r.get("avg_launch_ms") or 0), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
print("gens_to_ftol", d["gens_to_ftol"]["identical_to_reference"], d["gens_to_ftol"]["second_pin"]["identical_to_reference"])
PY
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s"
