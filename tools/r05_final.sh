#!/bin/bash
# Round 5, final evidence on an MI355X box: the driver's exact GPU suite command, smoke(), the default bench line (compact + detail).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"; grep -v "^  File" $O/gpu_suite.log | grep -i "error\|fatal\|fault\|FAILED" | head -5
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-200)"
timeout -k 5 600 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line bytes $(tail -1 $O/bench.json | wc -c)"
python - $O/bench_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline %.0f evals/s frac %.3f useful %.3f pinned %s speedup %.0f" % (d["value"], d["roofline"]["frac"], d["roofline"]["frac_useful"], (d.get("pinned_run") or {}).get("identical_to_reference"), d.get("speedup_vs_cpu_single_thread", 0)))
for k, v in (d.get("other_sizes") or {}).items():
    print(k, "%.0f evals/s" % v["value"], "frac %.3f" % (v.get("roofline_frac") or 0), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
for k, v in (d.get("other_workloads") or {}).items():
    r = v.get("roofline") or {}
    print(k, "%.0f evals/s  %.2f ms/step" % (v.get("value", 0), v.get("ms_per_step", 0)), r.get("bound"), "frac %.3f" % (r.get("frac") or 0), "achieved %.2f %s" % (r.get("achieved") or 0, r.get("unit")), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
print("gens_to_ftol", d["gens_to_ftol"]["identical_to_reference"], d["gens_to_ftol"]["second_pin"]["identical_to_reference"])
PY
timeout -k 5 300 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --detail $O/bench_driver_steps_detail.json > $O/bench_driver_steps.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_driver_steps_detail.json')); print('headline with the driver\'s --steps 20 --warmup 5: %.0f evals/s frac %.3f useful %.3f pinned %s' % (d['value'], d['roofline']['frac'], d['roofline']['frac_useful'], d['pinned_run']['identical_to_reference']))"
for w in mlsl isres; do
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/k_$w -o $w -- python bench.py --workload $w --steps $([ $w = mlsl ] && echo 2 || echo 3) --warmup 1 --no-cpu-baseline > $O/bench_${w}_under_rocprof.json 2> $O/k_$w.err
  f=$(find $O/k_$w -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/${w}_kernel_stats.csv; [ $w = mlsl ] && python profiles/summarize_rocpd.py $f --timeline 0 400 > $O/mlsl_timeline.txt; rm -rf $O/k_$w
  head -6 $O/${w}_kernel_stats.csv | cut -c1-110
done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s"
