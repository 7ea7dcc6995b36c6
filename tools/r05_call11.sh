#!/bin/bash
# Round 5, GPU call 11: the driver's suite command at the current state (everything of this session together), the default bench line,
# MLSL with the sample points inserted into the order array in one merge (timeline + HIP API trace of config 4).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c11; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"; grep -v "^  File" $O/gpu_suite.log | grep -i "error\|fatal\|fault\|FAILED" | head -5
timeout -k 5 600 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line bytes $(tail -1 $O/bench.json | wc -c)"; tail -1 $O/bench.json | cut -c1-700
python - $O/bench_detail.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("headline %.0f evals/s frac %.3f useful %.3f pinned %s" % (d["value"], d["roofline"]["frac"], d["roofline"]["frac_useful"], (d.get("pinned_run") or {}).get("identical_to_reference")))
for k, v in (d.get("other_sizes") or {}).items():
    print(k, "%.0f evals/s" % v["value"], "frac %.3f" % (v.get("roofline_frac") or 0), "speedup %.0f" % (v.get("speedup_vs_cpu_single_thread") or 0))
for k, v in (d.get("other_workloads") or {}).items():
    r = v.get("roofline") or {}
    print(k, "%.0f evals/s  %.2f ms/step" % (v.get("value", 0), v.get("ms_per_step", 0)), r.get("bound"), "frac %.3f" % (r.get("frac") or 0), "achieved %.2f %s" % (r.get("achieved") or 0, r.get("unit")))
PY
timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace -d $O/km -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mlsl_under_rocprof.json 2> $O/km.err
f=$(find $O/km -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/mlsl_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 0 400 > $O/mlsl_timeline.txt; python profiles/summarize_rocpd.py $f --api 10 > $O/mlsl_hip_api.csv; rm -rf $O/km
head -24 $O/mlsl_hip_api.csv | cut -c1-120
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s"
