#!/bin/bash
# Round 5, GPU call 27: MLSL — the commit walk's inner loop without index arithmetic (the accepted minima's f by candidate): MLSL device
# tests, config 4 against the build before (oldwalk).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c27; mkdir -p $O
date +%s > $O/t0
timeout -k 5 600 python -X faulthandler -m pytest tests/test_gpu_mlsl.py tests/test_gpu_mlsl_short_segments.py tests/test_gpu_fullsize.py tests/test_gpu_cobyla.py tests/test_gpu_multiproc.py -x -q -m gpu -k "not crs and not isres" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    print("%-50s %9.0f evals/s  %8.3f ms/step  frac %.4f avg launch %.3f ms  %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, r.get("avg_launch_ms") or 0,
          {k: (round(v * 1e3, 2) if ("_s_per_" in k) else v) for k, v in ph.items()}))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for steps in 2 2 2 6; do
  line "mlsl config 4 ($steps steps): walk by candidate" --workload mlsl --no-cpu-baseline --steps $steps --warmup 1
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_oldwalk.so line "mlsl config 4 ($steps steps): before" --workload mlsl --no-cpu-baseline --steps $steps --warmup 1
done
line "mlsl config 4, LD_MMA (2 steps): walk by candidate" --workload mlsl --local mma --no-cpu-baseline --steps 2 --warmup 1
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_oldwalk.so line "mlsl config 4, LD_MMA (2 steps): before" --workload mlsl --local mma --no-cpu-baseline --steps 2 --warmup 1
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/km -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mlsl_under_rocprof.json 2> $O/km.err
f=$(find $O/km -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/mlsl_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 0 400 > $O/mlsl_timeline.txt; rm -rf $O/km
grep -v "copyBuffer\|fillBuffer" $O/mlsl_timeline.txt | awk -F, 'NR>2 && $1>=112 && $1<=130' | cut -c1-120
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
