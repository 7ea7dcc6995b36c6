#!/bin/bash
# Round 5, GPU call 17: the mutation phase's rounds over blocks of 512 individuals (two workgroups per compute unit in the scan, the chain
# crossing the block in two parts) against blocks of 256 (variant build); ISRES device tests; kernel statistics.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c17; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_fullsize.py tests/test_gpu_multiproc.py -x -q -m gpu -k "isres or nan or config3" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"; grep -v "^  File" $O/tests.log | grep -i "assert\|error" | head -5
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    print("%-36s %9.0f evals/s  %8.3f ms/step  pipeline %.3f ms/launch  %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("avg_launch_ms") or 0,
          {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") else v) for k, v in ph.items() if k.endswith("_s_per_gen") or "rounds" in k}))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for rep in 1 2; do
  line "isres config 3, mutation blocks of 512" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_mb256.so line "isres config 3, mutation blocks of 256" --workload isres --no-cpu-baseline --steps 3 --warmup 1
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/ki -o isres -- python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_under_rocprof.json 2> $O/ki.err
f=$(find $O/ki -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; rm -rf $O/ki
head -12 $O/isres_kernel_stats.csv
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
