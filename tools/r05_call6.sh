#!/bin/bash
# Round 5, GPU call 6: ISRES mutation phase without a stage kernel and with the previous round's write beside the next round's scan
# (hip/isres_evolve2.hip), the chain kernel without the doorbell again: ISRES + CRS2_LM device tests, config 3 lines + kernel statistics,
# the timeline of a config-4 iteration.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c6; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_crs.py tests/test_gpu_crs_windows.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not mlsl" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    hs = d.get("host_split") or {}
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    if ph:
        print("%-36s %9.0f evals/s  %8.3f ms/step  %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") or k.endswith("_s_per_iter") else v) for k, v in ph.items()}))
    else:
        print("%-36s %9.0f evals/s  %8.3f ms/step  frac %.4f  engine %.4f kernel(sampled) %.4f s / %s passes; avg launch %.1f us" % (
            sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, hs.get("engine_s", 0), hs.get("gather_kernel_s", 0), hs.get("passes"), 1e3 * (r.get("avg_launch_ms") or 0)))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
NB="--headline-only --no-cpu-baseline --obj rastrigin --steps 4 --warmup 1 --evals-per-step 20000"
for rep in 1 2; do
  line "isres config 3" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  line "n=64"  --n 64 $NB
  line "n=512" --n 512 $NB
done
line "mlsl config 4" --workload mlsl --no-cpu-baseline --steps 3 --warmup 1
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/ki -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_isres_under_rocprof.json 2> $O/ki.err
f=$(find $O/ki -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; rm -rf $O/ki
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/km -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mlsl_under_rocprof.json 2> $O/km.err
f=$(find $O/km -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/mlsl_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 0 400 > $O/mlsl_timeline.txt; rm -rf $O/km
head -14 $O/isres_kernel_stats.csv; head -12 $O/mlsl_kernel_stats.csv
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
