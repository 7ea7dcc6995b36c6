#!/bin/bash
# Round 5, GPU call 5: new chain-kernel shapes below n = 2048, the resolver's fast step (A/B against a build without it), the doorbell with
# ONE system-scope fence (A/B against the stream synchronisation), the ISRES start predictor from the parents' expected redraws.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c5; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/test_gpu_crs.py tests/test_gpu_crs_windows.py tests/test_gpu_kernels.py tests/test_gpu_isres.py tests/test_gpu_fullsize.py tests/test_gpu_nan.py tests/test_gpu_stops.py -x -q -m gpu -k "not mlsl" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    hs = d.get("host_split") or {}
    r = d.get("roofline") or {}
    w = d.get("window") or {}
    ph = d.get("phases") or {}
    if ph:
        print("%-44s %9.0f evals/s  %8.3f ms/step  %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") or k.endswith("_s_per_iter") else v) for k, v in ph.items()}))
    else:
        print("%-44s %9.0f evals/s  %8.3f ms/step  frac %.4f  engine %.4f walk %.4f kernel(sampled) %.4f s / %s passes; avg launch %.1f us; slots %s used %s" % (
            sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, hs.get("engine_s", 0), hs.get("walk_s", 0),
            hs.get("gather_kernel_s", 0), hs.get("passes"), 1e3 * (r.get("avg_launch_ms") or 0), w.get("slots_started"), w.get("slots_used")))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
NB="--headline-only --no-cpu-baseline --obj rastrigin --steps 4 --warmup 1 --evals-per-step 20000"
for rep in 1 2; do
for n in 64 128 256 512 1024; do
  line "n=$n default (fast step, light doorbell)"  --n $n $NB
  line "n=$n amd_doorbell=0"                       --n $n $NB --param amd_doorbell=0
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_nofast.so line "n=$n no fast step, amd_doorbell=0" --n $n $NB --param amd_doorbell=0
done
line "isres config 3" --workload isres --no-cpu-baseline --steps 3 --warmup 1
done
line "headline default" --headline-only --no-cpu-baseline --steps 10 --warmup 2
for n in 64 512; do timeout -k 5 120 python tools/chain_latency.py $n 100000 > $O/chain_latency_n$n.txt 2>&1; tail -13 $O/chain_latency_n$n.txt; done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
