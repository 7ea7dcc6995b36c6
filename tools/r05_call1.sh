#!/bin/bash
# Round 5, GPU call 1: the four staged test files + every A/B round 4 left unmeasured (the driver's GPUTEST_r04 run is the green full suite at HEAD).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_first; mkdir -p $O
staged() {   # staged <log name> <limit s> <test file>
    timeout -k 5 $2 python -X faulthandler -m pytest $3 -q -m gpu -p no:cacheprovider > $O/$1.log 2>&1; echo "$3 rc=$? $(tail -1 $O/$1.log)"
}
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open("gpurun_out/r05_first/last.json"))
    hs = d.get("host_split") or {}
    print("%-58s %9.0f evals/s  %8.3f ms/step  frac %s  host: engine %.3f walk %.3f kernel %.3f s over %s passes  %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), hs.get("engine_s", 0), hs.get("walk_s", 0),
        hs.get("gather_kernel_s", 0), hs.get("passes"), d.get("phases", "")))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
date +%s > $O/t0
staged staged_resolver_small 300 tests/staged/test_gpu_chain_resolver_small.py
staged staged_isres 240 tests/staged/test_gpu_isres_rank_prefetch.py
staged staged_isres_scan 300 tests/staged/test_gpu_isres_fast_scan.py
staged staged_mlsl_seg 300 tests/staged/test_gpu_mlsl_short_segments.py
line "isres config 3 default"                      --workload isres --no-cpu-baseline
line "isres config 3 amd_isres_rank_prefetch=1"     --workload isres --no-cpu-baseline --param amd_isres_rank_prefetch=1
line "isres config 3 amd_isres_fast_scan=1"         --workload isres --no-cpu-baseline --param amd_isres_fast_scan=1
line "isres config 3 fast scan + rank prefetch"     --workload isres --no-cpu-baseline --param amd_isres_fast_scan=1 --param amd_isres_rank_prefetch=1
line "isres config 3 default (2)"                  --workload isres --no-cpu-baseline
line "isres config 3 fast scan + rank prefetch (2)" --workload isres --no-cpu-baseline --param amd_isres_fast_scan=1 --param amd_isres_rank_prefetch=1
line "mlsl config 4 default"                          --workload mlsl --no-cpu-baseline
for seg in 256 64 16; do
  line "mlsl config 4 amd_mlsl_seg_regens=$seg"       --workload mlsl --no-cpu-baseline --param amd_mlsl_seg_regens=$seg
done
line "mlsl config 4 amd_mlsl_prefetch=1"              --workload mlsl --no-cpu-baseline --param amd_mlsl_prefetch=1
line "mlsl config 4 prefetch + seg_regens=64"         --workload mlsl --no-cpu-baseline --param amd_mlsl_prefetch=1 --param amd_mlsl_seg_regens=64
for n in 512; do
  line "crs n=$n default (windows + resolver)"        --n $n --obj rastrigin --headline-only --no-cpu-baseline
  line "crs n=$n amd_max_spec=256"                    --n $n --obj rastrigin --headline-only --no-cpu-baseline --max-spec 256
  line "crs n=$n amd_chain_resolver=0"                --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_chain_resolver=0
done
for n in 64 128 256; do
  line "crs n=$n default (conservative passes)"       --n $n --obj rastrigin --headline-only --no-cpu-baseline
  line "crs n=$n windows + resolver"                  --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_forward=1 --param amd_chain_resolver=1
  line "crs n=$n windows + resolver, 256 slots"       --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_forward=1 --param amd_chain_resolver=1 --max-spec 256
done
for rep in 1 2; do
  line "crs headline default (lock version)"          --headline-only --no-cpu-baseline --steps 10 --warmup 2
  line "crs headline amd_chain_resolver=1"            --headline-only --no-cpu-baseline --steps 10 --warmup 2 --param amd_chain_resolver=1
  line "crs headline resolver, 256 slots"             --headline-only --no-cpu-baseline --steps 10 --warmup 2 --param amd_chain_resolver=1 --max-spec 256
done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
