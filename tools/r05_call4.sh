#!/bin/bash
# Round 5, GPU call 4: lean CRS2_LM windows (one upload, control block cleared by the commit kernel, doorbell) — the CRS2_LM device tests,
# then A/B lines: doorbell on / off, workgroup shapes of the chain kernel below n = 2048 (variant builds shA / shB), and what a window
# costs in the kernel's three regimes at n = 64 / 512 (tools/chain_latency.py).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c4; mkdir -p $O
date +%s > $O/t0
timeout -k 5 600 python -X faulthandler -m pytest tests/test_gpu_crs.py tests/test_gpu_crs_windows.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu -k "crs or chain or window or metric or config2 or config5" -p no:cacheprovider > $O/crs_tests.log 2>&1; echo "crs tests rc=$? $(tail -1 $O/crs_tests.log)"
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
    print("%-44s %9.0f evals/s  %8.3f ms/step  frac %.4f  engine %.4f walk %.4f kernel(sampled) %.4f s / %s passes; avg launch %.1f us; slots %s used %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, hs.get("engine_s", 0), hs.get("walk_s", 0),
        hs.get("gather_kernel_s", 0), hs.get("passes"), 1e3 * (r.get("avg_launch_ms") or 0), w.get("slots_started"), w.get("slots_used")))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
NB="--headline-only --no-cpu-baseline --obj rastrigin --steps 4 --warmup 1 --evals-per-step 20000"
for rep in 1 2; do
for n in 64 256 512 1024; do
  line "n=$n default (lean, doorbell)"      --n $n $NB
  line "n=$n amd_doorbell=0"                --n $n $NB --param amd_doorbell=0
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_shA.so line "n=$n shape A (mid W8, low W4, tiny W4)"  --n $n $NB
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_shB.so line "n=$n shape B (mid W8 U32, low W8, tiny W8)" --n $n $NB
done
done
line "headline default" --headline-only --no-cpu-baseline --steps 10 --warmup 2
for n in 64 512; do timeout -k 5 120 python tools/chain_latency.py $n 100000 > $O/chain_latency_n$n.txt 2>&1; tail -13 $O/chain_latency_n$n.txt; done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
