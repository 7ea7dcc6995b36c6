#!/bin/bash
# Round 5, first GPU calls: what round 4 wrote after its GPU minutes were spent, in the order of what depends on what.
#   gpurun --timeout 1200 -- 'bash tools/r05_first_call.sh 1'     (~13 min: suite, staged tests, ISRES / MLSL A/Bs)
#   gpurun --timeout 1200 -- 'bash tools/r05_first_call.sh 2'     (~12 min: CRS2_LM A/Bs, the default bench line)
#
# Part 1
#  (1) the whole -m gpu suite: the device-resolved CRS2_LM windows with the resolver wavefront became the default for 512 <= n < 2048 in
#      round 4's last GPU call; the full suite has not run on a device since (only tests/test_gpu_chain_resolver.py's kernel tests and
#      whole-run comparisons did: profiles/r04_crs_chain_resolver.txt) — unless the driver's round-4 record (GPUTEST_r04.json) is green;
#  (1b) tests/staged/test_gpu_chain_resolver_small.py: the windows + resolver forced on where they are not the default (golden cases, drawn
#      configurations with n < 512 and populations barely above n) — 40 tests that have only run over the emulated device.  Green: they
#      move back into tests/test_gpu_chain_resolver.py;
#  (2) the staged read-ahead kernel of the ISRES ranking pipeline (tests/staged/test_gpu_isres_rank_prefetch.py) and its A/B on config 3
#      (bench.py --workload isres --param amd_isres_rank_prefetch=1).  Green + faster: default 1 in isres_driver.c, the file moves to tests/;
#  (2b) the staged evolve scan with the exp off the serial chain (tests/staged/test_gpu_isres_fast_scan.py: the launcher with and without the
#      flag writes the same rows, state and workspace; whole runs bit-identical) and its A/B (--param amd_isres_fast_scan=1; the phases
#      field of the line splits rank / evolve).  Green + faster: default 1 in isres_driver.c, the file moves to tests/;
#  (2c) MLSL's stream in shorter segments (tests/staged/test_gpu_mlsl_short_segments.py; --param amd_mlsl_seg_regens=...: sampling_s_per_iter
#      in the phases field is what should move).  Green + faster: the best value becomes mlsl_driver.c's default.
# Part 2
#  (3) CRS2_LM A/Bs, one line each (bench.py prints host_split = engine call / in-order walk / gather kernel per run):
#        n = 512: default | amd_max_spec=256 | amd_chain_resolver=0 | amd_forward=0
#        n = 64, 128, 256: default (conservative passes) | amd_forward=1 amd_chain_resolver=1 [amd_max_spec=256]
#        n = 4096 (headline): default (lock version) | amd_chain_resolver=1 | amd_chain_resolver=1 amd_max_spec=256
#      The thresholds in crs_engine.c (NLA_CRS_FORWARD_MIN_N = 512; resolver below 2048) are where something was measured, not where the
#      break-evens are: move them to what these lines say;
#  (4) the driver's default bench line.
# Every step runs under its own timeout; a failed or hung step costs its limit, not the call.
PART=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_first; mkdir -p $O
staged() {   # staged <log name> <limit s> <test file>
    timeout -k 5 $2 python -X faulthandler -m pytest $3 -x -q -m gpu -p no:cacheprovider > $O/$1.log 2>&1; echo "$3 rc=$? $(tail -1 $O/$1.log)"
}
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open("gpurun_out/r05_first/last.json"))
    hs = d.get("host_split") or {}
    print("%-58s %9.0f evals/s  %8.3f ms/step  frac %s  host: engine %.3f walk %.3f kernel %.3f s over %s passes, list redrawn %s times (%s beside the device, %.4f s)  %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), hs.get("engine_s", 0), hs.get("walk_s", 0),
        hs.get("gather_kernel_s", 0), hs.get("passes"), hs.get("list_refreshes"), hs.get("list_refreshes_beside_device"), hs.get("list_refresh_s", 0),
        d.get("phases", "")))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
if [ "$PART" = 1 ] || [ "$PART" = all ]; then
  timeout -k 5 600 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
  staged staged_resolver_small 300 tests/staged/test_gpu_chain_resolver_small.py
  staged staged_isres 240 tests/staged/test_gpu_isres_rank_prefetch.py
  staged staged_isres_scan 300 tests/staged/test_gpu_isres_fast_scan.py
  staged staged_mlsl_seg 300 tests/staged/test_gpu_mlsl_short_segments.py
  for rep in 1 2; do
    line "isres config 3 default"                      --workload isres --no-cpu-baseline
    line "isres config 3 amd_isres_rank_prefetch=1"     --workload isres --no-cpu-baseline --param amd_isres_rank_prefetch=1
    line "isres config 3 amd_isres_fast_scan=1"         --workload isres --no-cpu-baseline --param amd_isres_fast_scan=1
    line "isres config 3 fast scan + rank prefetch"     --workload isres --no-cpu-baseline --param amd_isres_fast_scan=1 --param amd_isres_rank_prefetch=1
  done
  for seg in 1024 256 64 16; do
    line "mlsl config 4 amd_mlsl_seg_regens=$seg"       --workload mlsl --no-cpu-baseline --param amd_mlsl_seg_regens=$seg
  done
  # the generator working ahead on its own stream: never overlapped in round 4 because the distance scratch was freed and reallocated every
  # iteration (hipFree waits for every stream); it doubles now (mlsl_driver.c need_D)
  line "mlsl config 4 amd_mlsl_prefetch=1"              --workload mlsl --no-cpu-baseline --param amd_mlsl_prefetch=1
  line "mlsl config 4 prefetch + seg_regens=64"         --workload mlsl --no-cpu-baseline --param amd_mlsl_prefetch=1 --param amd_mlsl_seg_regens=64
fi
if [ "$PART" = 2 ] || [ "$PART" = all ]; then
  for n in 512; do
    line "crs n=$n default (windows + resolver)"        --n $n --obj rastrigin --headline-only --no-cpu-baseline
    line "crs n=$n amd_max_spec=256"                    --n $n --obj rastrigin --headline-only --no-cpu-baseline --max-spec 256
    line "crs n=$n amd_chain_resolver=0"                --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_chain_resolver=0
    line "crs n=$n amd_forward=0"                       --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_forward=0
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
  timeout -k 5 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$? $(head -c 300 $O/bench_default.json)"
fi
