#!/bin/bash
# Round 5, GPU call 9: the ranking pipeline's hand-over through the elements themselves (SR_ASYNC) against the counter protocol (variant
# "sync"), and two timing-only variants of the counter protocol that say where its 2.8 us per block go (no poll / no wait for the stores:
# wrong ranks, timed only).  ISRES device tests on the default build, config 3 lines.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c9; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_fullsize.py -x -q -m gpu -k "isres or nan or config3" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    print("%-44s %9.0f evals/s  %8.3f ms/step  pipeline %.3f ms/launch %.1f ns/tick (%d ticks)  %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("avg_launch_ms") or 0, r.get("achieved") or 0, r.get("serial_ticks_per_launch") or 0,
          {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") or k.endswith("_s_per_iter") else v) for k, v in ph.items() if k.endswith("_s_per_gen") or "rounds" in k}))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for rep in 1 2; do
  line "isres config 3: hand-over through the elements" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_sync.so line "isres config 3: counter protocol" --workload isres --no-cpu-baseline --steps 3 --warmup 1
done
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_nopoll.so line "TIMING ONLY counter protocol, no poll" --workload isres --no-cpu-baseline --steps 3 --warmup 1
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_nowait.so line "TIMING ONLY counter protocol, no store wait" --workload isres --no-cpu-baseline --steps 3 --warmup 1
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
