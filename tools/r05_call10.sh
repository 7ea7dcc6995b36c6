#!/bin/bash
# Round 5, GPU call 10: the hand-over through the elements with the stores delayed to tick 8 of the next block (no wait for a store at any block top).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c10; mkdir -p $O
date +%s > $O/t0
timeout -k 5 600 python -X faulthandler -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_fullsize.py -x -q -m gpu -k "isres or nan or config3" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(grep -v '^  File\|^$' $O/tests.log | head -12 | cut -c1-250)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>$O/last.err | tail -1 > $O/last.json
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
  head -5 $O/last.err | cut -c1-300
done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
