#!/bin/bash
# Round 5, GPU call 2: the suite after the promotions / deletions (resolver everywhere, windows at every n, MLSL prefetch + short segments),
# then the CRS2_LM lines (early exit behind a new best).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c2; mkdir -p $O
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    hs = d.get("host_split") or {}
    r = d.get("roofline") or {}
    w = d.get("window") or {}
    print("%-50s %9.0f evals/s  %8.3f ms/step  frac %.4f useful %s  engine %.3f walk %.3f kernel %.3f s / %s passes; slots %s used %s newbest %s role %s  %s" % (
        sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, r.get("frac_useful"), hs.get("engine_s", 0), hs.get("walk_s", 0),
        hs.get("gather_kernel_s", 0), hs.get("passes"), w.get("slots_started"), w.get("slots_used"), w.get("newbest"), w.get("role"), d.get("phases", "")))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
for rep in 1 2; do
  line "crs headline default"                 --headline-only --no-cpu-baseline --steps 10 --warmup 2
  line "crs headline 256 slots"               --headline-only --no-cpu-baseline --steps 10 --warmup 2 --max-spec 256
  line "crs headline 192 slots"               --headline-only --no-cpu-baseline --steps 10 --warmup 2 --max-spec 192
done
for n in 64 128 256 512 1024 2048; do
  line "crs n=$n default"                     --n $n --obj rastrigin --headline-only --no-cpu-baseline
done
line "crs n=64 conservative"                  --n 64 --obj rastrigin --headline-only --no-cpu-baseline --param amd_forward=0
line "mlsl config 4 default"                  --workload mlsl --no-cpu-baseline
line "isres config 3 default"                 --workload isres --no-cpu-baseline
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
