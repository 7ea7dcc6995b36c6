#!/bin/bash
# Round 5, GPU call 12: MLSL with the point set pre-sized and the samples merged into the order array; the headline's window size below 128.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c12; mkdir -p $O
date +%s > $O/t0
timeout -k 5 600 python -X faulthandler -m pytest tests/test_gpu_mlsl.py tests/test_gpu_mlsl_short_segments.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mlsl" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    w = d.get("window") or {}
    print("%-36s %9.0f evals/s  %8.3f ms/step  frac %.4f useful %s avg launch %.3f ms  %s %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, r.get("frac_useful"), r.get("avg_launch_ms") or 0,
          {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") or k.endswith("_s_per_iter") else v) for k, v in ph.items()}, {k: w.get(k) for k in ("slots_started", "slots_used", "role")} if w else ""))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for rep in 1 2; do
  line "mlsl config 4 (2 steps)" --workload mlsl --no-cpu-baseline --steps 2 --warmup 1
  line "mlsl config 4 (6 steps)" --workload mlsl --no-cpu-baseline --steps 6 --warmup 1
done
for rep in 1 2; do
for k in 128 112 96 64; do
  line "headline, windows of $k slots" --headline-only --no-cpu-baseline --steps 20 --warmup 5 --max-spec $k
done
done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
