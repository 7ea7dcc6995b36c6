#!/bin/bash
# Round 5, GPU call 18: the ranking pipeline — a unit that missed its look-ahead falls back a little (s_sleep) so that the next ones hit.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c18; mkdir -p $O
date +%s > $O/t0
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    print("%-44s %9.0f evals/s  %8.3f ms/step  pipeline %.3f ms/launch %.1f ns/tick  %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("avg_launch_ms") or 0, r.get("achieved") or 0,
          {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") else v) for k, v in ph.items() if k.endswith("_s_per_gen")}))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for rep in 1 2; do
  line "isres config 3, no fall-back" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  for v in lag24 lag48 lag96 lag48k48; do
    NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_$v.so line "isres config 3, $v" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  done
done
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_lag48.so timeout -k 5 300 python -m pytest tests/test_gpu_isres.py -x -q -m gpu -k "stochastic_ranking or golden" -p no:cacheprovider 2>&1 | tail -1
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
