#!/bin/bash
# ISRES iteration call: the ISRES GPU tests, then config-3 lines (default / ungated / one stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_isres; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_fullsize.py -x -q -m gpu -p no:cacheprovider -k "isres or nan" > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {
    local label=$1; shift
    timeout -k 5 150 python bench.py --workload isres --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2])); p = d["phases"]; r = d["roofline"]
    print("%-44s %8.0f evals/s %7.3f ms/gen | eval %.2f rank %.2f evolve %.2f rng %.2f ms | pipeline %.2f ms/launch %.1f ns/tick | rounds %.1f" % (
        sys.argv[1], d["value"], d["ms_per_step"], 1e3 * p["eval_s_per_gen"], 1e3 * p["rank_s_per_gen"], 1e3 * p["evolve_s_per_gen"],
        1e3 * p["rng_s_per_gen_inside_rank_and_evolve"], r.get("avg_launch_ms") or 0, r.get("achieved") or 0, p["evolve_rounds_per_gen"]))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
line "isres default $1"
line "isres amd_isres_gated=0 $1" --param amd_isres_gated=0
line "isres amd_isres_overlap=0 $1" --param amd_isres_overlap=0
line "isres default (2) $1"
