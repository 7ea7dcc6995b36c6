#!/bin/bash
# Round 5, GPU call 14: f(T) of a slot published before its mutation is formed (the resolver accepts on f(T) alone) against a build that
# publishes both words at the end; CRS2_LM device tests.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c14; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/test_gpu_crs.py tests/test_gpu_crs_windows.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_stops.py tests/test_gpu_multiproc.py -x -q -m gpu -k "not mlsl and not isres" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    hs = d.get("host_split") or {}
    r = d.get("roofline") or {}
    print("%-40s %9.0f evals/s  %8.3f ms/step  frac %.4f  engine %.4f kernel(sampled) %.4f s / %s passes; avg launch %.1f us" % (
        sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, hs.get("engine_s", 0), hs.get("gather_kernel_s", 0), hs.get("passes"), 1e3 * (r.get("avg_launch_ms") or 0)))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
NB="--headline-only --no-cpu-baseline --obj rastrigin --steps 6 --warmup 1 --evals-per-step 20000"
for rep in 1 2; do
for n in 64 256 512 1024; do
  line "n=$n f(T) published at once"   --n $n $NB
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_lateft.so line "n=$n both words at the end" --n $n $NB
done
done
line "headline" --headline-only --no-cpu-baseline --steps 10 --warmup 2
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_lateft.so line "headline, both words at the end" --headline-only --no-cpu-baseline --steps 10 --warmup 2
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
