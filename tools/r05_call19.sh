#!/bin/bash
# Round 5, GPU call 19: the STREAMING L-BFGS kernel (n > 4096, host callbacks, user kernels; here forced on config 4 with amd_lbfgs_streaming=1)
# at one workgroup per compute unit (no spilled VGPRs: -DLB_WAVES_PER_EU=1) against two (62-107 spilled VGPRs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c19; mkdir -p $O
date +%s > $O/t0
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 200 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    print("%-50s %9.0f evals/s  %8.3f ms/step  frac %.4f avg launch %.3f ms  %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("frac") or 0, r.get("avg_launch_ms") or 0,
          {k: (round(v * 1e3, 2) if k.endswith("_s_per_iter") else v) for k, v in ph.items()}))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for rep in 1 2; do
  line "mlsl config 4, streaming kernel, 2 workgroups per CU" --workload mlsl --no-cpu-baseline --steps 2 --warmup 1 --param amd_lbfgs_streaming=1
  NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_lbw1.so line "mlsl config 4, streaming kernel, 1 workgroup per CU" --workload mlsl --no-cpu-baseline --steps 2 --warmup 1 --param amd_lbfgs_streaming=1
done
line "mlsl n=8192, streaming kernel, 2 workgroups per CU" --workload mlsl --n 8192 --no-cpu-baseline --steps 2 --warmup 1
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_lbw1.so line "mlsl n=8192, streaming kernel, 1 workgroup per CU" --workload mlsl --n 8192 --no-cpu-baseline --steps 2 --warmup 1
NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_lbw1.so timeout -k 5 600 python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_exact_local.py tests/test_gpu_host_callbacks.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -1
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
