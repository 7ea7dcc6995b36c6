#!/bin/bash
# GPU call 3: device-resolved CRS2_LM windows (hip/crs_chain.hip): kernel vs sequential statement, traces, full-size parity, bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x --timeout 120 -k "chain" 2>&1 | tail -30 > gpurun_out/r02c/pytest_chain_kernel.log
if ! grep -q "passed" gpurun_out/r02c/pytest_chain_kernel.log || grep -q "failed" gpurun_out/r02c/pytest_chain_kernel.log; then
  echo "chain kernel test did not pass: falling back to A/B numbers only"; tail -30 gpurun_out/r02c/pytest_chain_kernel.log
fi
timeout -k 5 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_crs.py -m gpu -q -x --timeout 300 2>&1 | tail -30 > gpurun_out/r02c/pytest_crs.log
timeout -k 5 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "crs or config5 or metric" 2>&1 | tail -30 > gpurun_out/r02c/pytest_fullsize.log
timeout -k 5 600 python bench.py > gpurun_out/r02c/bench_chain.json 2> gpurun_out/r02c/bench_chain.err
NLA_CRS_FORWARD=0 timeout -k 5 400 python bench.py --no-cpu-baseline > gpurun_out/r02c/bench_nochain.json 2> gpurun_out/r02c/bench_nochain.err
NLA_CRS_PASS_LOG=gpurun_out/r02c/passlog_chain.csv timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for f in pytest_chain_kernel pytest_crs pytest_fullsize; do echo "== $f"; tail -n 4 gpurun_out/r02c/$f.log; done
python - <<'PY'
import json
for f in ("chain", "nochain"):
    try:
        d = json.loads(open("gpurun_out/r02c/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), round(d["ms_per_step"], 2), round(d["roofline"]["frac"], 3), d["roofline"].get("avg_trials_consumed_per_launch"), d["window"],
              {k: (round(v["value"]), round(v["roofline_frac"], 3)) for k, v in d.get("other_sizes", {}).items()})
        for k, v in d.get("other_workloads", {}).items():
            print("   ", k, v.get("value"), v.get("roofline", {}).get("kernel"), v.get("roofline", {}).get("frac"), v.get("error"))
        print("    e2e", d.get("nlopt_optimize_end_to_end"))
    except Exception as e:
        print(f, "failed", e)
PY
