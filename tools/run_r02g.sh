#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_crs.py -m gpu -q --timeout 200 2>&1 | tail -5 > gpurun_out/r02g/pytest_crs.log
timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "crs or config5 or metric" 2>&1 | tail -4 > gpurun_out/r02g/pytest_fullsize.log
for ms in 0 48 64 96 128 200 256; do
  timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only --max-spec $ms > gpurun_out/r02g/bench_ms$ms.json 2> gpurun_out/r02g/bench_ms$ms.err
done
NLA_CRS_FORWARD=0 timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only > gpurun_out/r02g/bench_nochain.json 2>/dev/null
NLA_CRS_PASS_LOG=gpurun_out/r02g/passlog.csv timeout -k 5 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --headline-only > /dev/null 2>&1
tail -2 gpurun_out/r02g/pytest_crs.log gpurun_out/r02g/pytest_fullsize.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02g/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 3), "per launch", round(d["roofline"].get("avg_trials_consumed_per_launch"), 1), "launch ms", round(d["roofline"]["avg_launch_ms"], 3), d["window"])
    except Exception as e:
        print(f, "failed", e)
PY
