#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_crs.py -m gpu -q --timeout 200 2>&1 | tail -n 3 > gpurun_out/r02h/pytest_crs.log
NLA_CHAIN_UNCACHED=1 timeout -k 5 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_crs.py -m gpu -q --timeout 600 -k "crs or config5 or metric" 2>&1 | tail -n 3 > gpurun_out/r02h/pytest_uncached.log
for unc in 0 1; do for ms in 24 32 40 48 64 96 128; do
  NLA_CHAIN_UNCACHED=$unc timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only --max-spec $ms > gpurun_out/r02h/bench_u${unc}_ms$ms.json 2> gpurun_out/r02h/bench_u${unc}_ms$ms.err
done; done
tail -n 2 gpurun_out/r02h/pytest_crs.log gpurun_out/r02h/pytest_uncached.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02h/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 3), "per launch", round(d["roofline"].get("avg_trials_consumed_per_launch"), 1), "launch ms", round(d["roofline"]["avg_launch_ms"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
