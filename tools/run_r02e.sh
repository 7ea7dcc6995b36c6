#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -k chain 2>&1 | grep -E "AssertionError|device next|passed|failed|E   " | cut -c1-900 > gpurun_out/r02e/pytest_chain.log
for ms in 0 64 128; do
  timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only --max-spec $ms > gpurun_out/r02e/bench_ms$ms.json 2> gpurun_out/r02e/bench_ms$ms.err
done
cat gpurun_out/r02e/pytest_chain.log | head -30
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02e/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 3), "per launch", round(d["roofline"].get("avg_trials_consumed_per_launch"), 1), "launch ms", round(d["roofline"]["avg_launch_ms"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
