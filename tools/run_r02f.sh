#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 200 -k chain 2>&1 | tail -5 > gpurun_out/r02f/pytest_chain.log
for t in 0 1 2 4 8 13; do
  NLA_CHAIN_TUNE=$t timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only --max-spec 128 > gpurun_out/r02f/bench_tune$t.json 2> gpurun_out/r02f/bench_tune$t.err
done
tail -3 gpurun_out/r02f/pytest_chain.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02f/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["ms_per_step"], 2), "frac", round(d["roofline"]["frac"], 3), "per launch", round(d["roofline"].get("avg_trials_consumed_per_launch"), 1), "launch ms", round(d["roofline"]["avg_launch_ms"], 3))
    except Exception as e:
        print(f, "failed", e)
PY
