#!/bin/bash
# GPU call 2: value forwarding in the CRS2_LM gather — kernel / trace / full-size parity, then the bench with it on and off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_crs.py tests/test_gpu_exact_local.py -m gpu -q -x --timeout 600 2>&1 | tail -30 > gpurun_out/r02b/pytest_crs.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 900 2>&1 | tail -30 > gpurun_out/r02b/pytest_fullsize.log
timeout 600 python bench.py > gpurun_out/r02b/bench_fwd.json 2> gpurun_out/r02b/bench_fwd.err
NLA_CRS_NO_FORWARD=1 timeout 600 python bench.py > gpurun_out/r02b/bench_nofwd.json 2> gpurun_out/r02b/bench_nofwd.err
NLA_CRS_PASS_LOG=gpurun_out/r02b/passlog_fwd.csv timeout 300 python bench.py --steps 5 --warmup 2 > /dev/null 2>&1
tail -4 gpurun_out/r02b/pytest_crs.log gpurun_out/r02b/pytest_fullsize.log
python - <<'PY'
import json
for f in ("fwd", "nofwd"):
    try:
        d = json.loads(open("gpurun_out/r02b/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("avg_trials_consumed_per_launch"), {k: (v["value"], v["roofline_frac"]) for k, v in d.get("other_sizes", {}).items()})
    except Exception as e:
        print(f, "failed", e)
PY
