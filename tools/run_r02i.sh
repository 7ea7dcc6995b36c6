#!/bin/bash
# GPU call: user-supplied objectives, then the whole -m gpu suite and the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_userobj.py -m gpu -q --timeout 300 2>&1 | tail -n 40 > gpurun_out/r02i/pytest_userobj.log
timeout -k 5 1500 python -m pytest tests -m gpu -q --timeout 900 -x --deselect tests/test_gpu_userobj.py 2>&1 | tail -n 30 > gpurun_out/r02i/pytest_all.log
timeout -k 5 600 python bench.py > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
tail -n 12 gpurun_out/r02i/pytest_userobj.log; tail -n 6 gpurun_out/r02i/pytest_all.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02i/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 2), d["roofline"]["frac"], d.get("speedup_vs_cpu_single_thread"), {k: (round(v["value"]), round(v["roofline_frac"], 3)) for k, v in d.get("other_sizes", {}).items()})
for k, v in d.get("other_workloads", {}).items():
    print("   ", k, v.get("value"), v.get("roofline", {}).get("kernel"), v.get("roofline", {}).get("frac"), v.get("cpu_baseline", {}).get("value"), v.get("error"))
print("    e2e", d.get("nlopt_optimize_end_to_end"))
PY
