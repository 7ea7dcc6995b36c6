#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py -k "isres" -m gpu -q --timeout 600 2>&1 | tail -n 8
timeout -k 5 300 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres.json 2> $O/bench_isres.err
python - <<PY
import json
d=json.loads(open("$O/bench_isres.json").read().strip().splitlines()[-1])
print(round(d["value"]), d["ms_per_step"], d["roofline"])
PY
