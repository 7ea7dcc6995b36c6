#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 5 100 python -m pytest tests/test_gpu_kernels.py -k "mt_stream or ranking_bits or init or vitter" tests/test_gpu_fullsize.py -m gpu -q --timeout 90 2>&1 | tail -n 3
timeout -k 5 100 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_j4.json 2> $O/bench_isres_j4.err
python - <<PY
import json
d=json.loads(open("$O/bench_isres_j4.json").read().strip().splitlines()[-1])
print("isres", round(d["value"]), round(d["ms_per_step"],2), d["phases"])
PY
