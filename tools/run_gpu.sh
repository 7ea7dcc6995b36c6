#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_gpu_kernels.py -k "ranking_bits" -m gpu -q --timeout 120 2>&1 | tail -n 12
timeout -k 5 300 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py -k "isres" -m gpu -q --timeout 200 2>&1 | tail -n 4
for tp in 0 1; do
NLA_ISRES_BITS_TWO_PASS=$tp timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_tp$tp.json 2> $O/bench_isres_tp$tp.err
python - <<PY
import json
d=json.loads(open("$O/bench_isres_tp$tp.json").read().strip().splitlines()[-1])
print("two_pass=$tp", round(d["value"]), round(d["ms_per_step"],2), d["phases"])
PY
done
