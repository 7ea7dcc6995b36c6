#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
for i in 1 2 3; do timeout -k 5 300 python -m pytest tests/test_gpu_kernels.py -k chain_kernel -m gpu -q --timeout 200 2>&1 | tail -n 2; done
timeout -k 5 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_crs.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 2>&1 | tail -n 5
PYTHONPATH=. timeout -k 5 300 python tools/chain_latency.py > $O/chain_latency2.txt 2>&1; grep "K =  48\|K = 192" $O/chain_latency2.txt
for i in 1 2; do
timeout -k 5 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only > $O/bench_b$i.json 2> $O/bench_b$i.err
python - <<PY
import json
d=json.loads(open("$O/bench_b$i.json").read().strip().splitlines()[-1])
print("bench", round(d["value"]), "frac", round(d["roofline"]["frac"],3), "launch ms", round(d["roofline"]["avg_launch_ms"],3), d["window"])
PY
done
