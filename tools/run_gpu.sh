#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 5 200 python -m pytest tests/test_gpu_kernels.py -k "mt_stream or ranking_bits or init" -m gpu -q --timeout 120 2>&1 | tail -n 4
timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_j.json 2> $O/bench_isres_j.err
python - <<PY
import json
d=json.loads(open("$O/bench_isres_j.json").read().strip().splitlines()[-1])
print("isres", round(d["value"]), round(d["ms_per_step"],2), d["phases"])
PY
cd /tmp; rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt -o i -- python $GRAFT_REPO_ROOT/bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT; f=$(find /tmp/kt -name '*.db' | head -1); python profiles/summarize_rocpd.py $f | head -9
