#!/bin/bash
# round 3, the very last GPU seconds: does a LOW-PRIORITY generator stream (hipStreamCreateWithPriority) keep the overlapped generator
# work out of the way of the evolve phase's throughput-bound scan kernels?  A/B of the config-3 generation, ISRES file with it on.
mkdir -p gpurun_out/r03_last4
for bg in 0 1; do
    NLA_ISRES_RS_BACKGROUND=$bg timeout 25 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r03_last4/bench_bg$bg.err | tail -1 > gpurun_out/r03_last4/bench_bg$bg.json
    python -c "
import json
d = json.load(open('gpurun_out/r03_last4/bench_bg$bg.json'))
print('background=$bg', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"
done 2>&1 | tee gpurun_out/r03_last4/ab.log
NLA_ISRES_RS_BACKGROUND=1 timeout 20 python -m pytest tests/test_gpu_isres.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r03_last4/isres_bg.log
