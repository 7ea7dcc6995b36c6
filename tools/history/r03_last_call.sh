#!/bin/bash
# round 3's last GPU seconds (5.6 minutes were left): (1) the default path after the last host-side commits — multi-rank files (set-up
# agreement, RCCL staging) and the ISRES file (generator stream refactor); (2) the opt-in ISRES overlap mode: its parity tests and
# the A/B of the config-3 generation time.
mkdir -p gpurun_out/r03_last
timeout 150 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03_last/multiproc.log
NLA_TEST_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_isres.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r03_last/isres.log
for ov in 0 1; do
    NLA_ISRES_OVERLAP=$ov timeout 60 python bench.py --workload isres --steps 3 --warmup 1 2>gpurun_out/r03_last/bench_ov$ov.err | tail -1 > gpurun_out/r03_last/bench_ov$ov.json
    python -c "
import json
d = json.load(open('gpurun_out/r03_last/bench_ov$ov.json'))
print('overlap=$ov', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"
done 2>&1 | tee gpurun_out/r03_last/ab.log
