#!/bin/bash
# Round 4, first thing on a GPU box: the ISRES overlap mode (isres_driver.c, "amd_isres_overlap" / NLA_ISRES_OVERLAP=1) was written in
# round 3 with no GPU minutes left to run it.  Parity first (the opt-in tests), then the A/B of the config-3 generation time.
#   gpurun --timeout 600 -- 'bash tools/r04_isres_overlap.sh'
# Make it the default (isres_driver.c: D.overlap) only if the tests are green AND the generation is faster.
mkdir -p gpurun_out/r04_overlap
timeout 300 python -m pytest tests/test_gpu_isres.py -x -q -m gpu -k "overlap" 2>&1 | tail -5 | tee gpurun_out/r04_overlap/tests.log
for ov in 0 1 0 1; do
    NLA_ISRES_OVERLAP=$ov timeout 120 python bench.py --workload isres --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r04_overlap/bench_ov$ov.json
    python -c "
import json
d = json.load(open('gpurun_out/r04_overlap/bench_ov$ov.json'))
print('overlap=$ov', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"
done 2>&1 | tee gpurun_out/r04_overlap/ab.log
