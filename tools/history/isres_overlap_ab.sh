#!/bin/bash
# A/B of the ISRES overlap mode (isres_driver.c, "amd_isres_overlap": the generator on a stream of its own — the default since round 3)
# against the one-stream generation, at BASELINE config 3: its parity tests, then the generation time both ways, twice each.
#   gpurun --timeout 600 -- 'bash tools/history/isres_overlap_ab.sh'
mkdir -p gpurun_out/isres_overlap_ab
timeout 300 python -m pytest tests/test_gpu_isres.py -x -q -m gpu -k "overlap" 2>&1 | tail -5 | tee gpurun_out/isres_overlap_ab/tests.log
for ov in 0 1 0 1; do
    NLA_ISRES_OVERLAP=$ov timeout 120 python bench.py --workload isres --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/isres_overlap_ab/bench_ov$ov.json
    python -c "
import json
d = json.load(open('gpurun_out/isres_overlap_ab/bench_ov$ov.json'))
print('overlap=$ov', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"
done 2>&1 | tee gpurun_out/isres_overlap_ab/ab.log
