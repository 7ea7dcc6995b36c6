#!/bin/bash
# round 3, the last 100 GPU seconds: the ISRES overlap mode is now the default — the ISRES file, smoke(), and the driver's default
# bench line (its other_workloads.isres entry is what changed)
mkdir -p gpurun_out/r03_last3
timeout 30 python -m pytest tests/test_gpu_isres.py -x -q -m gpu 2>&1 | tail -2 | tee gpurun_out/r03_last3/isres.log
timeout 20 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/r03_last3/smoke.log
timeout 75 python bench.py 2>gpurun_out/r03_last3/bench.err | tail -1 > gpurun_out/r03_last3/bench.json
python -c "
import json
d = json.load(open('gpurun_out/r03_last3/bench.json'))
print(d['metric'], round(d['value']), d['roofline']['frac'], d['roofline'].get('frac_useful'))
for k, v in d['other_workloads'].items(): print(k, round(v['value']), v['ms_per_step'])
" | tee gpurun_out/r03_last3/bench.log
