#!/bin/bash
# Round 4, twenty-sixth GPU call: the status records and the doorbell in explicitly coherent pinned memory: the CRS test file, n = 512
# twice, the headline once
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call26; mkdir -p $O
timeout -k 5 400 python -m pytest tests/test_gpu_crs.py -x -q -m gpu 2>&1 | tail -3 | tee $O/crs_tests.log
grep -q "failed\|rror" $O/crs_tests.log && exit 1
for r in 1 2; do timeout -k 5 120 python bench.py --n 512 --obj rastrigin --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('n=512', round(d['value']), 'evals/s')"; done 2>&1 | tee $O/n512.log
timeout -k 5 200 python bench.py --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('headline', round(d['value']), 'evals/s', d['roofline']['frac'], d['roofline'].get('frac_useful'))" | tee $O/headline.log
