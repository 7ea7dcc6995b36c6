#!/bin/bash
# Round 4, twenty-fifth GPU call: the conservative passes end with a doorbell (finish kernel -> pinned word, host spins) instead of a
# stream synchronisation ("amd_doorbell", default 1): CRS test files first (stop on failure), then n = 512 / 64 / the headline, A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call25; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_crs.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu -k "crs or CRS or golden or config1 or config2" 2>&1 | tail -4 | tee $O/crs_tests.log
grep -q "failed\|error" $O/crs_tests.log && exit 1
for n in 512 64; do for db in 1 0 1 0; do timeout -k 5 120 python bench.py --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_doorbell=$db 2>$O/err_${n}_$db.txt | tail -1 > $O/bench_n${n}_db$db.json; python -c "
import json
d = json.load(open('$O/bench_n${n}_db$db.json'))
print('n=$n doorbell=$db', round(d['value']), 'evals/s', round(d['ms_per_step'], 3), 'ms/step')"; done; done 2>&1 | tee $O/doorbell_ab.log
