#!/bin/bash
# round 6: lighter fences around the resolver's turn and the evaluation of the chain kernel below n = 2048 — the CRS2_LM device tests (golden traces,
# windows, full-size fixtures, kernels, uncached-pool stress), then the sizes of the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 5 1500 python -m pytest tests/test_gpu_crs.py tests/test_gpu_crs_windows.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_stops.py tests/test_gpu_maximise.py tests/test_gpu_fixed_dims.py tests/test_gpu_multiproc.py tests/test_gpu_zz_uncached_stress.py -x -q -p no:cacheprovider > $O/fences_tests.txt 2>&1; tail -3 $O/fences_tests.txt
: > $O/fences_sizes.txt
for rep in 1 2; do for cfg in "1024 rastrigin 32768" "512 rastrigin 65536" "256 rastrigin 131072" "128 rastrigin 262144" "64 rastrigin 524288" "4096 griewank 2000"; do set -- $cfg
timeout 300 python bench.py --n $1 --obj $2 --evals-per-step $3 --steps 4 --warmup 1 --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=$1 rep $rep %.0f evals/s frac %.3f' % (d['value'], d['roofline']['frac']))" >> $O/fences_sizes.txt; done; done
cat $O/fences_sizes.txt
