#!/bin/bash
# Round 4, call 12: the staged commits done inside the advance launch ("amd_fuse_commit", default on): the CRS files, A/B at n = 512 / 64
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call12; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_crs.py tests/test_gpu_kernels.py tests/test_gpu_stops.py tests/test_gpu_dropin.py tests/test_gpu_fixed_dims.py tests/test_gpu_maximise.py tests/test_gpu_userobj.py tests/test_gpu_host_callbacks.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not mlsl and not MLSL and not isres and not ISRES" 2>&1 | tail -6 | tee $O/crs_tests.log
for nn in 512 64; do for fz in 1 0 1 0; do timeout 200 python bench.py --n $nn --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only --param amd_fuse_commit=$fz 2>/dev/null | tail -1 > $O/bench_n${nn}_f$fz.json; python -c "
import json
d = json.load(open('$O/bench_n${nn}_f$fz.json'))
print('n=$nn fuse=$fz', round(d['value']), 'evals/s', round(d['ms_per_step'], 3), 'ms/step', d['roofline'].get('frac'))"; done; done 2>&1 | tee $O/ab_fuse.log
