#!/bin/bash
# Round 4, call 13: the fused commit again (slot index no longer truncated to 8 bits).  Tests FIRST, the benches only if they are green,
# every step under a hard time limit (call 12: a wrong kernel made 8 bench runs sit in their 200 s limits — 29 GPU-minutes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call13; mkdir -p $O
timeout -k 5 120 python -m pytest tests/test_gpu_crs.py -x -q -m gpu 2>&1 | tail -4 | tee $O/crs_golden.log
grep -q " passed" $O/crs_golden.log && ! grep -q "failed" $O/crs_golden.log || { echo "golden cases not green: stopping"; exit 1; }
timeout -k 5 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stops.py tests/test_gpu_dropin.py tests/test_gpu_fixed_dims.py tests/test_gpu_maximise.py tests/test_gpu_userobj.py tests/test_gpu_host_callbacks.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not mlsl and not MLSL and not isres and not ISRES" 2>&1 | tail -4 | tee $O/crs_tests.log
grep -q " passed" $O/crs_tests.log && ! grep -q "failed" $O/crs_tests.log || { echo "CRS files not green: stopping"; exit 1; }
for nn in 512 64; do for fz in 1 0 1 0; do timeout -k 5 60 python bench.py --n $nn --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only --param amd_fuse_commit=$fz 2>/dev/null | tail -1 > $O/bench_n${nn}_f$fz.json; python -c "
import json
d = json.load(open('$O/bench_n${nn}_f$fz.json'))
print('n=$nn fuse=$fz', round(d['value']), 'evals/s', round(d['ms_per_step'], 3), 'ms/step', d['roofline'].get('frac'))" || exit 1; done; done 2>&1 | tee $O/ab_fuse.log
