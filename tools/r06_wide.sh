#!/bin/bash
# round 6: the 32-coordinates-per-thread build of the resident L-BFGS kernel (4096 < n <= 8192) — its bit-for-bit test against the streaming
# kernel, the COBYLA / MLSL files, then MLSL at n = 8192 on both kernels (round 5: 163 ms per iteration on the streaming kernel, 0.049 of HBM)
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_cobyla.py tests/test_gpu_exact_local.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r06/wide_tests.txt 2>&1; tail -6 gpurun_out/r06/wide_tests.txt
O=gpurun_out/r06/lbfgs_wide_ab.txt; : > $O
for p in "" "--param amd_lbfgs_streaming=1"; do
  for n in 8192 6144; do
    timeout 300 python bench.py --workload mlsl --n $n --steps 2 --warmup 1 --no-cpu-baseline $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mlsl n=$n $p', d['value'], 'evals/s', d['ms_per_step'], 'ms/step', json.dumps(d.get('roofline')), json.dumps(d.get('phases'))[:300])" >> $O 2>&1
  done
done
cat $O
