#!/bin/bash
# GPU-box script (round 3, step a): the new multi-rank CRS tests + maximise regression tests, the L-BFGS occupancy change (tests + bench)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_multiproc.py tests/test_gpu_maximise.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r03a_new_tests.log
( timeout 600 python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_mlsl.py tests/test_gpu_exact_local.py tests/test_gpu_fullsize.py tests/test_gpu_host_callbacks.py -q -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > gpurun_out/r03a_lbfgs_tests.log
( timeout 300 python bench.py --workload mlsl --steps 2 --warmup 1 2>&1 | tail -3 ) > gpurun_out/r03a_bench_mlsl.json
cat gpurun_out/r03a_new_tests.log gpurun_out/r03a_lbfgs_tests.log; cut -c1-1500 gpurun_out/r03a_bench_mlsl.json
