#!/bin/bash
# GPU call 1: correctness of the reworked local optimisers + full-size fixtures, then baseline bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_exact_local.py tests/test_gpu_host_callbacks.py tests/test_gpu_dropin.py tests/test_gpu_lbfgs.py tests/test_gpu_mma.py tests/test_gpu_mlsl.py tests/test_gpu_stops.py -m gpu -q -x --timeout 600 2>&1 | tail -40 > gpurun_out/r02a/pytest_local.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 900 -s 2>&1 | tail -60 > gpurun_out/r02a/pytest_fullsize.log
timeout 600 python bench.py > gpurun_out/r02a/bench_crs.json 2> gpurun_out/r02a/bench_crs.err
timeout 600 python bench.py --workload mlsl --steps 3 --warmup 1 > gpurun_out/r02a/bench_mlsl.json 2> gpurun_out/r02a/bench_mlsl.err
tail -5 gpurun_out/r02a/pytest_local.log gpurun_out/r02a/pytest_fullsize.log
cat gpurun_out/r02a/bench_mlsl.json | cut -c1-600
