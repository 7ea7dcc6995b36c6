#!/bin/bash
# round 6: the wavefront COBYLA kernel — its tests (kernel against the real reference, GN_MLSL in both modes), the MLSL files, and the numbers
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_cobyla.py tests/test_gpu_mlsl.py tests/test_gpu_mma.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r06/cobyla_tests.txt 2>&1; tail -15 gpurun_out/r06/cobyla_tests.txt
timeout 600 python tools/cobyla_bench.py > gpurun_out/r06/cobyla_batched.txt 2>&1; cat gpurun_out/r06/cobyla_batched.txt
