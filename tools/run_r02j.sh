#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_userobj.py -m gpu -q --timeout 300 -x 2>&1 | tail -n 40 > gpurun_out/r02j/pytest_userobj.log
timeout -k 5 600 python -m pytest tests/test_gpu_mma.py tests/test_gpu_exact_local.py tests/test_gpu_multiproc.py -m gpu -q --timeout 300 -k "refusals or maximis or multiproc or isres" 2>&1 | tail -n 20 > gpurun_out/r02j/pytest_misc.log
tail -n 25 gpurun_out/r02j/pytest_userobj.log | cut -c1-400; tail -n 8 gpurun_out/r02j/pytest_misc.log | cut -c1-400
