#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_gpu_userobj.py -m gpu -q --timeout 300 2>&1 | tail -n 40 > gpurun_out/r02j/pytest_userobj.log
tail -n 30 gpurun_out/r02j/pytest_userobj.log | cut -c1-300
