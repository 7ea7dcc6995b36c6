#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 200 python -m pytest tests/test_gpu_maximise.py tests/test_gpu_isres.py -m gpu -q --timeout 120 2>&1 | tail -n 8
