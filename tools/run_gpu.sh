#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 5 120 python -m pytest tests/test_gpu_cobyla.py tests/test_gpu_dropin.py -m gpu -q --timeout 60 2>&1 | tail -n 8
