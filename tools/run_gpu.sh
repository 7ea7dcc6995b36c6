#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 5 45 python -m pytest tests/test_gpu_stops.py -m gpu -x -q -k "device_resident" 2>&1 | tail -n 4
