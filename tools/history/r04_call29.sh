#!/bin/bash
# Round 4, very last GPU call: call 28's first attempt died with a core dump 1.7 s into pytest (output lost); the same command passed on
# the next box.  Fresh processes of the first test it would have run (the metric configuration), full output kept.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call29; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout -k 5 40 python -X faulthandler -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "metric_config" -p no:cacheprovider > $O/run$i.log 2>&1; echo "run $i rc=$? $(tail -1 $O/run$i.log | cut -c1-100)"; done
