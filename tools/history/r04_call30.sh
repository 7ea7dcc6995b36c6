#!/bin/bash
# Round 4, closing GPU call: the library as shipped (default pinned memory behind the doorbell, the configuration of the full-suite run):
# the CRS test file and one n = 512 line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call30; mkdir -p $O
timeout -k 5 100 python -X faulthandler -m pytest tests/test_gpu_crs.py -x -q -m gpu -p no:cacheprovider > $O/crs_tests.log 2>&1; echo "rc=$? $(tail -1 $O/crs_tests.log)"
timeout -k 5 60 python bench.py --n 512 --obj rastrigin --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('n=512', round(d['value']), 'evals/s')" | tee $O/n512.log
