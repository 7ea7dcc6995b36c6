#!/bin/bash
# the whole -m gpu suite, one file at a time under its own time limit (a hang costs one file's limit, not the call's)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/suite; mkdir -p $O; rm -f $O/*.log
for f in tests/test_gpu_*.py; do
  b=$(basename $f .py)
  timeout -k 5 ${SUITE_FILE_LIMIT:-120} python -m pytest $f -m gpu -q --timeout 100 > $O/$b.log 2>&1
  echo "$b rc=$? $(tail -n 1 $O/$b.log)"
done
grep -h "^FAILED\|^ERROR" $O/*.log | head -40
