#!/bin/bash
# Round 4, the last 2 GPU minutes (a gamble: getting the box may use them up): the staged resolver of the device-resolved windows —
# its direct kernel test, then n = 512 with the lock version and with the resolver
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call31; mkdir -p $O
timeout -k 3 45 python -X faulthandler -m pytest tests/staged/test_gpu_chain_resolver.py -x -q -m gpu -p no:cacheprovider -k "test_chain_kernel_with or changes_nothing" > $O/staged.log 2>&1; echo "rc=$? $(tail -1 $O/staged.log)"
for P in "amd_forward=1 amd_chain_resolver=1" "amd_forward=1" ""; do
  A=""; for kv in $P; do A="$A --param $kv"; done
  timeout -k 3 30 python bench.py --n 512 --obj rastrigin --headline-only --no-cpu-baseline $A 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('n=512 [$P]', round(d['value']), 'evals/s')" | tee -a $O/n512.log
done
