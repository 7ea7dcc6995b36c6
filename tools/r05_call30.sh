#!/bin/bash
# Round 5, GPU call 30: ISRES — the generator's segment-state array sized for 16 generations from the start (its doublings freed device memory
# in the middle of generations: 1.8 ms each, call 29).  ISRES device tests, config 3 twice, the frees in the API trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c30; mkdir -p $O
timeout -k 5 110 python -X faulthandler -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py -x -q -m gpu -k "isres" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)"
for rep in 1 2; do
  timeout -k 5 100 python bench.py --workload isres --no-cpu-baseline --steps 3 --warmup 1 --full-line 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('isres config 3: %.0f evals/s %.2f ms/generation' % (d['value'], d['ms_per_step']), {k: round(v*1e3,2) for k,v in d['phases'].items() if '_s_per_' in k})" | tee -a $O/ab.log
done
