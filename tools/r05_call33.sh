#!/bin/bash
# Round 5, GPU call 33 (the last seconds of the budget): MLSL with the generator's state array sized up front — device tests of the MLSL
# files, then config 4.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c33; mkdir -p $O
timeout -k 3 12 python bench.py --workload mlsl --no-cpu-baseline --steps 2 --warmup 1 --full-line 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mlsl config 4: %.0f evals/s %.3f ms/iteration, launch %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']), {k: round(v*1e3,2) for k,v in d['phases'].items() if '_s_per_' in k})" | tee -a $O/ab.log
timeout -k 3 32 python -X faulthandler -m pytest tests/test_gpu_mlsl.py tests/test_gpu_mlsl_short_segments.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mlsl or sobol" -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)" | tee -a $O/ab.log
