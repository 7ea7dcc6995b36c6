#!/bin/bash
# round 6: the reference-order sums of the resident L-BFGS kernel as chains of v_fmac_f64 with a row_newbcast DPP operand (one instruction per term instead of
# two v_readlane + add): the exact-order device tests, then config 4 in that mode and in the default one
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 5 1200 python -m pytest tests/test_gpu_exact_local.py tests/test_gpu_lbfgs.py tests/test_gpu_mlsl.py tests/test_gpu_host_callbacks.py tests/test_gpu_fullsize.py -x -q -p no:cacheprovider -k "not crs and not isres and not config3 and not metric" > $O/exact_tests.txt 2>&1; tail -4 $O/exact_tests.txt
: > $O/exact_ab.txt
for rep in 1 2; do
  timeout -k 5 300 python bench.py --workload mlsl --exact --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('exact order rep=$rep  %.2f ms/iteration  launch %.2f ms  %.3f ns per dependent add  identical_to_reference=%s' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], d.get('pinned_run',{}).get('identical_to_reference')))" >> $O/exact_ab.txt
  timeout -k 5 300 python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('default rep=$rep  %.2f ms/iteration  launch %.2f ms  frac %.3f  identical_to_reference=%s' % (d['ms_per_step'], r['avg_launch_ms'], r['frac'], d.get('pinned_run',{}).get('identical_to_reference')))" >> $O/exact_ab.txt
done
cat $O/exact_ab.txt
