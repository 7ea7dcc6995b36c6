#!/bin/bash
# round 6: ISRES — the evolve phase's deviates generated beside the ranking on a stream of their own instead of behind the bits' generator
# (A/B on the build with development switches: NLA_ISRES_DEVIATES_STREAM=0 is the placement before), ISRES / MLSL device tests at the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py tests/test_gpu_mlsl.py tests/test_gpu_crs_windows.py -x -q -p no:cacheprovider > $O/deviates_tests.txt 2>&1; tail -3 $O/deviates_tests.txt
: > $O/deviates_ab.txt
for rep in 1 2 3; do
  for j in 1 0; do
    NLA_ISRES_DEVIATES_STREAM=$j NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_dbg.so timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('deviates on their own stream=$j rep=$rep  %.3f ms/generation  pipeline %.3f ms/launch  identical_to_reference=%s' % (d['ms_per_step'], r['avg_launch_ms'], d.get('pinned_run',{}).get('identical_to_reference')))" >> $O/deviates_ab.txt
  done
done
cat $O/deviates_ab.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/k -o c -- python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_under_rocprof.json 2> $O/k.err
f=$(find $O/k -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 0 4000 > $O/isres_timeline.txt; rm -rf $O/k
head -8 $O/isres_kernel_stats.csv | cut -c1-110
