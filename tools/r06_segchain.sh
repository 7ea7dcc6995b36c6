#!/bin/bash
# round 6: ISRES evolve — the segment tables of a look-up round built in the scan launch's tail, a small chain kernel over them
# (the A/B against the chain kernel of rounds 3-5 is on record in profiles/r06_isres_segchain_ab.txt; that kernel is gone).  Here: the ISRES device
# tests, the placement of the jump-ahead kernel once more (the old chain kernel's 147 KB of LDS were why it left the evolve rounds), a profile.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 5 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py tests/test_gpu_nan.py tests/test_gpu_stops.py -x -q -p no:cacheprovider > $O/segchain_tests.txt 2>&1; tail -3 $O/segchain_tests.txt
: > $O/segchain_jump_ab.txt
for rep in 1 2 3; do
  for j in 1 0; do
    NLA_ISRES_JUMP_IN_RANK=$j NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_dbg.so timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('segment chain, jump beside the pipeline=$j rep=$rep  %.3f ms/generation  pipeline %.3f ms/launch  identical_to_reference=%s' % (d['ms_per_step'], r['avg_launch_ms'], d.get('pinned_run',{}).get('identical_to_reference')))" >> $O/segchain_jump_ab.txt
  done
done
cat $O/segchain_jump_ab.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/k -o c -- python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_under_rocprof.json 2> $O/k.err
f=$(find $O/k -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 0 4000 > $O/isres_timeline.txt; rm -rf $O/k
head -10 $O/isres_kernel_stats.csv | cut -c1-110
