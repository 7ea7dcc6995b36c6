#!/bin/bash
# round 6: the algorithm-level shim on the device, then the headline's rocprofv3 evidence (kernel trace + separate FETCH / WRITE passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_algs_shim.py tests/test_gpu_isres.py -x -q -p no:cacheprovider > $O/shim_isres_tests.txt 2>&1; tail -5 $O/shim_isres_tests.txt
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $O/k -o crs -- python bench.py --headline-only --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/k.err
f=$(find $O/k -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 2000 300 > $O/timeline.txt; rm -rf $O/k; head -8 $O/kernel_stats.csv | cut -c1-120
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 300 rocprofv3 --pmc $c -d $O/p -o p -- python bench.py --headline-only --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> $O/p.err
  f=$(find $O/p -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --pmc > $O/pmc_$(echo $c | tr A-Z a-z | sed 's/_size//').csv; rm -rf $O/p
done
head -4 $O/pmc_fetch.csv | cut -c1-200 | tail -3; head -4 $O/pmc_write.csv | tail -3 | cut -c1-200
timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_isres.json').read().strip().splitlines()[-1]); print('isres', d['value'], d['ms_per_step'], d.get('pinned_run'))"
