#!/bin/bash
# Round 4, call 11: software-pipelined pair-distance kernel, MLSL's per-iteration host arrays pinned; prefetch A/B once more
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlsl.py tests/test_gpu_fullsize.py tests/test_gpu_exact_local.py tests/test_gpu_multiproc.py tests/test_gpu_cobyla.py tests/test_gpu_mma.py -q -m gpu -k "mlsl or MLSL or pair_distance" 2>&1 | tail -5 | tee $O/mlsl_tests.log
for pf in 0 1 0 1; do timeout 200 python bench.py --workload mlsl --no-cpu-baseline --param amd_mlsl_prefetch=$pf 2>/dev/null | tail -1 > $O/bench_mlsl_pf$pf.json; python -c "
import json
d = json.load(open('$O/bench_mlsl_pf$pf.json'))
print('prefetch=$pf', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'))"; done 2>&1 | tee $O/ab_prefetch.log
timeout 300 rocprofv3 --kernel-trace -d $O/m -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline --param amd_mlsl_prefetch=1 > /dev/null 2> $O/m.err
f=$(find $O/m -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --timeline > $O/mlsl_prefetch_timeline.csv; python profiles/summarize_rocpd.py $f > $O/mlsl_kernel_stats.csv; rm -rf $O/m; head -6 $O/mlsl_kernel_stats.csv
