#!/bin/bash
# Round 4, fifteenth GPU call: ISRES evolve rounds with the chain walk on the scalar unit (one workgroup of 1024 copies E, wavefront 0
# walks with v_readlane + one LDS read per individual) and the scan's first draw fetched beside the sigma' deviate: the ISRES files
# first (stop on failure), config 3 twice, the kernel statistics of one run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call15; mkdir -p $O
timeout -k 5 600 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py tests/test_gpu_nan.py -x -q -m gpu -k "isres or ISRES or config3 or nan" 2>&1 | tail -6 | tee $O/isres_tests.log
grep -q "failed\|error" $O/isres_tests.log && exit 1
for r in 1 2; do timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_isres_$r.json; python -c "
import json
d = json.load(open('$O/bench_isres_$r.json'))
print(round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"; done 2>&1 | tee $O/bench.log
timeout -k 5 300 rocprofv3 --kernel-trace -d $O/i -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/i.err
f=$(find $O/i -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; rm -rf $O/i
head -12 $O/isres_kernel_stats.csv
