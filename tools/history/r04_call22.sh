#!/bin/bash
# Round 4, twenty-second GPU call: the scan counts draws with sigma' from v_exp_f32 and a slack around the bounds, exact expressions only
# for draws inside the slack: ISRES tests, config 3 twice, kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call22; mkdir -p $O
timeout -k 5 300 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py -x -q -m gpu -k "isres or ISRES or config3" 2>&1 | tail -3 | tee $O/isres_tests.log
grep -q "failed\|error" $O/isres_tests.log && exit 1
for r in 1 2; do timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_isres_$r.json; python -c "
import json
d = json.load(open('$O/bench_isres_$r.json')); p = d.get('phases')
print(round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation; evolve', round(p['evolve_s_per_gen'] * 1e3, 2), 'ms, rounds', round(p['evolve_rounds_per_gen'], 1), 'enqueued', round(p['evolve_rounds_enqueued_per_gen'], 1))"; done 2>&1 | tee $O/bench.log
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/i -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/i.err
f=$(find $O/i -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; rm -rf $O/i
head -12 $O/isres_kernel_stats.csv
