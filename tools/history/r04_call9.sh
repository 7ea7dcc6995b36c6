#!/bin/bash
# Round 4, ninth GPU call: ISRES with the ranking pipeline started beside the generation of its bits ("amd_isres_gated", default on):
# the ISRES files, A/B at config 3, the timeline; MLSL sample prefetch A/B again (the local phase no longer fills the device)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_fullsize.py tests/test_gpu_multiproc.py tests/test_gpu_stops.py -x -q -m gpu -k "isres or ISRES or nan or config3" 2>&1 | tail -6 | tee $O/isres_tests.log
for g in 1 0 1 0; do timeout 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline --param amd_isres_gated=$g 2>/dev/null | tail -1 > $O/bench_isres_g$g.json; python -c "
import json
d = json.load(open('$O/bench_isres_g$g.json'))
print('gated=$g', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"; done 2>&1 | tee $O/ab_gated.log
for pf in 0 1 0 1; do timeout 200 python bench.py --workload mlsl --no-cpu-baseline --param amd_mlsl_prefetch=$pf 2>/dev/null | tail -1 > $O/bench_mlsl_pf$pf.json; python -c "
import json
d = json.load(open('$O/bench_mlsl_pf$pf.json'))
print('prefetch=$pf', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'))"; done 2>&1 | tee $O/ab_prefetch.log
timeout 300 rocprofv3 --kernel-trace -d $O/i -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/i.err
f=$(find $O/i -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --timeline > $O/isres_timeline.csv; python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; rm -rf $O/i
timeout 300 rocprofv3 --kernel-trace -d $O/m -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline --param amd_mlsl_prefetch=1 > /dev/null 2> $O/m.err
f=$(find $O/m -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --timeline > $O/mlsl_prefetch_timeline.csv; rm -rf $O/m
