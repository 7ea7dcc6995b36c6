#!/bin/bash
# gated ISRES ranking with 4 blocks instead of 16 (a block must fill the device: one wavefront generates a segment in ~1.9 ms whatever the grid)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call10; mkdir -p $O
for g in 1 0 1 0; do timeout 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline --param amd_isres_gated=$g 2>/dev/null | tail -1 > $O/bench_isres_g$g.json; python -c "
import json
d = json.load(open('$O/bench_isres_g$g.json'))
print('gated=$g', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"; done 2>&1 | tee $O/ab_gated.log
timeout 300 rocprofv3 --kernel-trace -d $O/i -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/i.err
f=$(find $O/i -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --timeline > $O/isres_timeline.csv; rm -rf $O/i
