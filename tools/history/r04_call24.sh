#!/bin/bash
# Round 4, twenty-fourth GPU call: the device-resolved windows (crs_chain_kernel, "amd_forward") at the small sizes, where the default
# is the conservative passes: n = 512 and n = 64, both modes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call24; mkdir -p $O
for n in 512 64; do for fw in 0 1; do timeout -k 5 120 python bench.py --n $n --obj rastrigin --headline-only --no-cpu-baseline --param amd_forward=$fw 2>$O/err_${n}_$fw.txt | tail -1 > $O/bench_n${n}_fw$fw.json; python -c "
import json
d = json.load(open('$O/bench_n${n}_fw$fw.json'))
print('n=$n forward=$fw', round(d['value']), 'evals/s', round(d['ms_per_step'], 3), 'ms/step', d['config'].get('workload'), (d.get('roofline') or {}).get('frac'))"; done; done 2>&1 | tee $O/forward_small_n.log
