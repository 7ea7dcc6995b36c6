#!/bin/bash
# Round 4, twentieth GPU call: the chain kernel with the straight-line precompute — stop reasons and the sum check (development build),
# then the shipped library: ISRES tests, config 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call20; mkdir -p $O
NLOPT_AMD_LIB=$GRAFT_REPO_ROOT/nlopt_amd/lib/libnlopt_amd_rnew.so timeout -k 5 100 python bench.py --workload isres --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep -A1 "evolve phase" | tail -8 | tee $O/reasons.log
timeout -k 5 300 python -m pytest tests/test_gpu_isres.py tests/test_gpu_fullsize.py -x -q -m gpu -k "isres or ISRES or config3" 2>&1 | tail -3 | tee $O/isres_tests.log
grep -q "failed\|error" $O/isres_tests.log && exit 1
for r in 1 2; do timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_isres_$r.json; python -c "
import json
d = json.load(open('$O/bench_isres_$r.json')); p = d.get('phases')
print(round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation; evolve', round(p['evolve_s_per_gen'] * 1e3, 2), 'ms, rounds', round(p['evolve_rounds_per_gen'], 1), 'enqueued', round(p['evolve_rounds_enqueued_per_gen'], 1))"; done 2>&1 | tee $O/bench.log
