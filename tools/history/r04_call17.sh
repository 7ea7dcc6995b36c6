#!/bin/bash
# Round 4, seventeenth GPU call: the register-resident chain walk resolves fewer individuals per round than the LDS walk (326 against
# 246 rounds per generation, call 16) with identical results — which part: the row registers (xlds: entries from LDS instead) or the
# scalar loads of rho (xvol: volatile)?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call17; mkdir -p $O
for v in "" _chainlds _xvol _xlds; do NLOPT_AMD_LIB=$GRAFT_REPO_ROOT/nlopt_amd/lib/libnlopt_amd$v.so timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_isres$v.json; python -c "
import json
d = json.load(open('$O/bench_isres$v.json')); p = d.get('phases')
print('lib$v', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation; evolve', round(p['evolve_s_per_gen'] * 1e3, 2), 'ms, rounds', round(p['evolve_rounds_per_gen'], 1), 'enqueued', round(p['evolve_rounds_enqueued_per_gen'], 1))"; done 2>&1 | tee $O/bench.log
