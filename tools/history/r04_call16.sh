#!/bin/bash
# Round 4, sixteenth GPU call: why the register-resident chain walk (22 us instead of 49 us per round) left the evolve time where it
# was — rounds that resolved something per generation (new stats fields) with the shipped chain kernel and with the LDS walk it replaced
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call16; mkdir -p $O
for v in "" _chainlds "" _chainlds; do NLOPT_AMD_LIB=$GRAFT_REPO_ROOT/nlopt_amd/lib/libnlopt_amd$v.so timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_isres$v.json; python -c "
import json
d = json.load(open('$O/bench_isres$v.json'))
print('lib$v', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/generation', d.get('phases'))"; done 2>&1 | tee $O/bench.log
