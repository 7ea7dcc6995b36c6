#!/bin/bash
# Round 5, GPU call 16: where the 13 us go that ev2_scan0_kernel takes longer than the scan alone did: write workgroups first in the grid (A/B),
# no write workgroups at all (timing only, wrong results).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c16; mkdir -p $O
date +%s > $O/t0
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f > $2; rm -rf $1; }
for v in default wfirst nowrite; do
  if [ $v = default ]; then unset NLOPT_AMD_LIB; else export NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_$v.so; fi
  timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/k_$v -o isres -- python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$v.json 2> $O/$v.err
  summ $O/k_$v $O/stats_$v.csv
  echo "== $v: $(tail -1 $O/bench_$v.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],2), "ms/gen")')"; grep "ev2_\|stochrank" $O/stats_$v.csv | head -7
done
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s"
