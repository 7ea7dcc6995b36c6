#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
rm -rf $O/kt; timeout 400 rocprofv3 --kernel-trace -d $O/kt -o crs -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only > $O/bench_under_rocprof.json 2> $O/kt.err
f=$(find $O/kt -name '*.db' | head -1); python tools/chain_overlap.py $f | tee $O/chain_overlap.txt
rm -rf $O/kt
