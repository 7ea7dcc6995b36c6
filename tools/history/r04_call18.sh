#!/bin/bash
# Round 4, eighteenth GPU call: what ends a block's walk, per phase, in the two chain kernels (development builds with -DEV2X_REASONS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call18; mkdir -p $O
for v in _rnew _rold; do echo "== lib$v"; NLOPT_AMD_LIB=$GRAFT_REPO_ROOT/nlopt_amd/lib/libnlopt_amd$v.so timeout -k 5 200 python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep "evolve phase" | tail -6; done 2>&1 | tee $O/reasons.log
