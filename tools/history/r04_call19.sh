#!/bin/bash
# Round 4, nineteenth GPU call: the register walk's sum of mutated coordinates against the direct sum (development build, -DEV2X_REASONS)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call19; mkdir -p $O
NLOPT_AMD_LIB=$GRAFT_REPO_ROOT/nlopt_amd/lib/libnlopt_amd_rnew.so timeout -k 5 100 python bench.py --workload isres --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep -A1 "evolve phase" | tail -8 | tee $O/reasons.log
