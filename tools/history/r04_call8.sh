#!/bin/bash
# Round 4, eighth GPU call: MLSL's commit walk with pts_update_newlm on the device (no batch x npts copy to the host), the single-chain
# reference-order sums: the MLSL / local-optimiser files, config 4 in both modes, the timeline of an iteration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mlsl.py tests/test_gpu_exact_local.py tests/test_gpu_fullsize.py tests/test_gpu_maximise.py tests/test_gpu_host_callbacks.py tests/test_gpu_multiproc.py tests/test_gpu_mma.py tests/test_gpu_cobyla.py tests/test_gpu_lbfgs.py -q -m gpu -k "mlsl or MLSL or lbfgs or LBFGS or local or resident" 2>&1 | tail -6 | tee $O/mlsl_tests.log
for m in "" "--exact"; do timeout 200 python bench.py --workload mlsl --no-cpu-baseline $m 2>/dev/null | tail -1 > $O/bench_mlsl$m.json; python -c "
import json
d = json.load(open('$O/bench_mlsl$m.json'))
print('$m', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'), d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['traffic'])"; done 2>&1 | tee $O/bench.log
timeout 300 rocprofv3 --kernel-trace -d $O/m -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/m.err
f=$(find $O/m -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --timeline > $O/mlsl_timeline.csv; rm -rf $O/m
