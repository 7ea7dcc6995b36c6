#!/bin/bash
# Round 4, third GPU call: the resident L-BFGS kernel (hip/lbfgs_resident.hip) on the device for the first time:
#  (1) bit-for-bit against the streaming kernel + the L-BFGS / MLSL / exact-order / full-size / maximise files
#  (2) its phase profile (instrumented build) and the config-4 bench, A/B against the streaming kernel ("amd_lbfgs_streaming")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lbfgs.py -x -q -m gpu 2>&1 | tail -15 | tee $O/lbfgs_tests.log
timeout 900 python -m pytest tests/test_gpu_mlsl.py tests/test_gpu_exact_local.py tests/test_gpu_fullsize.py tests/test_gpu_maximise.py tests/test_gpu_multiproc.py tests/test_gpu_host_callbacks.py -q -m gpu -k "mlsl or MLSL or lbfgs or LBFGS or local" 2>&1 | tail -8 | tee $O/mlsl_tests.log
NLOPT_AMD_LIB=nlopt_amd/lib/libnlopt_amd_prof.so timeout 120 python tools/lbfgs_prof.py 2 > $O/lbfgs_prof_resident.txt 2>&1; cat $O/lbfgs_prof_resident.txt
for i in 1 2; do timeout 120 python bench.py --workload mlsl --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_mlsl_$i.json; python -c "
import json
d = json.load(open('$O/bench_mlsl_$i.json'))
print(round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'), d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done 2>&1 | tee $O/bench.log
