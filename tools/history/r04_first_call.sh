#!/bin/bash
# (HISTORICAL: the NLA_* environment switches this script sets exist only in -DNLA_DEBUG_SWITCHES builds since later in round 4; the
#  product reads nlopt_set_param values instead — "amd_mlsl_prefetch"; the tiled pair-distance kernel is the only one)
# Round 4, first GPU call: what round 3 wrote after its GPU minutes were spent.
#  (1) the pair-distance kernel's direct bit-for-bit test, both variants (tests/test_gpu_mlsl.py, NLA_TEST_EXPERIMENTAL=1);
#  (2) the MLSL files with the register-tiled variant forced on (NLA_MLSL_DIST2_TILED=1);
#  (3) the MLSL files with the sample-word prefetch forced on (NLA_MLSL_PREFETCH=1) and its A/B;
#  (4) A/B of config 4 (bench.py --workload mlsl): the sampling phase is where the distances are (6.4 of 7.5 ms per iteration).
# If (1) and (2) are green and (4) is faster: make the tiled kernel the default (hip/mlsl_kernels.hip, nla_k_mlsl_dist2) and drop
# the skip in the test.
#   gpurun --timeout 600 -- 'bash tools/history/r04_first_call.sh'
mkdir -p gpurun_out/r04_first
timeout 120 python -m pytest tests/test_gpu_mlsl.py -x -q -m gpu -k pair_distance 2>&1 | tail -3 | tee gpurun_out/r04_first/dist2_tests.log
NLA_MLSL_DIST2_TILED=1 timeout 300 python -m pytest tests/test_gpu_mlsl.py tests/test_gpu_exact_local.py tests/test_gpu_fullsize.py -q -m gpu -k "mlsl or MLSL" 2>&1 | tail -3 | tee gpurun_out/r04_first/mlsl_tiled.log
NLA_MLSL_PREFETCH=1 timeout 300 python -m pytest tests/test_gpu_mlsl.py tests/test_gpu_exact_local.py tests/test_gpu_fullsize.py tests/test_gpu_multiproc.py -q -m gpu -k "mlsl or MLSL" 2>&1 | tail -3 | tee gpurun_out/r04_first/mlsl_prefetch.log
for pf in 0 1 0 1; do
    NLA_MLSL_PREFETCH=$pf timeout 120 python bench.py --workload mlsl --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_first/bench_pf$pf.json
    python -c "
import json
d = json.load(open('gpurun_out/r04_first/bench_pf$pf.json'))
print('prefetch=$pf', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'))"
done 2>&1 | tee gpurun_out/r04_first/ab_prefetch.log
for t in 0 1 0 1; do
    NLA_MLSL_DIST2_TILED=$t timeout 120 python bench.py --workload mlsl --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_first/bench_t$t.json
    python -c "
import json
d = json.load(open('gpurun_out/r04_first/bench_t$t.json'))
print('tiled=$t', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'))"
done 2>&1 | tee gpurun_out/r04_first/ab.log
# multi-rank ISRES with the overlap mode on (the default), on the device
timeout 300 python -m pytest tests/test_gpu_multiproc.py -q -m gpu -k isres 2>&1 | tail -3 | tee gpurun_out/r04_first/isres_multirank.log
