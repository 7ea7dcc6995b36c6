#!/bin/bash
# round 3, the remaining 3.3 GPU minutes: is the ISRES overlap mode safe to make the default?  (1) its parity tests (overlap vs the
# one-stream run in the same process, 4 configurations incl. BASELINE config 3 at full size) in a loop of fresh processes — a race
# would be intermittent; (2) every ISRES-related test of the GPU suite with the mode forced on through the environment.
mkdir -p gpurun_out/r03_last2
t0=$(date +%s); n=0; bad=0
while [ $(( $(date +%s) - t0 )) -lt 65 ]; do
    n=$((n+1))
    NLA_TEST_EXPERIMENTAL=1 timeout 40 python -m pytest tests/test_gpu_isres.py -x -q -m gpu -k overlap > gpurun_out/r03_last2/loop_last.log 2>&1 || { bad=$((bad+1)); cp gpurun_out/r03_last2/loop_last.log gpurun_out/r03_last2/loop_fail_$n.log; }
done
echo "overlap parity loop: $n processes, $bad failed; last: $(tail -1 gpurun_out/r03_last2/loop_last.log)" | tee gpurun_out/r03_last2/loop.log
NLA_ISRES_OVERLAP=1 timeout 95 python -m pytest tests/test_gpu_isres.py tests/test_gpu_nan.py tests/test_gpu_stops.py tests/test_gpu_maximise.py tests/test_gpu_userobj.py \
    tests/test_gpu_fixed_dims.py tests/test_gpu_host_callbacks.py "tests/test_gpu_fullsize.py::test_config3_isres_n256_pop5e4_two_generations_against_the_reference" \
    -q -m gpu -k "not lbfgs and not LBFGS and not mma and not MMA and not mlsl and not MLSL and not crs_with" 2>&1 | tail -6 | tee gpurun_out/r03_last2/forced_on.log
