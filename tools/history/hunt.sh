#!/bin/bash
# GPU-box script: reproduce / bisect the intermittent CRS2_LM divergence of the round-2 driver run (development aid).
#   tools/history/hunt.sh <loopsA> <secondsB> <secondsC> <fullsuite 0|1>
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LA=${1:-20}; SB=${2:-180}; SC=${3:-0}; FULL=${4:-1}
echo "== phase A: the round-2 driver's process, $LA times (cobyla -> cpp_client -> crs in one process)" > gpurun_out/hunt.log
for i in $(seq 1 $LA); do
  NLA_TEST_KEEP_ORDER=1 timeout 300 python -m pytest tests/test_gpu_cobyla.py tests/test_gpu_cpp_client.py tests/test_gpu_crs.py -q -m gpu -p no:cacheprovider > gpurun_out/huntA_last.log 2>&1
  rc=$?
  echo "A loop $i rc=$rc $(tail -1 gpurun_out/huntA_last.log)" >> gpurun_out/hunt.log
  if [ $rc -ne 0 ]; then cp gpurun_out/huntA_last.log gpurun_out/huntA_fail_$i.log; fi
done
if [ "$SB" -gt 0 ]; then
  echo "== phase B: in-process stress, drawn configurations, churn, init dumps" >> gpurun_out/hunt.log
  timeout $((SB + 120)) python tools/stress_crs.py --seconds $SB --tag B --churn 1 --dump 1 >> gpurun_out/hunt.log 2>&1
fi
if [ "$SC" -gt 0 ]; then
  echo "== phase C: the same with one stream (NLA_ONE_STREAM=1)" >> gpurun_out/hunt.log
  NLA_ONE_STREAM=1 timeout $((SC + 120)) python tools/stress_crs.py --seconds $SC --tag C --churn 1 --dump 1 --seed 2 >> gpurun_out/hunt.log 2>&1
fi
if [ "$FULL" = "1" ]; then
  echo "== full -m gpu suite at HEAD (new collection order, no -x)" >> gpurun_out/hunt.log
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/full_suite.log 2>&1
  echo "full suite rc=$? $(tail -1 gpurun_out/full_suite.log)" >> gpurun_out/hunt.log
fi
tail -40 gpurun_out/hunt.log
