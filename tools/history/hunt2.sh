#!/bin/bash
# GPU-box script: A/B the intermittent CRS2_LM divergence.  Round-robin over variants of the round-2 driver's process
# (cobyla -> cpp_client -> crs in one pytest process), LOOPS times each:   tools/history/hunt2.sh <loops> "<VAR=1>" "<VAR=1>" ...
# ("-" = baseline, no variable)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
LOOPS=$1; shift
echo "== hunt2: $LOOPS loops each of: $*" > gpurun_out/hunt2.log
for i in $(seq 1 $LOOPS); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then ev=""; else ev="$v"; fi
    env $ev NLA_TEST_KEEP_ORDER=1 timeout 300 python -m pytest tests/test_gpu_cobyla.py tests/test_gpu_cpp_client.py tests/test_gpu_crs.py -q -m gpu -p no:cacheprovider > gpurun_out/hunt2_last.log 2>&1
    rc=$?
    echo "variant [$v] loop $i rc=$rc $(tail -1 gpurun_out/hunt2_last.log)" >> gpurun_out/hunt2.log
    if [ $rc -ne 0 ]; then cp gpurun_out/hunt2_last.log "gpurun_out/hunt2_fail_${i}_$(echo $v | tr -c 'A-Za-z0-9' '_').log"; fi
  done
done
for v in "$@"; do echo "variant [$v]: $(grep -F "variant [$v]" gpurun_out/hunt2.log | grep -c 'rc=0') ok, $(grep -F "variant [$v]" gpurun_out/hunt2.log | grep -vc 'rc=0') failed" >> gpurun_out/hunt2.log; done
tail -8 gpurun_out/hunt2.log
