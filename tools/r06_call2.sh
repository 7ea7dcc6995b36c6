#!/bin/bash
# round 6, call 2: the column-sharded windows' first run — the new tests, then the probe
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu -k "windows" > gpurun_out/r06/windows_tests.txt 2>&1
tail -30 gpurun_out/r06/windows_tests.txt
timeout 600 python tools/shard_probe.py > gpurun_out/r06/shard_probe.txt 2>&1
cat gpurun_out/r06/shard_probe.txt
