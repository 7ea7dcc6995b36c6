#!/bin/bash
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_multiproc.py -x -q -m gpu -k "windows" > gpurun_out/r06/windows_tests.txt 2>&1
tail -30 gpurun_out/r06/windows_tests.txt
timeout 120 python tools/r06_dbg.py 2 '{"obj":"griewank","n":4096,"pop":100000,"seed":42,"maxeval":101000}' 2>&1 | tail -5
timeout 600 python tools/shard_probe.py > gpurun_out/r06/shard_probe.txt 2>&1
cat gpurun_out/r06/shard_probe.txt
