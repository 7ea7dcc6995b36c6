#!/bin/bash
# round 6: the driver's GPU tier — the whole -m gpu suite, smoke(), the default bench line
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/gpu_suite.log 2>&1
tail -15 gpurun_out/r06/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke.log 2>&1; tail -2 gpurun_out/r06/smoke.log
timeout 900 python bench.py --detail gpurun_out/r06/bench_detail.json > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err; tail -c 3000 gpurun_out/r06/bench.json
