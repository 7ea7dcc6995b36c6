#!/bin/bash
# round 6: the driver's default bench command at the final code (+ the 20-step variant the driver used in earlier rounds)
mkdir -p gpurun_out/r06
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke.log 2>&1; tail -1 gpurun_out/r06/smoke.log | cut -c1-200
timeout 900 python bench.py --detail gpurun_out/r06/bench_detail.json > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err; tail -c 600 gpurun_out/r06/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline --detail gpurun_out/r06/bench_driver_steps_detail.json > gpurun_out/r06/bench_driver_steps.json 2>/dev/null; tail -c 300 gpurun_out/r06/bench_driver_steps.json
