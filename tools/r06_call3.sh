#!/bin/bash
mkdir -p gpurun_out/r06
timeout 120 tools/cu_mask_probe > gpurun_out/r06/cu_mask_probe.txt 2>&1; cat gpurun_out/r06/cu_mask_probe.txt
timeout 120 python tools/r06_dbg.py 2 '{"obj":"rosenbrock","n":512,"pop":5000,"seed":7,"maxeval":9000}' 2>&1 | tail -5
timeout 120 python tools/r06_dbg.py 2 '{"obj":"griewank","n":4096,"pop":4200,"seed":42,"maxeval":4500}' 2>&1 | tail -5
timeout 120 python tools/r06_dbg.py 2 '{"obj":"griewank","n":4096,"pop":100000,"seed":42,"maxeval":101000}' 2>&1 | tail -5
