set -x
( time timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/gpu_tests.txt 2>&1
timeout 600 python tools/mlsl_bench.py 4096 1000 20000 > gpurun_out/mlsl_cfg4_b128.txt 2>&1
timeout 600 python tools/mlsl_bench.py 4096 1000 60000 cpu > gpurun_out/mlsl_cfg4_60k.txt 2>&1
cat gpurun_out/gpu_tests.txt gpurun_out/mlsl_cfg4_b128.txt gpurun_out/mlsl_cfg4_60k.txt
