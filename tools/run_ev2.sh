set -x
timeout 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_multiproc.py -x -q 2>&1 | tail -15 > gpurun_out/ev2_tests.txt
NLA_ISRES_DEBUG=1 timeout 600 python tools/isres_bench.py 256 50000 3 > gpurun_out/ev2_cfg3.txt 2>&1
NLA_ISRES_EVOLVE_SERIAL=1 timeout 600 python tools/isres_bench.py 256 50000 3 > gpurun_out/ev2_cfg3_serial.txt 2>&1
NLA_ISRES_DEBUG=1 timeout 600 python tools/isres_bench.py 64 20000 6 > gpurun_out/ev2_n64.txt 2>&1
NLA_ISRES_EVOLVE_SERIAL=1 timeout 600 python tools/isres_bench.py 64 20000 6 > gpurun_out/ev2_n64_serial.txt 2>&1
cat gpurun_out/ev2_tests.txt gpurun_out/ev2_cfg3.txt gpurun_out/ev2_cfg3_serial.txt gpurun_out/ev2_n64.txt gpurun_out/ev2_n64_serial.txt
