timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/gpu_tests.txt
for v in 832 816; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --gather-variant $v > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err; done
cat gpurun_out/gpu_tests.txt
