set -x
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_tests.txt
for v in 816 832 416; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --gather-variant $v > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err; done
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --max-spec 64 > gpurun_out/bench_k64.json 2> gpurun_out/bench_k64.err
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --max-spec 24 > gpurun_out/bench_k24.json 2> gpurun_out/bench_k24.err
cat gpurun_out/gpu_tests.txt
