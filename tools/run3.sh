for v in 816 832 432 10832; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --gather-variant $v > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err; done
for k in 12 16 48; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --max-spec $k > gpurun_out/bench_k$k.json 2> gpurun_out/bench_k$k.err; done
