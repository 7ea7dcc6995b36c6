set -x
timeout 300 python bench.py --n 512 --obj rastrigin --evals-per-step 20000 --steps 5 --warmup 1 --cpu-sample-trials 4000 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err
timeout 300 python bench.py --n 64 --obj rastrigin --evals-per-step 20000 --steps 5 --warmup 1 --cpu-sample-trials 4000 > gpurun_out/bench_n64.json 2> gpurun_out/bench_n64.err
NLA_ISRES_DEBUG=1 timeout 600 python tools/isres_bench.py 256 50000 3 > gpurun_out/isres_cfg3.txt 2>&1
timeout 900 python tools/mlsl_bench.py 4096 1000 20000 cpu > gpurun_out/mlsl_cfg4.txt 2>&1
tail -3 gpurun_out/bench_cfg2.json gpurun_out/bench_n64.json gpurun_out/isres_cfg3.txt gpurun_out/mlsl_cfg4.txt
