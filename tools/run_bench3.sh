set -x
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_crs.json 2> gpurun_out/bench_crs.err
timeout 600 python bench.py --workload isres --steps 2 --warmup 1 > gpurun_out/bench_isres.json 2> gpurun_out/bench_isres.err
timeout 600 python bench.py --workload mlsl --steps 3 --warmup 1 > gpurun_out/bench_mlsl.json 2> gpurun_out/bench_mlsl.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mlsl_torchrun1.json 2> gpurun_out/bench_mlsl_torchrun1.err
tail -n 3 gpurun_out/bench_*.err
cat gpurun_out/bench_crs.json gpurun_out/bench_isres.json gpurun_out/bench_mlsl.json gpurun_out/bench_mlsl_torchrun1.json
