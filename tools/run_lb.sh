timeout 900 python -m pytest tests/test_gpu_lbfgs.py tests/test_gpu_mlsl.py tests/test_gpu_multiproc.py -x -q 2>&1 | tail -5
timeout 600 python bench.py --workload mlsl --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mlsl_regs.json 2> gpurun_out/bench_mlsl_regs.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_mlsl_regs.json")); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["phases"], d["minf"])
PY
