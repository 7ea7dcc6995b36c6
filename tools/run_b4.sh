timeout 600 python bench.py --workload mlsl --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mlsl_b320.json 2> gpurun_out/bench_mlsl_b320.err
timeout 600 python bench.py --workload isres --steps 3 --warmup 1 > gpurun_out/bench_isres2.json 2> gpurun_out/bench_isres2.err
timeout 600 python -m pytest tests/test_gpu_mlsl.py -x -q 2>&1 | tail -3
python - <<'PY'
import json
for f in ("bench_mlsl_b320","bench_isres2"):
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["phases"], d.get("cpu_baseline",{}).get("value"))
PY
