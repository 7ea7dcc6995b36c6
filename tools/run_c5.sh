timeout 900 python bench.py --pop 1000000 --steps 5 --warmup 1 --evals-per-step 4000 --no-cpu-baseline > gpurun_out/bench_cfg5_1gpu.json 2> gpurun_out/bench_cfg5.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_cfg5_1gpu.json")); r=d["roofline"]; print(d["value"], d["ms_per_step"], r["frac"], r["launches"], r["avg_launch_ms"], r["avg_trials_consumed_per_launch"], d["window"], d["init"])
PY
tail -2 gpurun_out/bench_cfg5.err
