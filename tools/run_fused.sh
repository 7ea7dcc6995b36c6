timeout 900 python -m pytest tests/test_gpu_crs.py tests/test_gpu_kernels.py tests/test_gpu_dropin.py tests/test_gpu_multiproc.py -x -q 2>&1 | tail -5
for v in fused unfused; do
  if [ $v = unfused ]; then export NLA_CRS_UNFUSED=1; else unset NLA_CRS_UNFUSED; fi
  timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${v}_4096.json 2>> gpurun_out/bench_fused.err
  timeout 300 python bench.py --n 512 --obj rastrigin --evals-per-step 20000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${v}_512.json 2>> gpurun_out/bench_fused.err
  timeout 300 python bench.py --n 64 --obj rastrigin --evals-per-step 20000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${v}_64.json 2>> gpurun_out/bench_fused.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_*fused_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, round(d["value"]), round(d["ms_per_step"],2), round(r["frac"],3), r["launches"], round(r["avg_launch_ms"],4), d["minf"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/bench_fused.err
