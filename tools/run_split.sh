timeout 900 python -m pytest tests/test_gpu_crs.py tests/test_gpu_kernels.py tests/test_gpu_multiproc.py -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_split$i.json 2> gpurun_out/bench_split.err
NLA_CRS_NOSPLIT=1 timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nosplit$i.json 2>> gpurun_out/bench_split.err
done
timeout 300 python bench.py --n 512 --obj rastrigin --evals-per-step 20000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_split_512.json 2>> gpurun_out/bench_split.err
NLA_CRS_NOSPLIT=1 timeout 300 python bench.py --n 512 --obj rastrigin --evals-per-step 20000 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_nosplit_512.json 2>> gpurun_out/bench_split.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_*split*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, round(d["value"]), round(d["ms_per_step"],2), round(r["frac"],3), r["launches"], round(r["avg_launch_ms"],4), d["window"])
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/bench_split.err
