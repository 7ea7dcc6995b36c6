for v in 832 816 432 1616 10832 10816 10864 11632; do
  timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --gather-variant $v > gpurun_out/bench_var_$v.json 2>> gpurun_out/bench_var.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bench_var_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, round(d["value"]), round(r["frac"],3), round(r["avg_launch_ms"],4))
    except Exception as e: print(f, "ERR", e)
PY
