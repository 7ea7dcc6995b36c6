#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
for ev in 0 4 8 16; do
NLA_CRS_RNG_CU_EVERY=$ev timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --headline-only > $O/bench_cu$ev.json 2> $O/bench_cu$ev.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_cu$ev.json").read().strip().splitlines()[-1])
    print("rng on every $ev-th CU:", round(d["value"]), "frac", round(d["roofline"]["frac"],3), "launch ms", round(d["roofline"]["avg_launch_ms"],3))
except Exception as e:
    print("every=$ev failed", e, open("$O/bench_cu$ev.err").read()[-400:])
PY
done
