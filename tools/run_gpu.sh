#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
for n in 512 64; do for fw in 0 1; do
NLA_CRS_FORWARD=$fw timeout -k 5 120 python bench.py --n $n --pop 100000 --obj rastrigin --evals-per-step 20000 --steps 3 --warmup 1 --no-cpu-baseline --headline-only > $O/bench_n${n}_fw$fw.json 2> $O/bench_n${n}_fw$fw.err
python - <<PY
import json
try:
    d=json.loads(open("$O/bench_n${n}_fw$fw.json").read().strip().splitlines()[-1])
    print("n=$n forward=$fw", round(d["value"]), "launch ms", d["roofline"]["avg_launch_ms"], "frac", d["roofline"]["frac"], d["window"])
except Exception as e:
    print("n=$n forward=$fw failed", e, open("$O/bench_n${n}_fw$fw.err").read()[-300:])
PY
done; done
