#!/bin/bash
# scratch runner for one gpurun call (edited per experiment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
timeout -k 5 100 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "mt or stream or init" --timeout 60 2>&1 | tail -n 2
timeout -k 5 200 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_isres_seg.json 2> $O/bench_isres_seg.err
timeout -k 5 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --headline-only > $O/bench_crs_seg.json 2> $O/bench_crs_seg.err
timeout -k 5 200 python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mlsl_seg.json 2> $O/bench_mlsl_seg.err
python - <<PY
import json
for w in ("isres","crs","mlsl"):
    try:
        d=json.loads(open("$O/bench_%s_seg.json" % w).read().strip().splitlines()[-1])
        print(w, round(d["value"]), round(d["ms_per_step"],2), d.get("phases", d.get("init")))
    except Exception as e:
        print(w, "failed", e, open("$O/bench_%s_seg.err" % w).read()[-500:])
PY
