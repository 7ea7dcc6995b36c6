#!/bin/bash
# gather tiling sweep of the conservative CRS2_LM passes at n = 512 / 64 (bench.py --gather-variant WAVES*100+U)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03v
for v in 0 432 816 832 1616; do
  timeout 120 python bench.py --n 512 --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only --gather-variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=512 variant $v', round(d['value']), 'evals/s  gather ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
done
for v in 0 216 416 432 816; do
  timeout 120 python bench.py --n 64 --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only --gather-variant $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=64 variant $v', round(d['value']), 'evals/s  gather ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
done
