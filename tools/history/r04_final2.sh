#!/bin/bash
# Round 4, final evidence after the late changes (ISRES chain walk / scan, CRS doorbell, multi-rank failure flag): the driver's exact GPU
# suite command, smoke, the default bench line, the kernel statistics of config 3.  Every step under a hard time limit.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final2; mkdir -p $O
timeout -k 5 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-160)"
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.json
timeout -k 5 200 rocprofv3 --kernel-trace -d $O/i -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/i.err
f=$(find $O/i -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f > $O/isres_kernel_stats.csv; rm -rf $O/i
head -12 $O/isres_kernel_stats.csv
