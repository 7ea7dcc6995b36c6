#!/bin/bash
# Round 4, final evidence on the MI355X box: the driver's exact GPU suite command, smoke, the default bench line, rocprofv3 kernel trace +
# FETCH / WRITE passes of the headline command, the host-side API trace of the n = 512 passes.  Every step under a hard time limit.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_final; mkdir -p $O
timeout -k 5 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-160)"
timeout -k 5 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f $3 $4 > $2; rm -rf $1; }
HB="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/kt -o crs -- $HB > $O/bench_under_rocprof.json 2> $O/kt.err; summ $O/kt $O/kernel_stats.csv
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $O/fe -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/fe.err; summ $O/fe $O/pmc_fetch.csv --pmc
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $O/wr -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/wr.err; summ $O/wr $O/pmc_write.csv --pmc
timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace -d $O/ha -o n512 -- python bench.py --n 512 --obj rastrigin --steps 2 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only > /dev/null 2> $O/ha.err; summ $O/ha $O/n512_hip_api.csv --api 60
head -5 $O/kernel_stats.csv; head -4 $O/pmc_fetch.csv | cut -c1-200; head -40 $O/n512_hip_api.csv | cut -c1-160
