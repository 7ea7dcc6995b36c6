#!/bin/bash
# round-3 evidence on the MI355X box: the driver's exact GPU suite command, the default bench line, rocprofv3 kernel traces + PMC passes
# (headline CRS, L-BFGS / MLSL, the small-n CRS passes), the shard probe.     tools/history/r03_final.sh [suite 0|1]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
if [ "${1:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
  timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-120)"
fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f $3 > $2; rm -rf $1; }
HB="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o crs -- $HB > $O/bench_under_rocprof.json 2> $O/kt.err; summ $O/kt $O/kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fe -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/fe.err; summ $O/fe $O/pmc_fetch.csv --pmc
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/wr -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/wr.err; summ $O/wr $O/pmc_write.csv --pmc
MB="python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/mkt -o mlsl -- $MB > $O/bench_mlsl_under_rocprof.json 2> $O/mkt.err; summ $O/mkt $O/mlsl_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/mfe -o mlsl -- $MB > /dev/null 2> $O/mfe.err; summ $O/mfe $O/mlsl_pmc_fetch.csv --pmc
SB="python bench.py --n 512 --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/skt -o crs512 -- $SB > $O/bench_n512_under_rocprof.json 2> $O/skt.err; summ $O/skt $O/n512_kernel_stats.csv
timeout 120 $SB > $O/bench_n512.json 2>/dev/null
timeout 120 python bench.py --n 64 --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only > $O/bench_n64.json 2>/dev/null
timeout 400 python tools/shard_probe.py > $O/shard_probe.txt 2> $O/shard_probe.err
ls -la $O | head -30; head -5 $O/kernel_stats.csv; head -3 $O/pmc_fetch.csv; head -5 $O/mlsl_kernel_stats.csv; head -8 $O/n512_kernel_stats.csv; cat $O/shard_probe.txt | cut -c1-400; cut -c1-600 $O/bench_n512.json; cut -c1-400 $O/bench_n64.json
