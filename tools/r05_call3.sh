#!/bin/bash
# Round 5, GPU call 3 (second session): the driver's suite command at HEAD (incl. the new steady-regime fixture test), the default bench
# line in its compact form + the detail file, kernel traces of the four workloads (headline, n = 512 windows incl. the HIP API trace,
# config 3, config 4).  Every step under a hard time limit.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c3; mkdir -p $O
date +%s > $O/t0
timeout -k 5 900 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
timeout -k 5 600 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$? line bytes $(tail -1 $O/bench.json | wc -c)"; tail -1 $O/bench.json | cut -c1-1500
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f $3 $4 $5 > $2; rm -rf $1; }
NB="--no-cpu-baseline --headline-only"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/kt -o crs -- python bench.py --steps 5 --warmup 1 $NB > $O/bench_under_rocprof.json 2> $O/kt.err; summ $O/kt $O/kernel_stats.csv
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/kn -o n512 -- python bench.py --n 512 --obj rastrigin --steps 3 --warmup 1 --evals-per-step 20000 $NB > $O/bench_n512_under_rocprof.json 2> $O/kn.err
f=$(find $O/kn -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/n512_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 400 120 > $O/n512_timeline.txt; rm -rf $O/kn
timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace -d $O/ha -o n512 -- python bench.py --n 512 --obj rastrigin --steps 2 --warmup 1 --evals-per-step 20000 $NB > /dev/null 2> $O/ha.err; summ $O/ha $O/n512_hip_api.csv --api 80
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/ki -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_isres_under_rocprof.json 2> $O/ki.err; summ $O/ki $O/isres_kernel_stats.csv
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/km -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mlsl_under_rocprof.json 2> $O/km.err; summ $O/km $O/mlsl_kernel_stats.csv
head -6 $O/kernel_stats.csv; head -8 $O/n512_kernel_stats.csv; head -12 $O/isres_kernel_stats.csv; head -10 $O/mlsl_kernel_stats.csv
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s"
