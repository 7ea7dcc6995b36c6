#!/bin/bash
# Round 4, second GPU call: measure before rewriting.
#  (1) tools/lbfgs_prof.py over the instrumented library: per-phase device time of the batched L-BFGS searches at config 4 (fast, exact)
#  (2) kernel TIMELINES (rocprofv3 --kernel-trace, profiles/summarize_rocpd.py --timeline) of config 4, config 3 and CRS n=512:
#      idle gaps and overlap, which the per-kernel sums do not show
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call2; mkdir -p $O
NLOPT_AMD_LIB=nlopt_amd/lib/libnlopt_amd_prof.so timeout 120 python tools/lbfgs_prof.py 2 > $O/lbfgs_prof_fast.txt 2>&1; cat $O/lbfgs_prof_fast.txt
NLOPT_AMD_LIB=nlopt_amd/lib/libnlopt_amd_prof.so timeout 300 python tools/lbfgs_prof.py 2 exact > $O/lbfgs_prof_exact.txt 2>&1; cat $O/lbfgs_prof_exact.txt
tl() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && { python profiles/summarize_rocpd.py $f > $2_stats.csv; python profiles/summarize_rocpd.py $f --timeline > $2_timeline.csv; }; rm -rf $1; }
timeout 300 rocprofv3 --kernel-trace -d $O/m -o mlsl -- python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_mlsl.json 2> $O/m.err; tl $O/m $O/mlsl
timeout 300 rocprofv3 --kernel-trace -d $O/i -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_isres.json 2> $O/i.err; tl $O/i $O/isres
timeout 300 rocprofv3 --kernel-trace -d $O/c -o crs512 -- python bench.py --n 512 --obj rastrigin --steps 2 --warmup 1 --evals-per-step 20000 --no-cpu-baseline --headline-only > $O/bench_n512.json 2> $O/c.err; tl $O/c $O/n512
ls -la $O; head -12 $O/mlsl_stats.csv; head -14 $O/isres_stats.csv
