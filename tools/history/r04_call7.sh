#!/bin/bash
# Round 4, seventh GPU call: the resident kernel in the reference's summation order ("amd_exact_dot" = 1), its bench line, the phase
# profile, and the PMC passes (FETCH_SIZE / WRITE_SIZE for the resident kernel's HBM traffic; VALUBusy + MemUnitBusy for the three workloads)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_lbfgs.py -x -q -m gpu 2>&1 | tail -6 | tee $O/lbfgs_tests.log
timeout 900 python -m pytest tests/test_gpu_mlsl.py tests/test_gpu_exact_local.py tests/test_gpu_fullsize.py tests/test_gpu_maximise.py tests/test_gpu_host_callbacks.py tests/test_gpu_isres.py -q -m gpu -k "mlsl or MLSL or lbfgs or LBFGS or local or 2pow20" 2>&1 | tail -6 | tee $O/mlsl_tests.log
NLOPT_AMD_LIB=nlopt_amd/lib/libnlopt_amd_prof.so timeout 300 python tools/lbfgs_prof.py 2 exact > $O/lbfgs_prof_resident_exact.txt 2>&1; cat $O/lbfgs_prof_resident_exact.txt
for m in "" "--exact"; do timeout 200 python bench.py --workload mlsl --no-cpu-baseline $m 2>/dev/null | tail -1 > $O/bench_mlsl$m.json; python -c "
import json
d = json.load(open('$O/bench_mlsl$m.json'))
print('$m', round(d['value']), 'evals/s', round(d['ms_per_step'], 2), 'ms/iteration', d.get('phases'), d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done 2>&1 | tee $O/bench.log
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f $3 > $2; rm -rf $1; }
MB="python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline"
IB="python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline"
HB="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --headline-only"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/mkt -o mlsl -- $MB > /dev/null 2> $O/mkt.err; summ $O/mkt $O/mlsl_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/mfe -o mlsl -- $MB > /dev/null 2> $O/mfe.err; summ $O/mfe $O/mlsl_pmc_fetch.csv --pmc
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/mwr -o mlsl -- $MB > /dev/null 2> $O/mwr.err; summ $O/mwr $O/mlsl_pmc_write.csv --pmc
timeout 300 rocprofv3 --pmc VALUBusy MemUnitBusy -d $O/mvb -o mlsl -- $MB > /dev/null 2> $O/mvb.err; summ $O/mvb $O/mlsl_pmc_busy.csv --pmc
timeout 300 rocprofv3 --pmc VALUBusy MemUnitBusy -d $O/ivb -o isres -- $IB > /dev/null 2> $O/ivb.err; summ $O/ivb $O/isres_pmc_busy.csv --pmc
timeout 300 rocprofv3 --pmc VALUBusy MemUnitBusy -d $O/cvb -o crs -- $HB > /dev/null 2> $O/cvb.err; summ $O/cvb $O/crs_pmc_busy.csv --pmc
head -4 $O/mlsl_pmc_fetch.csv $O/mlsl_pmc_write.csv $O/mlsl_pmc_busy.csv $O/isres_pmc_busy.csv $O/crs_pmc_busy.csv; head -5 $O/mlsl_kernel_stats.csv
