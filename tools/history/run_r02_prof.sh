#!/bin/bash
# round-2 evidence: the driver's default bench line, rocprofv3 kernel traces and separate PMC passes for the three workloads
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02prof3; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f $3 > $2; find $1 -name '*.db' -size +20M -delete; }
HB="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --headline-only"
rm -rf $O/kt; timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o crs -- $HB > $O/bench_under_rocprof.json 2> $O/kt.err; summ $O/kt $O/kernel_stats.csv
rm -rf $O/fe; timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/fe -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/fe.err; summ $O/fe $O/pmc_fetch.csv --pmc
rm -rf $O/wr; timeout 400 rocprofv3 --pmc WRITE_SIZE -d $O/wr -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/wr.err; summ $O/wr $O/pmc_write.csv --pmc
rm -rf $O/sq; timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $O/sq -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/sq.err; summ $O/sq $O/pmc_sq.csv --pmc
rm -rf $O/vb; timeout 400 rocprofv3 --pmc VALUBusy MemUnitBusy -d $O/vb -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only > /dev/null 2> $O/vb.err; summ $O/vb $O/pmc_valubusy.csv --pmc
# the conservative passes for comparison (same counters)
rm -rf $O/kt0; NLA_CRS_FORWARD=0 timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt0 -o crs -- $HB > $O/bench_nochain_under_rocprof.json 2> $O/kt0.err; summ $O/kt0 $O/kernel_stats_nochain.csv
# ISRES (config 3) and MLSL (config 4)
IB="python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $O/ikt; timeout 400 rocprofv3 --kernel-trace --stats -d $O/ikt -o isres -- $IB > $O/bench_isres_under_rocprof.json 2> $O/ikt.err; summ $O/ikt $O/isres_kernel_stats.csv
rm -rf $O/ife; timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/ife -o isres -- $IB > /dev/null 2> $O/ife.err; summ $O/ife $O/isres_pmc_fetch.csv --pmc
rm -rf $O/isq; timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $O/isq -o isres -- $IB > /dev/null 2> $O/isq.err; summ $O/isq $O/isres_pmc_sq.csv --pmc
MB="python bench.py --workload mlsl --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $O/mkt; timeout 400 rocprofv3 --kernel-trace --stats -d $O/mkt -o mlsl -- $MB > $O/bench_mlsl_under_rocprof.json 2> $O/mkt.err; summ $O/mkt $O/mlsl_kernel_stats.csv
rm -rf $O/mfe; timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/mfe -o mlsl -- $MB > /dev/null 2> $O/mfe.err; summ $O/mfe $O/mlsl_pmc_fetch.csv --pmc
rm -rf $O/msq; timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $O/msq -o mlsl -- $MB > /dev/null 2> $O/msq.err; summ $O/msq $O/mlsl_pmc_sq.csv --pmc
rm -rf $O/kt $O/fe $O/wr $O/sq $O/vb $O/kt0 $O/ikt $O/ife $O/isq $O/mkt $O/mfe $O/msq
ls -la $O | head -40
head -6 $O/kernel_stats.csv; head -4 $O/pmc_fetch.csv; head -4 $O/pmc_valubusy.csv; head -8 $O/isres_kernel_stats.csv; head -6 $O/mlsl_kernel_stats.csv
