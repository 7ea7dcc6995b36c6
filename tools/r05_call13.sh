#!/bin/bash
# Round 5, GPU call 13: (a) the ranking pipeline's prefetch point (tick of a block at which the next block's inputs are loaded: 16 / 24 / 32 / 40 / 48,
# variant builds); (b) the round's profiles at the final kernels: headline kernel trace + FETCH_SIZE / WRITE_SIZE passes, the n = 512 windows
# (kernel trace, timeline, HIP API trace), n = 64 kernel trace, config 3 kernel trace + VALUBusy.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c13; mkdir -p $O
date +%s > $O/t0
line() {   # line <label> <bench args...>
    local label=$1; shift
    timeout -k 5 150 python bench.py --detail $O/last_detail.json --full-line "$@" 2>/dev/null | tail -1 > $O/last.json
    python - "$label" "$O/last.json" <<'PY' | tee -a $O/ab.log
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d.get("roofline") or {}
    ph = d.get("phases") or {}
    print("%-44s %9.0f evals/s  %8.3f ms/step  pipeline %.3f ms/launch %.1f ns/tick  %s" % (sys.argv[1], d["value"], d["ms_per_step"], r.get("avg_launch_ms") or 0, r.get("achieved") or 0,
          {k: (round(v * 1e3, 2) if k.endswith("_s_per_gen") else v) for k, v in ph.items() if k.endswith("_s_per_gen")}))
except Exception as e:
    print(sys.argv[1], "FAILED", repr(e))
PY
}
for rep in 1 2; do
  line "isres config 3, next inputs loaded at tick 32" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  for k in 16 24 40 48; do
    NLOPT_AMD_LIB=$PWD/nlopt_amd/lib/libnlopt_amd_pf$k.so line "isres config 3, next inputs loaded at tick $k" --workload isres --no-cpu-baseline --steps 3 --warmup 1
  done
done
summ() { f=$(find $1 -name '*.db' | head -1); [ -n "$f" ] && python profiles/summarize_rocpd.py $f $3 $4 $5 > $2; rm -rf $1; }
NB="--no-cpu-baseline --headline-only"
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/kt -o crs -- python bench.py --steps 5 --warmup 1 $NB > $O/bench_under_rocprof.json 2> $O/kt.err; summ $O/kt $O/kernel_stats.csv
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE -d $O/fe -o crs -- python bench.py --steps 2 --warmup 1 $NB > /dev/null 2> $O/fe.err; summ $O/fe $O/pmc_fetch.csv --pmc
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE -d $O/wr -o crs -- python bench.py --steps 2 --warmup 1 $NB > /dev/null 2> $O/wr.err; summ $O/wr $O/pmc_write.csv --pmc
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/kn -o n512 -- python bench.py --n 512 --obj rastrigin --steps 4 --warmup 1 --evals-per-step 20000 $NB > $O/bench_n512_under_rocprof.json 2> $O/kn.err
f=$(find $O/kn -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > $O/n512_kernel_stats.csv; python profiles/summarize_rocpd.py $f --timeline 400 100 > $O/n512_timeline.txt; rm -rf $O/kn
timeout -k 5 200 rocprofv3 --hip-trace --kernel-trace -d $O/ha -o n512 -- python bench.py --n 512 --obj rastrigin --steps 2 --warmup 1 --evals-per-step 20000 $NB > /dev/null 2> $O/ha.err; summ $O/ha $O/n512_hip_api.csv --api 60
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/k6 -o n64 -- python bench.py --n 64 --obj rastrigin --steps 4 --warmup 1 --evals-per-step 20000 $NB > $O/bench_n64_under_rocprof.json 2> $O/k6.err; summ $O/k6 $O/n64_kernel_stats.csv
timeout -k 5 200 rocprofv3 --kernel-trace --stats -d $O/ki -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_isres_under_rocprof.json 2> $O/ki.err; summ $O/ki $O/isres_kernel_stats.csv
timeout -k 5 200 rocprofv3 --pmc VALUBusy MemUnitBusy -d $O/vi -o isres -- python bench.py --workload isres --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/vi.err; summ $O/vi $O/isres_pmc_valubusy.csv --pmc
head -5 $O/kernel_stats.csv; head -4 $O/pmc_fetch.csv | cut -c1-160; head -3 $O/pmc_write.csv | cut -c1-160; head -6 $O/n512_kernel_stats.csv; head -5 $O/n64_kernel_stats.csv; head -10 $O/isres_kernel_stats.csv; grep -i "stochrank\|ev2_\|rankbits" $O/isres_pmc_valubusy.csv | head -8
echo "elapsed $(( $(date +%s) - $(cat $O/t0) )) s" | tee -a $O/ab.log
