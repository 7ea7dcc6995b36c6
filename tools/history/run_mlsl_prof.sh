cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for loc in lbfgs mma; do
  rm -rf gpurun_out/prof_mlsl_$loc
  timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mlsl_$loc -o mlsl -- python bench.py --workload mlsl --local $loc --no-cpu-baseline > gpurun_out/bench_mlsl_${loc}_prof.json 2> gpurun_out/bench_mlsl_${loc}_prof.err
  f=$(find gpurun_out/prof_mlsl_$loc -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > gpurun_out/mlsl_${loc}_kt_summary.csv
  head -8 gpurun_out/mlsl_${loc}_kt_summary.csv
done
find gpurun_out -name '*.db' -size +20M -delete
