cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c5; timeout 900 rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/prof_c5 -o c5 -- python bench.py --pop 1000000 --steps 2 --warmup 1 --evals-per-step 4000 --no-cpu-baseline > gpurun_out/bench_c5p.json 2> gpurun_out/bench_c5p.err
f=$(find gpurun_out/prof_c5 -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > gpurun_out/c5_kt.csv
find gpurun_out -name '*.db' -size +30M -delete
cat gpurun_out/c5_kt.csv
