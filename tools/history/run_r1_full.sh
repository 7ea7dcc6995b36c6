set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/gpu_tests.txt 2>&1
timeout 400 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
rm -rf gpurun_out/prof_kt; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o crs -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
rm -rf gpurun_out/prof_fetch; timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fetch.json 2> gpurun_out/bench_fetch.err
rm -rf gpurun_out/prof_write; timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_write.json 2> gpurun_out/bench_write.err
for d in prof_kt prof_fetch prof_write; do find gpurun_out/$d -name '*.db' | head -3; done
f=$(find gpurun_out/prof_kt -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > gpurun_out/kt_summary.csv
f=$(find gpurun_out/prof_fetch -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --pmc > gpurun_out/fetch_summary.csv
f=$(find gpurun_out/prof_write -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --pmc > gpurun_out/write_summary.csv
find gpurun_out -name '*.db' -size +20M -delete
cat gpurun_out/gpu_tests.txt; cat gpurun_out/bench_full.json
