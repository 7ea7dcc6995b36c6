cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > gpurun_out/gpu_all.txt 2>&1; grep -E "passed|failed|real" gpurun_out/gpu_all.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -2 gpurun_out/smoke.txt
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
rm -rf gpurun_out/prof_kt; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o crs -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err
rm -rf gpurun_out/prof_fetch; timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fetch.json 2> gpurun_out/bench_fetch.err
rm -rf gpurun_out/prof_write; timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o crs -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_write.json 2> gpurun_out/bench_write.err
f=$(find gpurun_out/prof_kt -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > gpurun_out/kt_summary.csv
f=$(find gpurun_out/prof_fetch -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --pmc > gpurun_out/fetch_summary.csv
f=$(find gpurun_out/prof_write -name '*.db' | head -1); python profiles/summarize_rocpd.py $f --pmc > gpurun_out/write_summary.csv
find gpurun_out -name '*.db' -size +20M -delete
head -5 gpurun_out/kt_summary.csv; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['speedup_vs_cpu_single_thread'])"
