set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/gpu_tests.txt
for v in 816 832 10816 10832 10864 11632; do timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --gather-variant $v > gpurun_out/bench_v$v.json 2> gpurun_out/bench_v$v.err; done
rm -rf gpurun_out/prof2; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d gpurun_out/prof2 -o crs -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof2.json 2> gpurun_out/bench_prof2.err
ls -la gpurun_out/prof2 | head
cat gpurun_out/gpu_tests.txt
