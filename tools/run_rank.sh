timeout 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_multiproc.py -x -q > gpurun_out/rank_tests.txt 2>&1; grep -E "passed|failed" gpurun_out/rank_tests.txt
timeout 600 python bench.py --workload isres --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_isres3.json 2> gpurun_out/bench_isres3.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_isres3.json")); print(d["value"], d["ms_per_step"], d["phases"], d["minf"])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ev3; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ev3 -o ev3 -- python tools/isres_bench.py 256 50000 2 > gpurun_out/ev3_prof.txt 2>&1
f=$(find gpurun_out/prof_ev3 -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > gpurun_out/ev3_kt.csv
find gpurun_out -name '*.db' -size +20M -delete
head -14 gpurun_out/ev3_kt.csv
