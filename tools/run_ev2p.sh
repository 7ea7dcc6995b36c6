cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ev2; timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ev2 -o ev2 -- python tools/isres_bench.py 256 50000 2 > gpurun_out/ev2_prof.txt 2>&1
f=$(find gpurun_out/prof_ev2 -name '*.db' | head -1); python profiles/summarize_rocpd.py $f > gpurun_out/ev2_kt.csv
find gpurun_out -name '*.db' -size +20M -delete
cat gpurun_out/ev2_kt.csv
