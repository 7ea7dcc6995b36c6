#!/bin/bash
# the last GPU seconds of round 2: with the reference's summation order as the default for host callbacks, are the three ctest
# cases of testopt that differed (-a 24 / 21 / 23 -o 0) identical, and which lines of the seeded Python client differ
set +e
mkdir -p gpurun_out/last2
cd oracle/_ref
for a in 24 21 23; do
  timeout 5 ./testopt_ref -r 0 -a $a -o 0 | grep -v "finished after" > /tmp/r_$a.txt
  timeout 6 ./testopt_amd -r 0 -a $a -o 0 | grep -v "finished after" > /tmp/a_$a.txt
  if cmp -s /tmp/r_$a.txt /tmp/a_$a.txt; then echo "testopt -a $a -o 0: same"; else echo "testopt -a $a -o 0: DIFF"; diff /tmp/r_$a.txt /tmp/a_$a.txt | head -4; fi
done > ../../gpurun_out/last2/testopt.txt 2>&1
cd ../..
cat gpurun_out/last2/testopt.txt
export PYTHONPATH=tests/pyapi:.
NLOPT_AMD_PYAPI_LIBRARY=oracle/_ref/libnlopt_ref.so timeout 10 python tests/pyapi/seeded_runs.py > gpurun_out/last2/seeded_ref.txt 2>&1 &
timeout 12 python tests/pyapi/seeded_runs.py > gpurun_out/last2/seeded_amd.txt 2>&1
wait
diff gpurun_out/last2/seeded_ref.txt gpurun_out/last2/seeded_amd.txt > gpurun_out/last2/seeded.diff
echo "seeded diff lines: $(wc -l < gpurun_out/last2/seeded.diff)"; cut -c1-150 gpurun_out/last2/seeded.diff | head -24
