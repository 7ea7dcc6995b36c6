#!/bin/bash
# round 6, last session: the device tests of the kernels changed at the end of the round (ISRES evolve hand-over inside the scan launch, CRS2_LM
# windows with the lighter fences) repeated on one box — a race would show as a run that differs from the oracle once in a while
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O; : > $O/repeat_tests.txt
for i in 1 2 3 4 5 6; do
  timeout -k 5 900 python -m pytest tests/test_gpu_isres.py tests/test_gpu_crs.py tests/test_gpu_crs_windows.py tests/test_gpu_fullsize.py tests/test_gpu_zz_uncached_stress.py -x -q -p no:cacheprovider 2>&1 | tail -1 >> $O/repeat_tests.txt
done
cat $O/repeat_tests.txt
