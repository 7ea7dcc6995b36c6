#!/bin/bash
# Round 4, last GPU calls: the GPU test files that drive CRS2_LM through the conservative passes, at the final library (coherent pinned
# memory behind the doorbell), under a hard limit; full output kept
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call28; mkdir -p $O
python -X faulthandler -c "import nlopt_amd; print('devices', nlopt_amd.device_count())" > $O/import.log 2>&1; tail -3 $O/import.log
timeout -k 5 170 python -X faulthandler -m pytest tests/test_gpu_fullsize.py tests/test_gpu_stops.py tests/test_gpu_nan.py tests/test_gpu_zz_clients.py tests/test_gpu_multiproc.py -x -q -m gpu -k "crs or CRS or config1 or config2 or client or tutorial or bounded" -p no:cacheprovider > $O/crs_paths_full.log 2>&1; echo rc=$?; head -40 $O/crs_paths_full.log; tail -5 $O/crs_paths_full.log
