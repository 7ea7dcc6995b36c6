#!/bin/bash
# Round 4, fifth GPU call: the whole -m gpu suite exactly as the driver runs it (after the resident L-BFGS kernel, the pooled
# uncached memory rewrite, the switches moved behind a build flag, the lifted ISRES population limit), the stand-alone uncached-memory
# experiment, and the default bench line (config 4 in both summation modes, the ISRES CPU baseline at the benchmark's population).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call5; mkdir -p $O
timeout 1200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
timeout 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-160)"
timeout 300 tools/_build/uc_stale_repro 300 6 > $O/uc_stale_repro.txt 2>&1; echo "uc repro rc=$?"; cat $O/uc_stale_repro.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json
