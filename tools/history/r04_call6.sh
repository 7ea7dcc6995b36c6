#!/bin/bash
# Round 4, sixth GPU call: the whole -m gpu suite as the driver runs it; the sharded-pass probe over the shared-memory transport
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_call6; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $O/gpu_suite.log 2>&1; echo "suite rc=$? $(tail -1 $O/gpu_suite.log)"
timeout 600 python tools/shard_probe.py > $O/shard_probe.txt 2> $O/shard_probe.err; echo "probe rc=$?"; cut -c1-420 $O/shard_probe.txt
