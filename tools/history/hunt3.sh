#!/bin/bash
# GPU-box script: is it the uncached allocations?  In-process loops over the golden CRS cases (tools/stress_crs.py --golden-only)
#   1  product as fixed (uncached blocks pooled, never freed) + deliberate raw uncached alloc/write/free between runs  -> provokes it if that is the cause
#   2  product as fixed, no provocation                                                                              -> must be clean
#   3  round-2 behaviour (NLA_UC_POOL=0 + uncached buffers for every run: NLA_CRS_FORWARD_MEM=1), no provocation      -> the baseline rate
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
S=${1:-100}
echo "== hunt3" > gpurun_out/hunt3.log
timeout $((S + 90)) python tools/stress_crs.py --seconds $S --tag H3_1_fixed_plus_ucchurn --golden-only 1 --uc-churn 300 --churn 1 2>&1 | grep -v "^DIVERGENCE" | tail -3 >> gpurun_out/hunt3.log
timeout $((S + 90)) python tools/stress_crs.py --seconds $S --tag H3_2_fixed --golden-only 1 --churn 1 2>&1 | grep -v "^DIVERGENCE" | tail -3 >> gpurun_out/hunt3.log
NLA_UC_POOL=0 NLA_UC_ALWAYS=1 timeout $((S + 90)) python tools/stress_crs.py --seconds $S --tag H3_3_round2 --golden-only 1 --churn 1 2>&1 | grep -v "^DIVERGENCE" | tail -3 >> gpurun_out/hunt3.log
cat gpurun_out/hunt3.log
