/* lbfgs_resident32.hip — the resident batched LD_LBFGS search (lbfgs_resident.hip, the same source) with 32 coordinates per thread:
 * 4096 < n <= 8192 with a compiled-in device objective.  x and the gradient take 2 x 64 KB of the compute unit's 160 KB of LDS (one
 * workgroup per compute unit), the search direction 64 VGPRs per thread.  Round 6: until then these dimensions ran on the streaming
 * kernel (lbfgs_kernels.hip), whose vector loops walk global memory and which spills registers (0.05 of HBM at
 * n = 8192).  Bit-identical to the streaming kernel in both summation modes (tests/test_gpu_lbfgs.py, tools/lbfgs_emu_check.py). */
#define LR_WIDE 1
#include "lbfgs_resident.hip"
