/* tools/fmac_chain_probe.hip — what does one step of a chain of dependent v_fmac_f64 (DPP row_newbcast operand) cost on gfx950?  The reference-order sums
 * of the resident L-BFGS kernel (hip/lbfgs_resident.hip, lr_chain16 / lr_chain16x2) are such chains; bench.py prices that mode against this number.
 * One chain per wavefront, two chains side by side, and the v_readlane + v_add_f64 form of rounds 2-5; 4 wavefronts per workgroup (one per SIMD) as in the
 * kernel, 256 / 512 workgroups (one / two per compute unit).
 *   hipcc --offload-arch=gfx950 -O3 tools/fmac_chain_probe.hip -o tools/_build/fmac_chain_probe && tools/_build/fmac_chain_probe */
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define FD(acc, v, J) "v_fmac_f64_dpp " acc ", " v ", %[one] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n\t"
#define C1(J) FD("%[a]", "%[va]", J)
#define C2(J) FD("%[a]", "%[va]", J) FD("%[b]", "%[vb]", J)
#define ALL(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
__device__ __forceinline__ double lane_of(double v, int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); }
template <int MODE> __global__ __launch_bounds__(256) void k(const double *in, double *out, int groups)
{
    double va = in[threadIdx.x & 63], vb = in[64 + (threadIdx.x & 63)], a = 0., b = 0.;
    const double one = 1.0;
    for (int g = 0; g < groups; ++g) {
        if (MODE == 0) asm volatile("s_nop 1\n\t" ALL(C1) : [a] "+v"(a) : [va] "v"(va), [one] "v"(one));
        else if (MODE == 1) asm volatile("s_nop 1\n\t" ALL(C2) : [a] "+v"(a), [b] "+v"(b) : [va] "v"(va), [vb] "v"(vb), [one] "v"(one));
        else {
#pragma unroll
            for (int l = 0; l < 16; ++l) a += lane_of(va, l);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b;
}
int main()
{
    double h[128], *din, *dout;
    for (int i = 0; i < 128; ++i) h[i] = 1e-3 * (i + 1);
    CK(hipMalloc((void **) &din, sizeof h)); CK(hipMalloc((void **) &dout, 8 * 256 * 512)); CK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    const int groups = 1 << 16;                       /* 16 steps each: 1 M steps per chain */
    for (int wg = 256; wg <= 512; wg *= 2)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wg), dim3(256), 0, 0, din, dout, groups);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wg), dim3(256), 0, 0, din, dout, groups);
                else hipLaunchKernelGGL(k<2>, dim3(wg), dim3(256), 0, 0, din, dout, groups);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            const double steps = 16.0 * groups;
            printf("%d workgroups of 4 wavefronts, %s: %.3f ms for %.0f steps per chain = %.3f ns per step%s\n", wg,
                   mode == 0 ? "one chain of v_fmac_f64_dpp" : mode == 1 ? "two chains side by side" : "v_readlane x 2 + v_add_f64 (rounds 2-5)",
                   best, steps, 1e6 * best / steps, mode == 1 ? " (of either chain: two additions)" : "");
        }
    return 0;
}
