/* tools/cu_mask_probe.hip — which compute units does a hipExtStreamCreateWithCUMask stream really use?
 * A grid of short workgroups records (XCC_ID, HW_ID.se / sh / cu) per workgroup; the distinct units are counted for an unmasked stream
 * and for masks "every k-th bit" (k = 2, 4, 8) — what nla_stream_create_cu_share builds — and a plain streaming read of 2 GB is timed on
 * each (does a share of the CUs limit the bandwidth a kernel can draw?).
 *   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe && tools/cu_mask_probe */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(uint32_t *out)
{
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);       /* HW_REG_HW_ID */
        const uint32_t xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);     /* HW_REG_XCC_ID */
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    for (volatile int i = 0; i < 2000; ++i) { }
}
__global__ __launch_bounds__(256) void stream_kernel(const double2 *__restrict__ a, size_t n, double *out)
{
    double s = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 123.456) *out = s;
}
int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32, blocks = 8192;
    uint32_t *d = nullptr; std::vector<uint32_t> h(2 * blocks);
    double2 *big = nullptr; double *dout = nullptr;
    const size_t nbig = (size_t) 1 << 27;                    /* 2 GiB of double2 */
    CK(hipMalloc((void **) &d, 8 * blocks)); CK(hipMalloc((void **) &big, nbig * sizeof(double2))); CK(hipMalloc((void **) &dout, 8));
    CK(hipMemset(big, 0, nbig * sizeof(double2)));
    printf("device reports %d compute units\n", ncu);
    for (int parts = 1; parts <= 8; parts *= 2) {
        hipStream_t st;
        if (parts == 1) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        else {
            uint32_t mask[64] = {0};
            for (int i = 0; i < ncu; ++i) if (i % parts == 0) mask[i >> 5] |= 1u << (i & 31);
            CK(hipExtStreamCreateWithCUMask(&st, (uint32_t) words, mask));
        }
        hipLaunchKernelGGL(where_kernel, dim3(blocks), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost));
        std::set<uint64_t> cus, xccs;
        for (int b = 0; b < blocks; ++b) {
            const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
            const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            cus.insert(((uint64_t) xcc << 16) | (se << 8) | (sh << 4) | cu); xccs.insert(xcc);
        }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(stream_kernel, dim3(ncu * 8), dim3(256), 0, st, big, nbig, dout);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("every %d-th CU bit: %zu distinct compute units on %zu XCCs; 2 GiB streaming read %.3f ms = %.2f TB/s\n", parts, cus.size(), xccs.size(), best,
               (double) nbig * 16 / 1e9 / best);
        CK(hipStreamDestroy(st));
    }
    return 0;
}
