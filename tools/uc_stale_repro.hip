// tools/uc_stale_repro.hip — stand-alone form of the experiment behind devrt.hip's uncached-memory pool (DESIGN.md "the intermittent
// divergence"): does memory that was mapped UNCACHED (hipDeviceMallocUncached), written by a kernel and given back to the driver, show
// stale contents when it comes back as an ORDINARY allocation?  Each round: A) uncached block, a kernel fills it with pattern P1,
// hipFree; B) ordinary blocks of the same size (the driver tends to hand the pages straight back), filled with P2 by one of three
// writers (kernel / hipMemcpy H2D / hipMemset + kernel add); C) a second kernel and a D2H copy read them back; every 64-byte line that is
// not P2 is counted, lines that still hold P1 separately.  A control pass does the same with an ordinary block in step A.
//   hipcc --offload-arch=gfx950 -O2 tools/uc_stale_repro.hip -o tools/_build/uc_stale_repro && tools/_build/uc_stale_repro [rounds] [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void fill(unsigned long long *p, size_t n, unsigned long long pat) { for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] = pat ^ i; }
__global__ void addk(unsigned long long *p, size_t n, unsigned long long pat) { for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) p[i] += pat ^ i; }
__global__ void check(const unsigned long long *p, size_t n, unsigned long long pat, unsigned long long old, unsigned long long *bad)
{
    for (size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x)
        if (p[i] != (pat ^ i)) { atomicAdd(&bad[0], 1ull); if (p[i] == (old ^ i)) atomicAdd(&bad[1], 1ull); }
}
int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    const size_t bytes = (size_t) (argc > 2 ? atoi(argv[2]) : 6) << 20, n = bytes / 8;
    unsigned long long *bad = nullptr, hb[2], tot[2][3][2] = {};
    std::vector<unsigned long long> h(n);
    CK(hipMalloc(&bad, 16));
    for (int uc = 1; uc >= 0; --uc)                      // uc = 1: the experiment; uc = 0: the control (ordinary memory in step A)
        for (int r = 0; r < rounds; ++r) {
            const unsigned long long P1 = 0x1111000000000000ull + (unsigned long long) r * 2, P2 = P1 + 0x7777000000000001ull;
            const int writer = r % 3;
            void *a = nullptr;
            if (uc) CK(hipExtMallocWithFlags(&a, bytes, hipDeviceMallocUncached)); else CK(hipMalloc(&a, bytes));
            fill<<<512, 256>>>((unsigned long long *) a, n, P1);
            CK(hipDeviceSynchronize());
            CK(hipFree(a));
            unsigned long long *b[3] = {};
            for (int k = 0; k < 3; ++k) CK(hipMalloc((void **) &b[k], bytes));       // one of them usually is the block just freed
            for (int k = 0; k < 3; ++k) {
                if (writer == 0) fill<<<512, 256>>>(b[k], n, P2);
                else if (writer == 1) { for (size_t i = 0; i < n; ++i) h[i] = P2 ^ i; CK(hipMemcpy(b[k], h.data(), bytes, hipMemcpyHostToDevice)); }
                else { CK(hipMemsetAsync(b[k], 0, bytes, 0)); addk<<<512, 256>>>(b[k], n, P2); }
            }
            CK(hipMemset(bad, 0, 16));
            for (int k = 0; k < 3; ++k) check<<<512, 256>>>(b[k], n, P2, P1, bad);
            CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
            for (int k = 0; k < 3; ++k) {                                               // the copy engine's view of the same memory
                CK(hipMemcpy(h.data(), b[k], bytes, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n; ++i) if (h[i] != (P2 ^ i)) { ++hb[0]; if (h[i] == (P1 ^ i)) ++hb[1]; }
                CK(hipFree(b[k]));
            }
            tot[uc][writer][0] += hb[0]; tot[uc][writer][1] += hb[1];
        }
    const char *wn[3] = {"kernel store", "hipMemcpy H2D", "hipMemset + kernel add"};
    for (int uc = 1; uc >= 0; --uc)
        for (int w = 0; w < 3; ++w)
            printf("%-28s step B written by %-22s: %llu wrong 8-byte words in %d rounds x 3 blocks x %zu MiB, %llu of them still the old pattern\n",
                   uc ? "after an UNCACHED block" : "after an ordinary block", wn[w], tot[uc][w][0], (rounds + 2 - w) / 3, bytes >> 20, tot[uc][w][1]);
    return (tot[1][0][0] | tot[1][1][0] | tot[1][2][0]) ? 1 : 0;
}
