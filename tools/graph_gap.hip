/* graph_gap.hip — does a hipGraph shorten a pass-shaped chain of dependent small operations on this stack?
 * (evidence tool for DESIGN.md §9.2; not part of the library)
 * chain = H2D 2 KB -> kernel -> kernel -> kernel -> D2H 1 KB, host synchronises after each chain (like a CRS2_LM pass)
 *   (a) five stream operations + hipStreamSynchronize      (b) the same chain captured once, hipGraphLaunch + synchronise
 * build: hipcc --offload-arch=gfx950 -O3 tools/graph_gap.hip -o tools/graph_gap */
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(int *p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }

int main()
{
    hipStream_t st;
    hipStreamCreate(&st);
    int *d, *h;
    hipMalloc(&d, 4096);
    hipHostMalloc(&h, 4096);
    hipMemset(d, 0, 4096);
    const int iters = 2000;
    auto chain = [&]() {
        hipMemcpyAsync(d + 256, h, 2048, hipMemcpyHostToDevice, st);
        tiny<<<64, 256, 0, st>>>(d, 1);
        tiny<<<700, 512, 0, st>>>(d, 2);
        tiny<<<44, 512, 0, st>>>(d, 3);
        hipMemcpyAsync(h + 512, d, 1024, hipMemcpyDeviceToHost, st);
    };
    for (int i = 0; i < 50; ++i) { chain(); hipStreamSynchronize(st); }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) { chain(); hipStreamSynchronize(st); }
    auto t1 = std::chrono::steady_clock::now();
    printf("stream launches : %.2f us per chain\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);

    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    chain();
    hipStreamEndCapture(st, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("graph instantiate failed\n"); return 1; }
    for (int i = 0; i < 50; ++i) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    t1 = std::chrono::steady_clock::now();
    printf("graph launch    : %.2f us per chain\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    return 0;
}
