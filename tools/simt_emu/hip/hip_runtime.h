/* tools/simt_emu/hip/hip_runtime.h — a stand-in for <hip/hip_runtime.h> that lets a .hip translation unit be compiled by g++ and
 * its kernels be RUN ON THE CPU, one std::thread per work-item, for debugging kernel LOGIC in the build container (no GPU there):
 *   __global__/__device__/__shared__ -> plain / static storage (workgroups run one after another, so one copy is the workgroup's)
 *   __syncthreads()                  -> a barrier over the workgroup's threads
 *   __shfl_xor(v, m, 64)             -> exchange through a workgroup array between two barriers (every work-item must call it, as
 *                                       the kernels' reductions do)
 *   hipLaunchKernelGGL               -> runs the grid synchronously
 * Development tooling only (tools/lbfgs_emu_check.py): nothing in the product, the tests' GPU path or bench.py uses it.  It says
 * nothing about performance, wave-level timing or memory-model visibility — only that the arithmetic and the control flow of a
 * kernel do what a reference does. */
#ifndef NLA_SIMT_EMU_HIP_RUNTIME_H
#define NLA_SIMT_EMU_HIP_RUNTIME_H
#define NLA_SIMT_EMU 1
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }

namespace simt {
struct Barrier {                       /* sense-reversing spin barrier that yields: hundreds of threads share a few cores here */
    std::atomic<unsigned> count{0}, gen{0}; unsigned total = 0;
    void reset(unsigned n) { total = n; count.store(0); }
    void wait()
    {
        const unsigned g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == total) { count.store(0, std::memory_order_relaxed); gen.store(g + 1, std::memory_order_release); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
};
inline Barrier &barrier() { static Barrier b; return b; }
inline uint64_t *xch(int which) { static uint64_t buf[2][1024]; return buf[which]; }
struct Idx { unsigned x, y, z; };
}
static thread_local simt::Idx threadIdx, blockIdx;
static thread_local simt::Idx blockDim, gridDim;

static inline void __syncthreads() { simt::barrier().wait(); }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64)
{
    static_assert(sizeof(T) <= 8, "emulated shuffle: at most 8 bytes");
    (void) width;
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    /* two exchange buffers used in turn, ONE barrier per shuffle: buffer b is written again two shuffles later, behind the barrier
     * of the shuffle in between, which every work-item passes only after it has read b */
    static thread_local int which = 0;
    simt::xch(which)[threadIdx.x] = raw;
    simt::barrier().wait();
    raw = simt::xch(which)[threadIdx.x ^ (unsigned) mask];
    which ^= 1;
    T out;
    std::memcpy(&out, &raw, sizeof(T));
    return out;
}
/* __all() over the 64 work-items of a wavefront (every work-item of the workgroup must call it) */
static inline bool simt_wave_all(bool p)
{
    static thread_local int which = 0;
    simt::xch(which)[threadIdx.x] = p ? 1 : 0;
    simt::barrier().wait();
    bool all = true;
    const unsigned w0 = threadIdx.x & ~63u;
    for (unsigned l = w0; l < w0 + 64 && l < blockDim.x; ++l) all = all && simt::xch(which)[l] != 0;
    which ^= 1;
    return all;
}
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
static inline unsigned long long wall_clock64() { return 0; }
static inline unsigned __smid() { return 0; }

template <class K, class... A> static inline void simt_launch(K kernel, dim3 grid, dim3 block, A... args)
{
    const unsigned nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        simt::barrier().reset(nthreads);
        std::vector<std::thread> th;
        th.reserve(nthreads);
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([=]() {
                threadIdx = { t % block.x, (t / block.x) % block.y, t / (block.x * block.y) };
                blockIdx = { bx, by, bz };
                blockDim = { block.x, block.y, block.z };
                gridDim = { grid.x, grid.y, grid.z };
                kernel(args...);
                /* a work-item that leaves early must not strand the others at a barrier: kernels here leave uniformly */
            });
        for (auto &t : th) t.join();
    }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) simt_launch(kernel, grid, block, __VA_ARGS__)
#define HIP_SYMBOL(x) x
#endif
