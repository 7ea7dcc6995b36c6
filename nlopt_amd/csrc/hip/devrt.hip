/* devrt.hip — the thin C-ABI device-runtime layer under the C host code: memory, streams, events.
 * No HIP type crosses the boundary (streams/events travel as void*), so the host side stays
 * plain C compiled by gcc, as north_star asks ("host code in C calling HIP through a thin C-ABI
 * layer").  There is no CPU fallback anywhere: with no visible device nla_dev_count() returns 0
 * and the optimisers fail with NLOPT_FAILURE and an errmsg. */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "../../../include/nlopt_amd.h"

extern "C" int nla_dev_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return c;
}
extern "C" int nla_dev_set(int dev) { return (int) hipSetDevice(dev); }

/* test aid: NLA_DEV_MALLOC_FILL=<0..255> fills every new device allocation with that byte (fresh allocations are whatever the
 * previous owner left — a kernel that reads memory it never wrote shows up as a result that depends on the fill) */
static int alloc_fill(void)
{
    static int fill = -2;
    if (fill == -2) { const char *e = getenv("NLA_DEV_MALLOC_FILL"); fill = e ? (atoi(e) & 255) : -1; }
    return fill;
}
extern "C" void *nla_dev_malloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    if (alloc_fill() >= 0 && (hipMemset(p, alloc_fill(), bytes ? bytes : 1) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) (void) hipGetLastError();
    return p;
}
extern "C" void nla_dev_free(void *p) { if (p) (void) hipFree(p); }
/* device memory no cache holds on to (MTYPE UC): what one workgroup stores any other workgroup loads, on whichever XCD it runs,
 * without cache maintenance — for the small buffers the workgroups of hip/crs_chain.hip hand results to each other through.
 *
 * These blocks are NEVER given back to the driver while the process lives: a released block goes to a free list and the next
 * request of at most its size gets it again.  Round 2 allocated and freed them per run, and runs then saw — rarely, a few cache
 * lines at a time — stale contents in ORDINARY allocations made afterwards ("the intermittent divergence"): memory that
 * has been mapped uncached must not come back as cached memory (or the other way round) while caches may still hold its lines. */
#include <mutex>
struct uc_block { void *p; size_t bytes; bool busy; };
static std::mutex uc_mu;
static uc_block uc_pool[64];
static int uc_n = 0;
static long uc_raw_allocs = 0, uc_driver_frees = 0;       /* what tests/test_gpu_crs.py watches: see nla_debug_uncached_stats */
static bool uc_pool_on()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("NLA_UC_POOL"); on = (e && atoi(e) == 0) ? 0 : 1; }    /* NLA_UC_POOL=0: the round-2 behaviour (A/B) */
    return on != 0;
}
extern "C" void *nla_dev_malloc_uncached(size_t bytes)
{
    void *p = nullptr;
    if (!bytes) bytes = 1;
    if (getenv("NLA_NO_UNCACHED")) return nla_dev_malloc(bytes);         /* A/B switch (debugging): ordinary memory instead */
    if (uc_pool_on()) {
        std::lock_guard<std::mutex> g(uc_mu);
        int best = -1;
        for (int i = 0; i < uc_n; ++i)
            if (!uc_pool[i].busy && uc_pool[i].bytes >= bytes && (best < 0 || uc_pool[i].bytes < uc_pool[best].bytes)) best = i;
        if (best >= 0) { uc_pool[best].busy = true; return uc_pool[best].p; }
    }
    if (uc_pool_on()) {
        /* whole 2 MB pages of its own, with a guard page on either side: no ordinary allocation shares a page-table fragment with
         * an uncached one */
        const size_t two = (size_t) 2 << 20, want = (bytes + two - 1) / two * two;
        void *raw = nullptr;
        if (hipExtMallocWithFlags(&raw, want + 2 * two, hipDeviceMallocUncached) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
        p = (void *) (((uintptr_t) raw + two + two - 1) / two * two);
        bytes = want;
    } else if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    { std::lock_guard<std::mutex> g(uc_mu); ++uc_raw_allocs; }
    if (alloc_fill() >= 0 && (hipMemset(p, alloc_fill(), bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) (void) hipGetLastError();
    if (uc_pool_on()) {
        std::lock_guard<std::mutex> g(uc_mu);
        if (uc_n < 64) { uc_pool[uc_n].p = p; uc_pool[uc_n].bytes = bytes; uc_pool[uc_n].busy = true; ++uc_n; }
        /* (a full table: the block is simply never pooled — and, by the rule above, never freed either: see nla_dev_free_uncached) */
    }
    return p;
}
extern "C" void nla_dev_free_uncached(void *p)
{
    if (!p) return;
    if (getenv("NLA_NO_UNCACHED")) { (void) hipFree(p); return; }
    if (uc_pool_on()) {
        std::lock_guard<std::mutex> g(uc_mu);
        for (int i = 0; i < uc_n; ++i) if (uc_pool[i].p == p) { uc_pool[i].busy = false; return; }
        return;                                   /* not in the table (it was full): kept until the process ends */
    }
    { std::lock_guard<std::mutex> g(uc_mu); ++uc_driver_frees; }
    (void) hipFree(p);
}
/* [0] uncached allocations obtained from the driver so far, [1] of them returned to it, [2] blocks in the pool, [3] of them in use */
extern "C" void nla_debug_uncached_stats(long out[4])
{
    std::lock_guard<std::mutex> g(uc_mu);
    int busy = 0;
    for (int i = 0; i < uc_n; ++i) busy += uc_pool[i].busy ? 1 : 0;
    out[0] = uc_raw_allocs; out[1] = uc_driver_frees; out[2] = uc_n; out[3] = busy;
}
/* development aid (tools/stress_crs.py --uc-churn): one raw uncached allocation, written once, given straight back to the driver —
 * what every round-2 run did with its trial-point buffers */
extern "C" int nla_debug_uncached_churn(size_t bytes)
{
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes ? bytes : 1, hipDeviceMallocUncached) != hipSuccess) { (void) hipGetLastError(); return -1; }
    hipError_t e = hipMemset(p, 0x5a, bytes ? bytes : 1);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void) hipFree(p);
    return (int) e;
}

extern "C" void *nla_host_malloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return p;
}
extern "C" void nla_host_free(void *p) { if (p) (void) hipHostFree(p); }

extern "C" int nla_memcpy_h2d(void *dst, const void *h_src, size_t bytes, void *stream)
{
    if (!bytes) return 0;
    return (int) hipMemcpyAsync(dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t) stream);
}
extern "C" int nla_memcpy_d2h(void *h_dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return 0;
    return (int) hipMemcpyAsync(h_dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t) stream);
}
extern "C" int nla_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return 0;
    return (int) hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream);
}
extern "C" int nla_memset(void *dst, int value, size_t bytes, void *stream)
{
    if (!bytes) return 0;
    return (int) hipMemsetAsync(dst, value, bytes, (hipStream_t) stream);
}

extern "C" void *nla_stream_create(void)
{
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return (void *) s;
}
/* a stream whose kernels are dispatched after those of ordinary streams when both have work ready: for background work that
 * should fill the gaps another stream's latency-bound kernels leave, without taking compute units from its throughput-bound ones */
extern "C" void *nla_stream_create_background(void)
{
    hipStream_t s = nullptr;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void) hipGetLastError(); least = 0; }
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) { (void) hipGetLastError(); return nla_stream_create(); }
    return (void *) s;
}
extern "C" void nla_stream_destroy(void *stream) { if (stream) (void) hipStreamDestroy((hipStream_t) stream); }
extern "C" int nla_stream_sync(void *stream) { return (int) hipStreamSynchronize((hipStream_t) stream); }

extern "C" int nla_stream_query(void *stream)
{
    const hipError_t e = hipStreamQuery((hipStream_t) stream);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) { (void) hipGetLastError(); return -1; }
    return (int) e;
}

extern "C" void *nla_event_create(void)
{
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return (void *) e;
}
extern "C" void nla_event_destroy(void *ev) { if (ev) (void) hipEventDestroy((hipEvent_t) ev); }
extern "C" int nla_event_record(void *ev, void *stream) { return (int) hipEventRecord((hipEvent_t) ev, (hipStream_t) stream); }
extern "C" int nla_event_sync(void *ev) { return (int) hipEventSynchronize((hipEvent_t) ev); }
extern "C" float nla_event_elapsed_ms(void *ev0, void *ev1)
{
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, (hipEvent_t) ev0, (hipEvent_t) ev1) != hipSuccess) { (void) hipGetLastError(); return -1.f; }
    return ms;
}
extern "C" int nla_stream_wait_event(void *stream, void *ev) { return (int) hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) ev, 0); }
extern "C" const char *nla_dev_error_string(int err) { return hipGetErrorString((hipError_t) err); }

/* ---- code objects supplied at run time (user device objectives, userobj.c) ------------------------------------- */
extern "C" void *nla_module_load_file(const char *path)
{
    hipModule_t m = nullptr;
    if (hipModuleLoad(&m, path) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return (void *) m;
}
extern "C" void *nla_module_load_data(const void *image)
{
    hipModule_t m = nullptr;
    if (hipModuleLoadData(&m, image) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return (void *) m;
}
extern "C" void nla_module_unload(void *module) { if (module) (void) hipModuleUnload((hipModule_t) module); }
extern "C" void *nla_module_function(void *module, const char *name)
{
    hipFunction_t f = nullptr;
    if (hipModuleGetFunction(&f, (hipModule_t) module, name) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return (void *) f;
}
/* params[i] = address of the kernel's i-th argument (hipModuleLaunchKernel's kernelParams form) */
extern "C" int nla_module_launch(void *function, unsigned grid_x, unsigned block_x, void **params, void *stream)
{
    return (int) hipModuleLaunchKernel((hipFunction_t) function, grid_x, 1, 1, block_x, 1, 1, 0, (hipStream_t) stream, params, nullptr);
}
