/* devrt.hip — the thin C-ABI device-runtime layer under the C host code: memory, streams, events.
 * No HIP type crosses the boundary (streams/events travel as void*), so the host side stays
 * plain C compiled by gcc, as north_star asks ("host code in C calling HIP through a thin C-ABI
 * layer").  There is no CPU fallback anywhere: with no visible device nla_dev_count() returns 0
 * and the optimisers fail with NLOPT_FAILURE and an errmsg. */
#include <hip/hip_runtime.h>
#include "../nla_switches.h"
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
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
    if (fill == -2) { const char *e = NLA_DBG_ENV("NLA_DEV_MALLOC_FILL"); fill = e ? (atoi(e) & 255) : -1; }      /* (nla_switches.h) */
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
 * A POOL, because of round 2's intermittent divergence: runs that allocated these buffers and gave them back to the
 * driver at their end made later ORDINARY allocations show — rarely, a few cache lines at a time — stale contents: memory that has
 * been mapped uncached must not come back as cached memory while caches may still hold its lines (tools/uc_stale_repro.hip is the
 * stand-alone form of that experiment).  Rules:
 *   - a released block goes to the free list of ITS DEVICE and is handed out again to a request it fits without wasting more than
 *     half of it (size classes: a 512 MB trial-point block is not given to a 1 KB request);
 *   - blocks are whole 2 MB pages of their own with a guard page on either side: no ordinary allocation shares a page-table
 *     fragment with an uncached one;
 *   - nothing goes back to the driver while ANY uncached block of that device is in use.  When the last one has been released and
 *     more than NLA_UC_IDLE_CAP bytes sit idle, the largest idle blocks are freed down to the cap — after a hipDeviceSynchronize(),
 *     i.e. with no kernel of this process in flight and the caches written back at the last kernel's end; nlopt_amd_release_device_memory()
 *     does the same down to zero (a long-lived process that is done with its n >= 2048 runs). */
#include <mutex>
#include <vector>
#define NLA_UC_IDLE_CAP ((size_t) 1 << 30)
struct uc_block { void *p, *raw; size_t bytes; int dev; bool busy; };
static std::mutex uc_mu;
static std::vector<uc_block> uc_pool;
static long uc_raw_allocs = 0, uc_driver_frees = 0;       /* what tests/test_gpu_crs.py watches: see nla_debug_uncached_stats */
static int uc_switch(const char *name, int dflt)          /* development switches, read once (nla_switches.h: none in the shipped library) */
{
    const char *e = NLA_DBG_ENV(name);
    return e ? atoi(e) : dflt;
}
static bool uc_pool_on() { static const int on = uc_switch("NLA_UC_POOL", 1); return on != 0; }            /* 0: round 2's alloc / free per run (A/B) */
static bool uc_disabled() { static const int off = uc_switch("NLA_NO_UNCACHED", 0); return off != 0; }     /* ordinary memory instead (A/B) */
/* frees idle blocks of `dev`, largest first, until at most `keep` idle bytes remain; caller holds uc_mu and has made sure no block of
 * the device is busy */
static void uc_trim_locked(int dev, size_t keep)
{
    size_t idle = 0;
    for (const uc_block &b : uc_pool) if (b.dev == dev && !b.busy) idle += b.bytes;
    if (idle <= keep) return;
    (void) hipDeviceSynchronize();
    while (idle > keep) {
        int big = -1;
        for (int i = 0; i < (int) uc_pool.size(); ++i)
            if (uc_pool[i].dev == dev && !uc_pool[i].busy && (big < 0 || uc_pool[i].bytes > uc_pool[big].bytes)) big = i;
        if (big < 0) break;
        idle -= uc_pool[big].bytes;
        (void) hipFree(uc_pool[big].raw);
        ++uc_driver_frees;
        uc_pool.erase(uc_pool.begin() + big);
    }
}
extern "C" void *nla_dev_malloc_uncached(size_t bytes)
{
    void *p = nullptr;
    int dev = 0;
    if (!bytes) bytes = 1;
    if (uc_disabled()) return nla_dev_malloc(bytes);
    if (hipGetDevice(&dev) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    if (!uc_pool_on()) {
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
        std::lock_guard<std::mutex> g(uc_mu);
        ++uc_raw_allocs;
        return p;
    }
    const size_t two = (size_t) 2 << 20, want = (bytes + two - 1) / two * two;
    {
        std::lock_guard<std::mutex> g(uc_mu);
        int best = -1;
        for (int i = 0; i < (int) uc_pool.size(); ++i) {
            const uc_block &b = uc_pool[i];
            if (b.dev == dev && !b.busy && b.bytes >= want && b.bytes <= 2 * want && (best < 0 || b.bytes < uc_pool[best].bytes)) best = i;
        }
        if (best >= 0) { uc_pool[best].busy = true; p = uc_pool[best].p; bytes = uc_pool[best].bytes; }
    }
    if (!p) {
        void *raw = nullptr;
        if (hipExtMallocWithFlags(&raw, want + 2 * two, hipDeviceMallocUncached) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
        p = (void *) (((uintptr_t) raw + two + two - 1) / two * two);
        bytes = want;
        std::lock_guard<std::mutex> g(uc_mu);
        ++uc_raw_allocs;
        uc_pool.push_back(uc_block{p, raw, want, dev, true});
    }
    if (alloc_fill() >= 0 && (hipMemset(p, alloc_fill(), bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) (void) hipGetLastError();
    return p;
}
extern "C" void nla_dev_free_uncached(void *p)
{
    if (!p) return;
    if (uc_disabled()) { (void) hipFree(p); return; }
    std::lock_guard<std::mutex> g(uc_mu);
    if (!uc_pool_on()) { ++uc_driver_frees; (void) hipFree(p); return; }
    for (uc_block &b : uc_pool)
        if (b.p == p) {
            b.busy = false;
            bool any_busy = false;
            for (const uc_block &o : uc_pool) any_busy = any_busy || (o.dev == b.dev && o.busy);
            if (!any_busy) {
                int cur = 0;
                const int dev = b.dev;
                if (hipGetDevice(&cur) == hipSuccess && (cur == dev || hipSetDevice(dev) == hipSuccess)) {
                    uc_trim_locked(dev, NLA_UC_IDLE_CAP);
                    if (cur != dev) (void) hipSetDevice(cur);
                } else (void) hipGetLastError();
            }
            return;
        }
}
/* gives every idle uncached block (of every device) back to the driver — for a long-lived process between its large runs; blocks in
 * use stay.  Returns the bytes released.  (include/nlopt_amd.h) */
extern "C" size_t nlopt_amd_release_device_memory(void)
{
    std::lock_guard<std::mutex> g(uc_mu);
    size_t before = 0, after = 0;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void) hipGetLastError(); return 0; }
    for (const uc_block &b : uc_pool) if (!b.busy) before += b.bytes;
    std::vector<int> devs;
    for (const uc_block &b : uc_pool) {
        bool seen = false, busy = false;
        for (int d : devs) seen = seen || d == b.dev;
        for (const uc_block &o : uc_pool) busy = busy || (o.dev == b.dev && o.busy);
        if (!seen && !busy) devs.push_back(b.dev);
    }
    for (int d : devs) if (d == cur || hipSetDevice(d) == hipSuccess) uc_trim_locked(d, 0);
    (void) hipSetDevice(cur);
    for (const uc_block &b : uc_pool) if (!b.busy) after += b.bytes;
    return before - after;
}
/* [0] uncached allocations obtained from the driver so far, [1] of them returned to it, [2] blocks in the pool, [3] of them in use */
extern "C" void nla_debug_uncached_stats(long out[4])
{
    std::lock_guard<std::mutex> g(uc_mu);
    int busy = 0;
    for (const uc_block &b : uc_pool) busy += b.busy ? 1 : 0;
    out[0] = uc_raw_allocs; out[1] = uc_driver_frees; out[2] = (long) uc_pool.size(); out[3] = busy;
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
/* host memory the caller owns (the shm transport's segment) made page-locked and device-visible */
extern "C" int nla_host_register(void *p, size_t bytes)
{
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) { (void) hipGetLastError(); return -1; }
    return 0;
}
extern "C" void nla_host_unregister(void *p) { if (p && hipHostUnregister(p) != hipSuccess) (void) hipGetLastError(); }

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
/* a stream confined to a share of the compute units: CU i belongs to part (i mod parts).  For ranks that SHARE one device and whose
 * kernels wait for each other inside a launch (the column-sharded CRS2_LM windows on a one-GPU test box): with disjoint shares both
 * kernels are resident whatever their sizes.  parts < 2: an ordinary stream. */
extern "C" void *nla_stream_create_cu_share(int part, int parts)
{
    if (parts < 2) return nla_stream_create();
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    uint32_t mask[64];
    if (words > 64 || part < 0 || part >= parts) return nullptr;
    for (int w = 0; w < words; ++w) mask[w] = 0;
    for (int i = 0; i < ncu; ++i) if (i % parts == part) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t) words, mask) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    return (void *) s;
}

/* ---- device memory another PROCESS can map (hipIpc*): the column-sharded windows' TX / flag buffers ---------------------------------
 * export: an opaque NLA_IPC_BYTES blob for a pointer obtained from nla_dev_malloc_uncached / nla_dev_malloc (blocks of the uncached pool
 * lie inside a larger driver allocation: the blob carries the offset); open: the peer's pointer as mapped in this process; close. */
struct nla_ipc_blob { hipIpcMemHandle_t h; uint64_t offset; uint64_t magic; };
static_assert(sizeof(nla_ipc_blob) <= 96, "NLA_IPC_BYTES");
struct ipc_open_rec { void *user, *base; };
static std::mutex ipc_mu;
static std::vector<ipc_open_rec> ipc_opened;
extern "C" int nla_ipc_export(const void *p, void *blob96)
{
    nla_ipc_blob b;
    memset(&b, 0, sizeof b);
    const void *base = p;
    {
        std::lock_guard<std::mutex> g(uc_mu);
        for (const uc_block &u : uc_pool) if (u.p == p) { base = u.raw; break; }
    }
    hipError_t e = hipIpcGetMemHandle(&b.h, const_cast<void *>(base));
    if (e != hipSuccess) { (void) hipGetLastError(); return (int) e; }
    b.offset = (uint64_t) ((const char *) p - (const char *) base);
    b.magic = 0x6e6c61697063ull;
    memset(blob96, 0, 96);
    memcpy(blob96, &b, sizeof b);
    return 0;
}
extern "C" void *nla_ipc_open(const void *blob96)
{
    nla_ipc_blob b;
    memcpy(&b, blob96, sizeof b);
    if (b.magic != 0x6e6c61697063ull) return nullptr;
    void *base = nullptr;
    if (hipIpcOpenMemHandle(&base, b.h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    void *user = (char *) base + b.offset;
    std::lock_guard<std::mutex> g(ipc_mu);
    ipc_opened.push_back(ipc_open_rec{user, base});
    return user;
}
extern "C" void nla_ipc_close(void *user)
{
    if (!user) return;
    void *base = nullptr;
    {
        std::lock_guard<std::mutex> g(ipc_mu);
        for (size_t i = 0; i < ipc_opened.size(); ++i)
            if (ipc_opened[i].user == user) { base = ipc_opened[i].base; ipc_opened.erase(ipc_opened.begin() + (long) i); break; }
    }
    if (base && hipIpcCloseMemHandle(base) != hipSuccess) (void) hipGetLastError();
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

/* a gate in a stream (nlopt_amd.h): one wavefront, one lane polling with an agent-scope load between sleeps; wall_clock64 ticks at 100 MHz */
__global__ void nla_gate_kernel(const int32_t *counter, int32_t from, int32_t need, unsigned long long timeout_ticks, int32_t *gave_up)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while ((int32_t) ((uint32_t) __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (uint32_t) from) < need) {
        if (wall_clock64() - t0 > timeout_ticks) {
            /* what it waits for is not running beside it (streams that cannot overlap: a serialising profiler, one hardware queue): say so,
             * the caller stops gating (round-5 advisor) */
            if (gave_up) __hip_atomic_store(gave_up, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}
extern "C" int nla_k_gate(const int32_t *counter, int32_t from, int32_t need, double timeout_ms, int32_t *gave_up, void *stream)
{
    if (!counter || need <= 0) return 0;
    hipLaunchKernelGGL(nla_gate_kernel, dim3(1), dim3(64), 0, (hipStream_t) stream, counter, from, need, (unsigned long long) (timeout_ms * 1e5), gave_up);
    return (int) hipGetLastError();
}
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
