/* devrt.hip — the thin C-ABI device-runtime layer under the C host code: memory, streams, events.
 * No HIP type crosses the boundary (streams/events travel as void*), so the host side stays
 * plain C compiled by gcc, as north_star asks ("host code in C calling HIP through a thin C-ABI
 * layer").  There is no CPU fallback anywhere: with no visible device nla_dev_count() returns 0
 * and the optimisers fail with NLOPT_FAILURE and an errmsg. */
#include <hip/hip_runtime.h>
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
 * without cache maintenance — for the small buffers the workgroups of hip/crs_chain.hip hand results to each other through */
extern "C" void *nla_dev_malloc_uncached(size_t bytes)
{
    void *p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes ? bytes : 1, hipDeviceMallocUncached) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
    if (alloc_fill() >= 0 && (hipMemset(p, alloc_fill(), bytes ? bytes : 1) != hipSuccess || hipDeviceSynchronize() != hipSuccess)) (void) hipGetLastError();
    return p;
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
