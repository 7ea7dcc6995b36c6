/* tools/ipc_probe.hip — can two PROCESSES on one MI355X hand data to each other INSIDE their running kernels?
 *
 * What the column-sharded chain kernel (hip/crs_chain.hip, sharded instance) needs from the platform, probed before it is built on:
 *   1. hipIpcGetMemHandle / hipIpcOpenMemHandle on the memory kinds the engine could use for TX / flags
 *      (hipDeviceMallocUncached, hipDeviceMallocFinegrained, plain hipMalloc);
 *   2. kernels of two processes resident AT THE SAME TIME (each on a CU-masked stream: hipExtStreamCreateWithCUMask, half the
 *      compute units each) so that a kernel may wait for a word the other process's kernel writes;
 *   3. what such a hand-over costs: a ping-pong of one word between the two kernels (system-scope stores into the peer's buffer,
 *      system-scope loads of the own one), and a 1 KiB payload + flag (the chain kernel's chunk: 128 doubles);
 *   4. the same when each kernel is wide enough to FILL its half of the chip (co-residency under load).
 * Every wait in a kernel gives up after 2 s of the 100 MHz clock, and the parent kills both children after 60 s: no hang.
 *
 *   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.hip -o tools/ipc_probe && tools/ipc_probe [kind]     kind: uncached|fine|plain|all
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[rank %d] %s -> %s\n", g_rank, #x, hipGetErrorString(e_)); fflush(stdout); return 1; } } while (0)
static int g_rank;

__device__ __forceinline__ uint32_t ld_sys(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

/* buffer layout (u32 words): [0] flag A, [64] flag B, [1024 ...] payload area (doubles) */
/* ping-pong: rank 0 writes i into the peer's flag, waits for i in its own; rank 1 the other way round.  out[0] = rounds done, out[1] = ticks */
__global__ void pingpong_kernel(uint32_t *mine, uint32_t *peer, int rank, int iters, int payload, uint64_t *out)
{
    const uint64_t t0 = wall_clock64();
    const int lane = threadIdx.x;
    double *pay_peer = reinterpret_cast<double *>(peer + 1024), *pay_mine = reinterpret_cast<double *>(mine + 1024);
    int i, bad = 0;
    for (i = 1; i <= iters; ++i) {
        if (rank == 0) {
            if (payload) { for (int k = lane; k < payload; k += 64) st_sys64(reinterpret_cast<uint64_t *>(pay_peer + k), (uint64_t) __double_as_longlong((double) (i * 1000 + k))); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (lane == 0) st_sys(peer, (uint32_t) i);
        }
        /* wait for round i in the own buffer */
        uint64_t tw = wall_clock64();
        while ((int32_t) (ld_sys(mine) - (uint32_t) i) < 0) {
            if (wall_clock64() - tw > 200000000ull) { if (lane == 0) { out[0] = (uint64_t) (i - 1); out[1] = wall_clock64() - t0; out[2] = 1; } return; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (payload) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            for (int k = lane; k < payload; k += 64) { const double v = __builtin_nontemporal_load(pay_mine + k); if (v != (double) (i * 1000 + k)) ++bad; }
        }
        if (rank == 1) {
            if (payload) { for (int k = lane; k < payload; k += 64) st_sys64(reinterpret_cast<uint64_t *>(pay_peer + k), (uint64_t) __double_as_longlong((double) (i * 1000 + k))); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            if (lane == 0) st_sys(peer, (uint32_t) i);
        }
    }
    bad = __reduce_add_sync(~0ull, bad);
    if (lane == 0) { out[0] = (uint64_t) iters; out[1] = wall_clock64() - t0; out[2] = 0; out[3] = (uint64_t) bad; }
}

/* co-residency under load: `blocks` workgroups per process; block b writes the peer's flag[b] and waits for its own flag[b].  Only
 * completes if both grids make progress at the same time (each wider than its half of the chip: later blocks start when earlier leave) */
__global__ void wide_kernel(uint32_t *mine, uint32_t *peer, uint32_t seq, uint64_t *out)
{
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        st_sys(peer + 2048 + b, seq);
        uint64_t tw = wall_clock64();
        while ((int32_t) (ld_sys(mine + 2048 + b) - seq) < 0) {
            if (wall_clock64() - tw > 200000000ull) { atomicAdd((unsigned long long *) &out[4], 1ull); break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}

static int alloc_kind(const char *kind, void **p, size_t bytes)
{
    if (!strcmp(kind, "uncached")) return (int) hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    if (!strcmp(kind, "fine")) return (int) hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    return (int) hipMalloc(p, bytes);
}

static int child(int rank, const char *kind, int rd, int wr)
{
    g_rank = rank;
    CK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    const size_t bytes = 1 << 20;
    void *mine = NULL, *peer = NULL;
    uint64_t *out = NULL;
    int rc = alloc_kind(kind, &mine, bytes);
    if (rc) { printf("[rank %d] %s allocation failed: %s\n", rank, kind, hipGetErrorString((hipError_t) rc)); return 1; }
    CK(hipMemset(mine, 0, bytes));
    CK(hipHostMalloc((void **) &out, 64));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t h, hp;
    rc = (int) hipIpcGetMemHandle(&h, mine);
    if (rc) { printf("[rank %d] hipIpcGetMemHandle(%s) failed: %s\n", rank, kind, hipGetErrorString((hipError_t) rc)); memset(&h, 0, sizeof h); }
    int ok = rc == 0, okp = 0;
    if (write(wr, &ok, sizeof ok) != sizeof ok || write(wr, &h, sizeof h) != sizeof h) return 1;
    if (read(rd, &okp, sizeof okp) != sizeof okp || read(rd, &hp, sizeof hp) != sizeof hp) return 1;
    if (!ok || !okp) return 1;
    rc = (int) hipIpcOpenMemHandle(&peer, hp, hipIpcMemLazyEnablePeerAccess);
    if (rc) { printf("[rank %d] hipIpcOpenMemHandle(%s) failed: %s\n", rank, kind, hipGetErrorString((hipError_t) rc)); ok = 0; }
    if (write(wr, &ok, sizeof ok) != sizeof ok || read(rd, &okp, sizeof okp) != sizeof okp || !ok || !okp) return 1;
    /* half of the compute units each: even / odd bits (whatever the bit -> (XCD, CU) mapping is, the halves are disjoint) */
    uint32_t mask[16];
    const int words = (ncu + 31) / 32;
    for (int w = 0; w < words; ++w) mask[w] = rank == 0 ? 0x55555555u : 0xaaaaaaaau;
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t) words, mask));
    for (int payload = 0; payload <= 128; payload += 128) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 2000;
            memset(out, 0, 64);
            CK(hipMemsetAsync(mine, 0, 4096 * 4, st));
            CK(hipStreamSynchronize(st));
            /* both sides cleared before either starts */
            int go = 1, gop = 0;
            if (write(wr, &go, sizeof go) != sizeof go || read(rd, &gop, sizeof gop) != sizeof gop) return 1;
            hipLaunchKernelGGL(pingpong_kernel, dim3(1), dim3(64), 0, st, (uint32_t *) mine, (uint32_t *) peer, rank, iters, payload, out);
            CK(hipStreamSynchronize(st));
            printf("[rank %d] %s ping-pong payload %4d B: %llu / %d rounds, %.2f us per round trip%s, bad payload words %llu\n", rank, kind, payload * 8,
                   (unsigned long long) out[0], iters, out[0] ? (double) out[1] / 100.0 / (double) out[0] : 0.0, out[2] ? "  TIMED OUT" : "", (unsigned long long) out[3]);
            fflush(stdout);
        }
    }
    for (uint32_t seq = 1; seq <= 3; ++seq) {
        const int blocks = 1024;                  /* 128 CUs x 8 resident single-wave workgroups: wider than the half chip only by queueing */
        memset(out, 0, 64);
        int go = 1, gop = 0;
        if (write(wr, &go, sizeof go) != sizeof go || read(rd, &gop, sizeof gop) != sizeof gop) return 1;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(wide_kernel, dim3(blocks), dim3(64), 0, st, (uint32_t *) mine, (uint32_t *) peer, seq, out);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("[rank %d] %s wide kernel %d workgroups, round %u: %.3f ms, %llu waits timed out\n", rank, kind, blocks, seq, ms, (unsigned long long) out[4]);
        fflush(stdout);
    }
    CK(hipIpcCloseMemHandle(peer));
    CK(hipStreamDestroy(st));
    CK(hipFree(mine));
    return 0;
}

static int run_kind(const char *kind)
{
    int p01[2], p10[2];
    if (pipe(p01) || pipe(p10)) return 1;
    pid_t a = fork();
    if (a == 0) { alarm(60); _exit(child(0, kind, p10[0], p01[1])); }
    pid_t b = fork();
    if (b == 0) { alarm(60); _exit(child(1, kind, p01[0], p10[1])); }
    int sa = 0, sb = 0;
    waitpid(a, &sa, 0); waitpid(b, &sb, 0);
    printf("== %s: rank 0 exit %d%s, rank 1 exit %d%s\n", kind, WEXITSTATUS(sa), WIFSIGNALED(sa) ? " (signal)" : "", WEXITSTATUS(sb), WIFSIGNALED(sb) ? " (signal)" : "");
    fflush(stdout);
    close(p01[0]); close(p01[1]); close(p10[0]); close(p10[1]);
    return (WIFEXITED(sa) && WEXITSTATUS(sa) == 0 && WIFEXITED(sb) && WEXITSTATUS(sb) == 0) ? 0 : 1;
}

int main(int argc, char **argv)
{
    const char *kind = argc > 1 ? argv[1] : "all";
    setvbuf(stdout, NULL, _IOLBF, 0);
    if (strcmp(kind, "all")) return run_kind(kind);
    int rc = 0;
    rc |= run_kind("uncached");
    rc |= run_kind("fine");
    rc |= run_kind("plain");
    return rc;
}
