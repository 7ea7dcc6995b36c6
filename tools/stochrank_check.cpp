/* tools/stochrank_check.cpp — CPU check of hip/isres_stochrank.h (development tooling; tests/test_host_logic.py builds and runs it): the
 * SAME SOURCE that hipcc compiles into isres_stochrank_kernel, compiled by g++ with the wavefront
 * primitives replaced: the 64 lanes of a unit are threads in lockstep (DPP wave shifts and v_readfirstlane are exchanges through an
 * array between barriers), every unit of a small pipeline runs at once (ticket order as on the device), global memory is ordinary
 * memory behind acquire / release atomics.  The kernel must produce the ranking of the reference's double loop (isres.c:206-228)
 * and the same per-sweep "swapped" flags, and every unit's progress counter must end at pop — for populations of one lane, of one
 * unit exactly, of several blocks per unit and of several units, with all units at speed and with units slowed down at random (a delay
 * in front of a unit's polls).  What this does NOT check is the device's memory model or timing — only that the kernel's logic computes
 * the reference's result.
 *
 *   g++ -O1 -std=c++17 -pthread -I nlopt_amd/csrc/hip tools/stochrank_check.cpp -o tools/_build/stochrank_check
 *   tools/_build/stochrank_check [seed]        -> "ok ..." / the first difference; exit code 0 / 1 */
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

namespace emu {
struct Barrier {
    std::atomic<unsigned> count{0}, gen{0};
    void wait()
    {
        const unsigned g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == 64u) { count.store(0, std::memory_order_relaxed); gen.store(g + 1, std::memory_order_release); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
};
struct Unit { Barrier bar; uint32_t xch[2][64]; int shared_int; };
struct Idx { unsigned x; };
static thread_local Unit *unit = nullptr;
static thread_local int which = 0;
static thread_local unsigned jitter = 0;      /* this unit's lane 0 yields so many times before an atomic load (slow unit) */
}
static thread_local emu::Idx threadIdx;

#define SR_KERNEL static
#define SR_SHARED_INT(name) int &name = emu::unit->shared_int
/* a wavefront executes in lockstep: when lane 0 stores a counter behind this wait, every lane's stores in front of it have been issued and
 * have landed; the threads that play the lanes meet at a barrier for that */
#define SR_SETPRIO_HIGH() ((void) 0)
#define SR_CLOCK() ((uint64_t) 0)
#define SR_WAIT_VMCNT0() do { std::atomic_thread_fence(std::memory_order_seq_cst); emu::unit->bar.wait(); std::atomic_thread_fence(std::memory_order_seq_cst); } while (0)
#define __restrict__
#define __device__
#define __forceinline__ inline
#define __ATOMIC_RELAXED_ 0
#define __HIP_MEMORY_SCOPE_AGENT 0
/* the device's relaxed agent-scope accesses to uncached lines: here acquire / release (the kernels add the fences the device needs) */
template <class T> static inline T __hip_atomic_load(const T *p, int, int)
{
    if (emu::jitter && threadIdx.x == 0) for (unsigned i = 0; i < emu::jitter; ++i) std::this_thread::yield();
    return __atomic_load_n(p, __ATOMIC_ACQUIRE);
}
template <class T, class V> static inline void __hip_atomic_store(T *p, V v, int, int) { __atomic_store_n(p, (T) v, __ATOMIC_RELEASE); }
static inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
static inline void __syncthreads() { emu::unit->bar.wait(); }
/* likewise the wavefront-scope acquire behind a poll that only lane 0 makes: no lane loads before lane 0 has seen the counter */
static inline void __builtin_amdgcn_fence(int, const char *) { emu::unit->bar.wait(); std::atomic_thread_fence(std::memory_order_seq_cst); }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_ACQ_REL); }
static inline uint32_t emu_exchange(uint32_t v, int from, uint32_t fallback)
{
    uint32_t *b = emu::unit->xch[emu::which];
    b[threadIdx.x] = v;
    emu::unit->bar.wait();
    const uint32_t r = (from >= 0 && from < 64) ? b[from] : fallback;
    emu::which ^= 1;                    /* two buffers in turn, one barrier per exchange */
    return r;
}
/* v_mov_b32_dpp with wave_shl:1 (0x130: lane i takes lane i + 1) / wave_shr:1 (0x138: lane i takes lane i - 1), bound_ctrl off: a lane
 * without a source keeps `old` */
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool)
{
    const int from = (ctrl == 0x130) ? (int) threadIdx.x + 1 : (int) threadIdx.x - 1;
    return (int) emu_exchange((uint32_t) src, from, (uint32_t) old);
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return (int) emu_exchange((uint32_t) v, 0, 0u); }
static inline bool emu_any(bool p)
{
    uint32_t *b = emu::unit->xch[emu::which];
    b[threadIdx.x] = p ? 1u : 0u;
    emu::unit->bar.wait();
    bool r = false;
    for (int l = 0; l < 64; ++l) r = r || b[l];
    emu::which ^= 1;
    return r;
}
#define SR_ANY(p) emu_any(p)

#include "isres_stochrank.h"

static uint64_t pack(uint32_t idx, uint32_t rf, uint32_t rp, bool pzero)     /* isres_pack (isres_kernels.hip) */
{
    const uint32_t lo = (rf << 12) | (idx & 0xFFFu), hi = (pzero ? 0x80000000u : 0u) | (rp << 8) | (idx >> 12);
    return ((uint64_t) hi << 32) | lo;
}
static uint32_t unpack_idx(uint64_t e) { return ((uint32_t) e & 0xFFFu) | ((((uint32_t) (e >> 32)) & 0xFFu) << 12); }

typedef void (*kernel_t)(int64_t, int64_t, uint64_t *, int *, const uint64_t *, int64_t, int *, uint8_t *, const int *, uint64_t, int64_t);

static int run_case(int pop, unsigned seed, int *n_cases)
{
    std::mt19937_64 rng(seed * 7919u + (unsigned) pop);
    const int nsweeps = pop, units = (nsweeps + 63) / 64, roww = (pop - 1 + 63) / 64 > 0 ? (pop - 1 + 63) / 64 : 1;
    /* values with many ties; dense ranks as isres_rank_count_kernel forms them (rank = number of distinct smaller values) */
    std::vector<int> f(pop), pen(pop);
    for (int i = 0; i < pop; ++i) { f[i] = (int) (rng() % (uint64_t) (pop / 2 + 2)); pen[i] = (rng() % 10 < 4) ? 0 : 1 + (int) (rng() % 5); }
    auto dense = [&](const std::vector<int> &v, int i) { int r = 0; std::vector<char> seen(pop + 8, 0); for (int k = 0; k < pop; ++k) if (v[k] < v[i] && !seen[v[k]]) { seen[v[k]] = 1; ++r; } return (uint32_t) r; };
    std::vector<uint64_t> bits((size_t) pop * roww, 0);
    for (int i = 0; i < pop; ++i) for (int j = 0; j < pop - 1; ++j) if (rng() % 100 < 45) bits[(size_t) i * roww + j / 64] |= 1ull << (j % 64);
    /* the reference's double loop (isres.c:206-228) */
    std::vector<int> ref(pop);
    std::vector<uint8_t> ref_sw(pop, 0);
    for (int i = 0; i < pop; ++i) ref[i] = i;
    for (int i = 0; i < nsweeps; ++i)
        for (int j = 0; j < pop - 1; ++j) {
            const int a = ref[j], b = ref[j + 1];
            const bool byf = ((bits[(size_t) i * roww + j / 64] >> (j % 64)) & 1) || (pen[a] == 0 && pen[b] == 0);
            if (byf ? f[a] > f[b] : pen[a] > pen[b]) { ref[j] = b; ref[j + 1] = a; ref_sw[i] = 1; }
        }
    for (int variant = 0; variant < 2; ++variant) {           /* all units at speed; some units slowed down */
        std::vector<uint64_t> streams((size_t) (units + 1) * pop, ~0ull);       /* (the launcher's fill: SR_UNWRITTEN) */
        for (int i = 0; i < pop; ++i) streams[i] = pack((uint32_t) i, dense(f, i), dense(pen, i), pen[i] == 0);
        std::vector<int> progress(units + 1, 0);
        progress[0] = pop;
        int ticket = 0;
        std::vector<uint8_t> swapped(pop, 7);
        std::vector<emu::Unit> U(units);
        const kernel_t K = isres_stochrank_kernel;
        std::vector<std::thread> th;
        for (int u = 0; u < units; ++u)
            for (int l = 0; l < 64; ++l)
                th.emplace_back([&, u, l]() {
                    emu::unit = &U[u]; emu::which = 0; threadIdx.x = (unsigned) l;
                    emu::jitter = (variant == 1) ? (unsigned) ((u * 2654435761u >> 28) % 4) * 3u : 0u;
                    K(pop, nsweeps, streams.data(), progress.data(), bits.data(), roww, &ticket, swapped.data(), nullptr, 0, 0);
                });
        for (auto &t : th) t.join();
        ++*n_cases;
        for (int u = 0; u <= units; ++u)
            if (0 && progress[u] != pop) {      /* (the counter protocol's; the hand-over through the elements keeps no counters) */ printf("pop %d variant %d: progress[%d] = %d\n", pop, variant, u, progress[u]); return 1; }
        const uint64_t *last = streams.data() + (size_t) units * pop;
        for (int k = 0; k < pop; ++k)
            if ((int) unpack_idx(last[k]) != ref[k]) { printf("pop %d variant %d: position %d holds %u, reference %d\n", pop, variant, k, unpack_idx(last[k]), ref[k]); return 1; }
        for (int i = 0; i < pop; ++i)
            if (swapped[i] != ref_sw[i]) { printf("pop %d variant %d: swapped[%d] = %d, reference %d\n", pop, variant, i, swapped[i], ref_sw[i]); return 1; }
    }
    return 0;
}

int main(int argc, char **argv)
{
    const unsigned seed = argc > 1 ? (unsigned) atoi(argv[1]) : 1u;
    int n = 0;
    for (int pop : {2, 5, 64, 65, 130, 200, 300, 520})
        if (run_case(pop, seed, &n)) return 1;
    printf("ok %d runs (all units at speed, some units slowed down; populations 2 .. 520)\n", n);
    return 0;
}
