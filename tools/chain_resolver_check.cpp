/* tools/chain_resolver_check.cpp — CPU check of hip/crs_chain_resolver.h (development tooling; tests/test_host_logic.py builds and
 * runs it): the SAME SOURCE that hipcc compiles into crs_chain_kernel is compiled here by g++ with the wavefront
 * primitives replaced — 64 threads play the lanes in lockstep (ballot / readlane exchange through an array between barriers, as
 * tools/simt_emu does for shuffles), one more thread plays the evaluating workgroups: it publishes the slots' records out of order,
 * some of them only after the chain has got past an earlier slot (a slot waiting for a hazard row).  The published rowstate words,
 * next / pk and the final counters must equal the sequential statement of crs_trial's decisions (crs.c:125-156) below, whatever the
 * interleaving.  Says nothing about the device's memory model — only that the register walk takes the sequential statement's decisions.
 *
 *   g++ -O1 -std=c++17 -pthread -I nlopt_amd/csrc/hip tools/chain_resolver_check.cpp -o tools/_build/chain_resolver_check
 *   tools/_build/chain_resolver_check [cases] [seed]      -> prints "ok <cases>" or the first difference, exit code 0 / 1 */
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

/* ---- the wavefront, emulated ---- */
namespace emu {
struct Barrier {
    std::atomic<unsigned> count{0}, gen{0};
    unsigned total = 64;
    void wait()
    {
        const unsigned g = gen.load(std::memory_order_acquire);
        if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == total) { count.store(0, std::memory_order_relaxed); gen.store(g + 1, std::memory_order_release); }
        else while (gen.load(std::memory_order_acquire) == g) std::this_thread::yield();
    }
};
static Barrier bar;
static uint64_t xch[2][64];
static thread_local int lane_id = 0, which = 0;
static std::atomic<uint64_t> fake_clock{0};
}

#define CH_PRIMITIVES_DEFINED 1
#define CH_DEV static inline
CH_DEV int ch_lane() { return emu::lane_id; }
CH_DEV uint64_t ch_exchange(uint64_t v, uint32_t from)
{
    uint64_t *b = emu::xch[emu::which];
    b[emu::lane_id] = v;
    emu::bar.wait();
    const uint64_t r = b[from & 63u];
    emu::which ^= 1;                    /* two buffers in turn, one barrier per exchange (tools/simt_emu/hip/hip_runtime.h) */
    return r;
}
CH_DEV uint64_t ch_readlane_u64(uint64_t v, uint32_t l) { return ch_exchange(v, l); }
CH_DEV uint64_t ch_shuffle_u64(uint64_t v, uint32_t src) { return ch_exchange(v, src); }     /* (ds_bpermute: each lane names its own source) */
CH_DEV uint64_t ch_ballot(bool p)
{
    uint64_t *b = emu::xch[emu::which];
    b[emu::lane_id] = p ? 1 : 0;
    emu::bar.wait();
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) m |= b[l] << l;
    emu::which ^= 1;
    return m;
}
CH_DEV uint64_t ch_ld64(const uint64_t *p) { return reinterpret_cast<const std::atomic<uint64_t> *>(p)->load(std::memory_order_acquire); }
CH_DEV void ch_st32(uint32_t *p, uint32_t v) { reinterpret_cast<std::atomic<uint32_t> *>(p)->store(v, std::memory_order_release); }
CH_DEV void ch_release() { std::atomic_thread_fence(std::memory_order_release); }
CH_DEV void ch_sleep() { std::this_thread::yield(); }
/* the device reads its clock on the scalar unit: ONE value per wavefront.  Lanes that are threads must be given lane 0's reading — with a
 * reading of their own, the lanes on either side of the watchdog's deadline part ways and the next exchange never completes (the
 * intermittent hang of this check until round 4's third session) */
CH_DEV uint64_t ch_clock() { return ch_exchange(emu::fake_clock.load(std::memory_order_relaxed), 0); }
CH_DEV double ch_f_of_bits(uint64_t b) { b = ~b; double f; std::memcpy(&f, &b, sizeof f); return f; }

#include "crs_chain_resolver.h"

/* ---- chain_resolve() of crs_chain.hip, stated sequentially over all K records ---- */
struct Outcome {
    std::vector<uint32_t> rowstate;
    uint32_t next = 0, wp = 0, nextra = 0, naccept = 0, halt = 0, pk = 0;
};
static Outcome reference(int K, int nW, const std::vector<int64_t> &W, const std::vector<double> &Wf, const std::vector<double> &fT,
                         const std::vector<double> &fM, double f_best, int64_t i0)
{
    Outcome o;
    o.rowstate.assign(nW + 1, 0);
    double xf[CH_EXTRA]; int64_t xrow[CH_EXTRA];
    uint32_t j = 0;
    while (j < (uint32_t) K && !o.halt) {
        double fw = -HUGE_VAL; int64_t rw = -1; int xi = -1;
        if (o.wp < (uint32_t) nW) { fw = Wf[o.wp]; rw = W[o.wp]; }
        for (uint32_t e = 0; e < o.nextra; ++e)
            if (rw < 0 || xf[e] > fw || (xf[e] == fw && xrow[e] > rw)) { fw = xf[e]; rw = xrow[e]; xi = (int) e; }
        if (rw < 0) { o.halt = 1; break; }
        int kind = 0; double fnew = 0;
        if (fT[j] < fw) { kind = 1; fnew = fT[j]; }
        else if (fM[j] < fw) { kind = 2; fnew = fM[j]; }
        if (kind) {
            if (xi >= 0) { xf[xi] = xf[o.nextra - 1]; xrow[xi] = xrow[o.nextra - 1]; --o.nextra; }
            else { o.rowstate[o.wp] = 1u | ((uint32_t) kind << 1) | (j << 3); ++o.wp; }
            ++o.naccept;
            if (nW > 0 && (fnew > Wf[nW - 1] || (fnew == Wf[nW - 1] && rw > W[nW - 1]))) {
                if (o.nextra == CH_EXTRA) o.halt = 1;
                else { xf[o.nextra] = fnew; xrow[o.nextra] = rw; ++o.nextra; }
            }
            if (fnew < f_best || (fnew == f_best && rw < i0)) o.halt = 2u | ((j + 1u) << 8);     /* new best at slot j (crs_chain.hip: later slots are not gathered) */
        }
        j += (kind == 1) ? 1u : 2u;
        if (!o.halt) { o.next = j; o.pk = j | ((j - o.wp) << 16); }
    }
    if (o.halt) { o.next = (uint32_t) K + 2u; o.pk = 0xffffffffu; }
    return o;
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 300;
    const unsigned seed = argc > 2 ? (unsigned) atoi(argv[2]) : 1u;
    std::mt19937_64 rng(seed);
    auto U = [&](int lo, int hi) { return (int) (lo + rng() % (uint64_t) (hi - lo + 1)); };
    long seen_halt = 0, seen_extra = 0, seen_accept = 0, seen_slots = 0, seen_full = 0;
    for (int c = 0; c < cases; ++c) {
        const int K = (c % 7 == 0) ? U(1, 4) : U(1, 256);
        const int nW = (c % 11 == 3) ? U(0, K) : K;            /* a population smaller than the window: the list ends early */
        const int style = c % 5;                                /* how coarse the values are: ties, many landings, many rejections */
        std::vector<int64_t> W(nW + 1, -1);
        std::vector<double> Wf(nW + 1, 0.), fT(K), fM(K);
        /* worst first: decreasing f, ties broken by larger row first (crs_compare: the worst of equal keys is the larger address) */
        double top = 1000.;
        for (int k = 0; k < nW; ++k) {
            top -= (style == 0) ? (double) U(0, 1) : 0.25 * U(0, 8);
            Wf[k] = top;
            W[k] = 5000 - 7 * k - U(0, 3);
        }
        const double f_best = (style == 4) ? top - 5. : top - 400.;
        const int64_t i0 = 17;
        for (int a = 0; a < K; ++a) {
            auto draw = [&]() {
                const int r = U(0, 99);
                if (r < (style == 2 ? 40 : 8)) return 2000. + U(0, 9);                       /* rejected */
                if (r < (style == 1 ? 60 : 20)) return top + 0.25 * U(0, 4 * 40);            /* lands among the worst rows left (maybe) */
                if (r < 21 && style == 4) return f_best - 1.;                                /* a new best: halt */
                if (r < 23) return std::nan("");
                if (r < 24) { uint64_t b = ~0ull; double f; std::memcpy(&f, &b, sizeof f); return f; }   /* the all-ones NaN */
                return top - 1. - U(0, 300);
            };
            fT[a] = draw(); fM[a] = draw();
        }
        const Outcome ref = reference(K, nW, W, Wf, fT, fM, f_best, i0);
        seen_halt += ref.halt != 0; seen_extra += ref.nextra > 0; seen_accept += ref.naccept; seen_slots += K; seen_full += !ref.halt;

        /* device-side memory: ctrl words, records, rowstate */
        std::vector<uint32_t> ctrl(8, 0), rowstate(nW + 1, 0);
        std::vector<uint64_t> recs(2 * (size_t) K, 0);
        /* feeder: a random order; slot a may have to wait until the chain has passed dep[a] < a (or halted) */
        std::vector<int> order(K), dep(K, 0);
        for (int a = 0; a < K; ++a) { order[a] = a; if (a > 0 && U(0, 3) == 0) dep[a] = U(0, a - 1); }
        for (int a = K - 1; a > 0; --a) { const int b = U(std::max(0, a - 40), a); std::swap(order[a], order[b]); }   /* local disorder */
        std::thread feeder([&]() {
            std::vector<char> done(K, 0);
            int left = K;
            while (left) {
                int progressed = 0;
                for (int t = 0; t < K; ++t) {
                    const int a = order[t];
                    if (done[a] == 1) continue;
                    const uint32_t pk = reinterpret_cast<std::atomic<uint32_t> *>(&ctrl[CH_CTRL_PK])->load(std::memory_order_acquire);
                    if (dep[a] > 0 && (pk & 0xffffu) < (uint32_t) dep[a] && pk != 0xffffffffu) continue;
                    /* f(T) first; f(M) at once, a moment later, or only on a later pass of the feeder (the evaluating workgroup publishes
                     * f(T) before it forms the mutation: the resolver must get by without f(M) behind an accepted trial and wait for it
                     * behind a rejected one) */
                    if (done[a] == 0) {
                        reinterpret_cast<std::atomic<uint64_t> *>(&recs[2 * (size_t) a])->store(ch_bits_of_f(fT[a]), std::memory_order_release);
                        ++progressed;
                        const int how = U(0, 3);
                        if (how == 0) { done[a] = 2; continue; }       /* f(M) on a later pass */
                        if (how == 1) std::this_thread::yield();
                    }
                    reinterpret_cast<std::atomic<uint64_t> *>(&recs[2 * (size_t) a + 1])->store(ch_bits_of_f(fM[a]), std::memory_order_release);
                    done[a] = 1; --left; ++progressed;
                    if (U(0, 7) == 0) break;                    /* look at the chain's progress again */
                }
                if (!progressed) std::this_thread::yield();
            }
        });
        std::vector<std::thread> lanes;
        for (int l = 0; l < 64; ++l)
            lanes.emplace_back([&, l]() {
                emu::lane_id = l; emu::which = 0;
                chain_resolver_wave(ctrl.data(), recs.data(), rowstate.data(), K, nW, W.data(), Wf.data(), f_best, i0, ~0ull);
            });
        for (auto &t : lanes) t.join();
        feeder.join();
        int bad = 0;
        if (ctrl[CH_CTRL_NEXT] != ref.next || ctrl[CH_CTRL_PK] != ref.pk || ctrl[CH_CTRL_HALT] != ref.halt || ctrl[CH_CTRL_WP] != ref.wp ||
            ctrl[CH_CTRL_NACCEPT] != ref.naccept || ctrl[CH_CTRL_NEXTRA] != ref.nextra) bad = 1;
        for (int k = 0; k < nW && !bad; ++k) if (rowstate[k] != ref.rowstate[k]) bad = 2;
        if (bad) {
            printf("case %d (K=%d nW=%d style=%d): %s differs: next %u/%u pk %x/%x halt %u/%u wp %u/%u naccept %u/%u nextra %u/%u\n", c, K, nW, style,
                   bad == 1 ? "state" : "rowstate", ctrl[CH_CTRL_NEXT], ref.next, ctrl[CH_CTRL_PK], ref.pk, ctrl[CH_CTRL_HALT], ref.halt,
                   ctrl[CH_CTRL_WP], ref.wp, ctrl[CH_CTRL_NACCEPT], ref.naccept, ctrl[CH_CTRL_NEXTRA], ref.nextra);
            return 1;
        }
    }
    /* the watchdog: nothing is ever evaluated -> the wavefront halts by itself and releases every waiting slot */
    {
        const int K = 8, nW = 8;
        std::vector<int64_t> W(nW, 3); std::vector<double> Wf(nW, 1.);
        std::vector<uint32_t> ctrl(8, 0), rowstate(nW, 0);
        std::vector<uint64_t> recs(2 * K, 0);
        /* the clock runs until the wavefront has left — a clock that makes its jump and STOPS can do so before the lanes have read their
         * starting time, and then the deadline never comes (the other half of this check's intermittent hang) */
        std::atomic<int> lanes_gone{0};
        std::thread clock([&]() { while (!lanes_gone.load(std::memory_order_acquire)) { emu::fake_clock.fetch_add(1000); std::this_thread::yield(); } });
        std::vector<std::thread> lanes;
        for (int l = 0; l < 64; ++l)
            lanes.emplace_back([&, l]() { emu::lane_id = l; emu::which = 0; chain_resolver_wave(ctrl.data(), recs.data(), rowstate.data(), K, nW, W.data(), Wf.data(), 0., 1, 100000); });
        for (auto &t : lanes) t.join();
        lanes_gone.store(1, std::memory_order_release);
        clock.join();
        if (ctrl[CH_CTRL_PK] != 0xffffffffu || ctrl[CH_CTRL_HALT] != 1 || ctrl[CH_CTRL_NEXT] != (uint32_t) K + 2u) { printf("watchdog: no halt\n"); return 1; }
    }
    printf("ok %d  (windows that ran to the end %ld, halted %ld, ended with values among the worst rows %ld; %ld accepts over %ld slots)\n", cases,
           seen_full, seen_halt, seen_extra, seen_accept, seen_slots);
    return 0;
}
