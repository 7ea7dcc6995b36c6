/* tools/scan_fast_check.cpp — CPU check of hip/isres_scan_fast.h (development tooling; tests/test_host_logic.py builds and runs it): the
 * SAME SOURCE hipcc compiles into ev2_scan_fast_kernel's lane walk, compiled by g++, against the lane walk of ev2_scan_kernel restated
 * below (isres_evolve2.hip: "lane = candidate start d: the coordinates one after the other, exactly the serial loop").  The fast walk
 * must return the exact walk's E entry and write the exact walk's T column for every candidate start of every drawn individual, with
 *   - the fast path's exps WRONG by up to +-6 ulp (stands for: the device's exp against itself in the product form, any libm against
 *     any other): the margin, not the accuracy of exp, is what makes the count exact;
 *   - bounds PLANTED 0..3 ulp from a draw's exact value, on either side, for the first draw of a coordinate and for redraws — the
 *     undecided band must be entered (counted) and resolved by the exact expressions;
 *   - windows that end early (deviates run out: -2; window exceeded: -1), candidate starts in front of the stream, coordinates whose
 *     sigma' is capped, mutated-coordinate counts from 0 to 300, degenerate boxes (lb == ub), huge and tiny scales, sigma = 0.
 * What this does not check: the device's scheduling or its LDS staging (a dozen lines of ev2_scan_fast_kernel, the same as
 * ev2_scan_kernel's but for one more array) — tests/staged/test_gpu_isres_fast_scan.py does, on a device.
 *
 *   g++ -O1 -std=c++17 -ffp-contract=off -I nlopt_amd/csrc/hip tools/scan_fast_check.cpp -o tools/_build/scan_fast_check
 *   tools/_build/scan_fast_check [individuals] [seed] [plain]     -> "ok ..." / the first difference; exit code 0 / 1
 *   ("plain": ordinary individuals only — parents inside their boxes, sigma > 0, nothing planted — where NO draw may need the exact path) */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static thread_local std::mt19937_64 *g_rng = nullptr;
static int g_ulp = 6;
static double perturbed_exp(double x)
{
    double v = std::exp(x);
    if (!(v > 0) || std::isinf(v)) return v;
    int k = (int) ((*g_rng)() % (unsigned) (2 * g_ulp + 1)) - g_ulp;
    for (; k > 0; --k) v = std::nextafter(v, INFINITY);
    for (; k < 0; ++k) v = std::nextafter(v, 0.0);
    return v;
}

#define SF_PRIMITIVES_DEFINED
#define SF_DEV static inline
#define SF_EXP_FAST(x) perturbed_exp(x)
#define SF_EXP_EXACT(x) std::exp(x)
#define SF_ISSUE8(a, b, c, d, e, f, g, h) do { } while (0)
#include "isres_scan_fast.h"

#define EVD 256

/* ev2_scan_kernel's lane walk, statement for statement (exp = the function SF_EXP_EXACT names) */
static int walk_exact(int na, int d, int zwlen, bool zw_cut, bool before_stream, double taup, double tau, const double *zw, const double *xi,
                      const double *sg, const double *lo, const double *hi, const double *smax, short *Tcol, long tstride)
{
    const int chunk = (na + 63) >> 6;
    int res = 0;
    if (before_stream || d >= zwlen) res = zw_cut && !before_stream ? -2 : -1;
    const double taup_rand = res == 0 ? taup * zw[d] : 0.0;
    int cur = d + 1, red = 0, cnext = 0, c = 0;
    for (int a = 0; a < na; ++a) {
        if (a == cnext) { if (c < 64) Tcol[(long) c * tstride] = (short) red; ++c; cnext += chunk; }
        if (res != 0) continue;
        if (cur + 1 >= zwlen) { res = zw_cut ? -2 : -1; continue; }
        const double zs = zw[cur], z1 = zw[cur + 1], sa = sg[a], sm_ = smax[a], xa = xi[a], l = lo[a], h = hi[a];
        double s2 = sa * std::exp(taup_rand + tau * zs);
        if (s2 > sm_) s2 = sm_;
        int t = 1;
        double xn = xa + s2 * z1;
        while (xn < l || xn > h) {
            ++t;
            if (cur + t >= zwlen) { res = zw_cut ? -2 : -1; break; }
            xn = xa + s2 * zw[cur + t];
        }
        cur += 1 + t; red += t - 1;
    }
    return res != 0 ? res : 1 + 2 * na + red;
}

static double ulps(double v, int k)
{
    for (; k > 0; --k) v = std::nextafter(v, INFINITY);
    for (; k < 0; ++k) v = std::nextafter(v, -INFINITY);
    return v;
}

int main(int argc, char **argv)
{
    const int individuals = argc > 1 ? atoi(argv[1]) : 400;
    const uint64_t seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    const bool plain = argc > 3 && !strcmp(argv[3], "plain");     /* no planted bounds, no parents on a bound, no empty boxes, no sigma = 0: how often does an ordinary draw need the exact path? */
    std::mt19937_64 rng(seed), prng(seed * 977 + 5);
    g_rng = &prng;
    std::normal_distribution<double> N01(0., 1.);
    std::uniform_real_distribution<double> U(0., 1.);
    unsigned long long lanes = 0, undecided_total = 0, planted_total = 0, negatives = 0, redraws_total = 0, capped = 0;
    for (int it = 0; it < individuals; ++it) {
        const int n = 1 + (int) (rng() % 300);
        int na = (it % 7 == 0) ? (int) (rng() % 4) : (it % 3 == 0 ? n : 1 + (int) (rng() % (unsigned) n));
        if (na > n) na = n;
        const double scale = it % 11 == 0 ? 1e-9 : (it % 13 == 0 ? 1e12 : (it % 5 == 0 ? 5.12 : 1. + 100. * U(rng)));
        const double tau = 1. / std::sqrt(2. * std::sqrt((double) n)), taup = 1. / std::sqrt(2. * (double) n), sqn = std::sqrt((double) n);
        const int ZW = EVD + 3 * na + 65;
        int zwlen = ZW;
        bool zw_cut = false;
        if (it % 9 == 0) { zwlen = (int) (rng() % (unsigned) (ZW + 1)); zw_cut = (rng() & 1) != 0; }      /* the window ends early */
        const long base_neg = it % 17 == 0 ? (long) (rng() % 40) : 0;                                       /* so many candidate starts lie in front of the stream */
        std::vector<double> zw(ZW + 8, 0.), ezw(ZW + 8, 0.), xi(n), sg(n), lo(n), hi(n), smax(n), tol(n);
        for (int q = 0; q < ZW; ++q) zw[q] = (q < base_neg) ? 0.0 : N01(rng);
        for (int a = 0; a < na; ++a) {
            const double c = scale * (U(rng) - 0.5), w = !plain && it % 19 == 0 && a % 5 == 0 ? 0.0 : scale * (0.05 + U(rng));
            lo[a] = c - w; hi[a] = c + w;
            xi[a] = lo[a] + (hi[a] - lo[a]) * U(rng);
            if (!plain && a % 23 == 0) xi[a] = (rng() & 1) ? lo[a] : hi[a];                                           /* a parent on a bound */
            smax[a] = (hi[a] - lo[a]) / sqn;
            const double r = U(rng);
            sg[a] = !plain && r < 0.02 ? 0.0 : (r < 0.3 ? smax[a] * (0.5 + U(rng)) : smax[a] * std::pow(10., -4. * U(rng)));    /* some get capped */
        }
        /* plant bounds a few ulp from exact draw values: walk some candidate start exactly and move a bound of a coordinate next to the
         * value its first draw (or a redraw) takes from there */
        int planted = 0;
        if (!plain && na > 0 && zwlen == ZW && it % 2 == 0) {
            for (int rep = 0; rep < 6; ++rep) {
                const int d = (int) (rng() % EVD);
                if (d < base_neg) continue;
                const double taup_rand = taup * zw[d];
                int cur = d + 1;
                const int atarget = (int) (rng() % (unsigned) na);
                for (int a = 0; a < na && cur + 12 < zwlen; ++a) {
                    double s2 = sg[a] * std::exp(taup_rand + tau * zw[cur]);
                    if (s2 > smax[a]) s2 = smax[a];
                    int t = 1;
                    double xn = xi[a] + s2 * zw[cur + 1];
                    if (a == atarget) {
                        const int k = (int) (rng() % 7) - 3;
                        if (xn > xi[a]) hi[a] = ulps(xn, k); else lo[a] = ulps(xn, k);
                        if (lo[a] > hi[a]) { const double m = lo[a]; lo[a] = hi[a]; hi[a] = m; }
                        ++planted;
                    }
                    while ((xn < lo[a] || xn > hi[a]) && cur + t + 1 < zwlen && t < 10) { ++t; xn = xi[a] + s2 * zw[cur + t]; }
                    cur += 1 + t;
                    if (a == atarget) break;
                }
            }
        }
        for (int a = 0; a < na; ++a) tol[a] = sf_tol(lo[a], hi[a], xi[a]);
        for (int q = 0; q < zwlen; ++q) ezw[q] = sf_stage_e(tau, zw[q]);
        std::vector<short> Te((size_t) 64 * EVD, (short) -77), Tf((size_t) 64 * EVD, (short) -77);
        unsigned und = 0;
        for (int d = 0; d < EVD; ++d) {
            const bool before = d < base_neg;
            const int ee = walk_exact(na, d, zwlen, zw_cut, before, taup, tau, zw.data(), xi.data(), sg.data(), lo.data(), hi.data(), smax.data(), Te.data() + d, EVD);
            const int ef = ev2_walk_fast(na, d, zwlen, zw_cut, before, taup, tau, zw.data(), ezw.data(), xi.data(), sg.data(), lo.data(), hi.data(), smax.data(),
                                         tol.data(), Tf.data() + d, EVD, &und);
            ++lanes;
            negatives += ee < 0;
            if (ee > 0) redraws_total += (unsigned long long) (ee - 1 - 2 * na);
            if (ee != ef) {
                printf("DIFFERENT: individual %d (n %d, na %d, scale %g, zwlen %d of %d, cut %d) start %d: exact %d, fast %d\n", it, n, na, scale, zwlen, ZW, (int) zw_cut, d, ee, ef);
                return 1;
            }
        }
        if (memcmp(Te.data(), Tf.data(), sizeof(short) * Te.size()) != 0) {
            for (size_t q = 0; q < Te.size(); ++q)
                if (Te[q] != Tf[q]) { printf("DIFFERENT T: individual %d chunk %zu start %zu: exact %d, fast %d\n", it, q / EVD, q % EVD, Te[q], Tf[q]); return 1; }
        }
        for (int a = 0; a < na; ++a) capped += sg[a] > smax[a];
        undecided_total += und;
        planted_total += (unsigned long long) planted;
    }
    if (plain && undecided_total != 0) { printf("ordinary draws took the exact path %llu times: the margin is wider than it should be\n", undecided_total); return 1; }
    if (planted_total > 0 && undecided_total == 0) { printf("the planted bounds never reached the undecided band: the check checks nothing\n"); return 1; }
    printf("ok %d individuals, %llu lane walks (%llu ended outside their window), %llu redraws, %llu planted bounds, %llu draws resolved by the exact expressions, exp off by up to %d ulp\n",
           individuals, lanes, negatives, redraws_total, planted_total, undecided_total, g_ulp);
    return 0;
}
