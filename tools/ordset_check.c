/* tools/ordset_check.c — CPU check of the ordered set and the list of worst rows of crs_driver.c (development tooling;
 * tests/test_host_logic.py builds and runs it).  The file is INCLUDED, so the static functions under test are the product's: os_push /
 * os_topk / os_repair (a 4-ary max-heap with the keys in the nodes) and tl_need / tl_worst / tl_accepted / tl_refresh / tl_idle (the
 * worst rows between two looks at the heap: no heap operation per accepted trial, the heap's work done while a window is in flight).
 * Against the obvious statement — the row with the largest (f, row) key of a plain array, at EVERY trial, and the sorted array for
 * every window's list — for drawn populations with MANY TIES (values from a handful of integers), populations smaller than the window
 * (the list is the whole population), lists that run short (redrawn on the spot), redraws while a window is in flight, values that
 * land among the listed rows again and again, acceptance rates from 5 % to 100 %; after a redraw the heap must be a heap, hold every row
 * once, and every node's key must be F[row].
 *   gcc -O1 -std=gnu11 -I nlopt_amd/csrc -I include tools/ordset_check.c -o tools/_build/ordset_check -L oracle -l:libnlopt_amd_emu.so -lm
 *   tools/_build/ordset_check [rounds] [seed]        -> "ok ..." / the first difference; exit code 0 / 1 */
#include "../nlopt_amd/csrc/crs_driver.c"
#include <stdio.h>

static uint64_t rs_ = 88172645463325252ULL;
static uint64_t rnd(void) { rs_ ^= rs_ << 13; rs_ ^= rs_ >> 7; rs_ ^= rs_ << 17; return rs_; }
static double urand(void) { return (double) (rnd() >> 11) * (1.0 / 9007199254740992.0); }

static int64_t naive_worst(const double *F, int64_t N)
{
    int64_t w = 0;
    for (int64_t i = 1; i < N; ++i) if (key_less(F, w, i)) w = i;
    return w;
}
static const double *g_F;
static int cmp_desc(const void *a, const void *b)
{
    const int64_t x = *(const int64_t *) a, y = *(const int64_t *) b;
    return key_less(g_F, x, y) ? 1 : (key_less(g_F, y, x) ? -1 : 0);
}

static int check_heap(const ordset *s, int64_t N)
{
    char *seen = (char *) calloc((size_t) N, 1);
    int bad = 0;
    for (int64_t i = 0; i < s->nheap && !bad; ++i) {
        const osnode v = OSN(s, i);
        if (v.row < 0 || v.row >= N || seen[v.row]) { printf("node %ld holds row %ld (out of range or twice)\n", (long) i, (long) v.row); bad = 1; break; }
        seen[v.row] = 1;
        if (!(v.f == s->F[v.row])) { printf("node %ld: key %g, F[%ld] = %g\n", (long) i, v.f, (long) v.row, s->F[v.row]); bad = 1; }
        if (i > 0 && node_less(OSN(s, (i - 1) / 4), v)) { printf("node %ld larger than its parent\n", (long) i); bad = 1; }
    }
    if (!bad && s->nheap != N) { printf("%ld nodes for %ld rows\n", (long) s->nheap, (long) N); bad = 1; }
    free(seen);
    return bad;
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    unsigned long long windows = 0, accepts = 0, reinserted = 0, refreshes = 0, idle = 0, ties = 0, whole = 0;
    if (argc > 2) rs_ ^= strtoull(argv[2], NULL, 10) * 0x9E3779B97F4A7C15ULL;
    for (int r = 0; r < rounds; ++r) {
        const int64_t N = r % 9 == 0 ? 1 + (int64_t) (rnd() % 40) : 1 + (int64_t) (rnd() % 3000);
        const int Kmax = 1 + (int) (rnd() % 300);
        const int levels = r % 3 == 0 ? 3 : (r % 3 == 1 ? 50 : 0);          /* 3 / 50 distinct values (ties everywhere) / continuous */
        double *F = (double *) malloc(sizeof(double) * (size_t) N);
        int64_t *sorted = (int64_t *) malloc(sizeof(int64_t) * (size_t) N);
        ordset os;
        toplist tl;
        memset(&os, 0, sizeof os);
        if (os_alloc(&os, N, 3 * Kmax + 64) || tl_alloc(&tl, Kmax)) return 2;
        os.F = F;
        for (int64_t i = 0; i < N; ++i) { F[i] = levels ? (double) (rnd() % (unsigned) levels) : urand(); os_push(&os, i); }
        if (check_heap(&os, N)) { printf("after the pushes (round %d, N %ld)\n", r, (long) N); return 1; }
        if (os.best != ({ int64_t b = 0; for (int64_t i = 1; i < N; ++i) if (key_less(F, i, b)) b = i; b; })) { printf("best row wrong after the pushes (round %d)\n", r); return 1; }
        for (int w = 0; w < 16; ++w) {
            const int K = 1 + (int) (rnd() % (unsigned) Kmax);
            const double pacc = w % 4 == 0 ? 1.0 : (w % 4 == 1 ? 0.05 : urand());
            const int mode = (int) (rnd() % 3);                                 /* where accepted values land: anywhere below / just below the worst / far below */
            const int nW = K < N ? K : (int) N;
            const uint64_t r0 = tl.refreshes;
            tl_need(&tl, &os, nW, 3 * K + 16);
            /* the window's list = the nW worst rows of the plain array, in order */
            for (int64_t i = 0; i < N; ++i) sorted[i] = i;
            g_F = F; qsort(sorted, (size_t) N, sizeof(int64_t), cmp_desc);
            if (tl.tail - tl.head < nW) { printf("the list holds %d rows, the window needs %d (round %d window %d)\n", tl.tail - tl.head, nW, r, w); return 1; }
            for (int a = 0; a < nW; ++a)
                if (tl.T[tl.head + a].row != sorted[a] || tl.T[tl.head + a].f != F[sorted[a]]) {
                    printf("list entry %d: row %ld (key %g), the %d-th worst row is %ld (round %d window %d, N %ld, K %d)\n", a, (long) tl.T[tl.head + a].row, tl.T[tl.head + a].f, a, (long) sorted[a], r, w, (long) N, K);
                    return 1;
                }
            ++windows;
            tl.inflight = K;
            if (w % 2 == 0) { const uint64_t q0 = tl.refreshes; tl_idle(&tl, &os); idle += tl.refreshes - q0; }   /* the engine's call between launch and wait */
            for (int t = 0; t < K; ++t) {
                if (tl.head == tl.tail) tl_need(&tl, &os, 1, 3 * K + 16);
                const int64_t worst = tl_worst(&tl), nw = naive_worst(F, N);
                if (worst != nw) { printf("worst row %ld, the array says %ld (round %d window %d trial %d: N %ld K %d list %d whole %d)\n", (long) worst, (long) nw, r, w, t, (long) N, K, tl.tail - tl.head, tl.whole); return 1; }
                if (urand() < pacc) {
                    double v;
                    if (levels) { v = F[worst] - (double) (mode == 1 ? 1 : 1 + rnd() % 3); ++ties; }
                    else v = mode == 1 ? F[worst] * (1.0 - 1e-3 * urand()) : (mode == 2 ? F[worst] * 0.01 * urand() : F[worst] * urand());
                    if (!(v < F[worst])) continue;
                    F[worst] = v;
                    { const int len0 = tl.tail - tl.head; tl_accepted(&tl, &os); if (tl.tail - tl.head == len0) ++reinserted; }
                    ++accepts;
                }
            }
            tl.inflight = 0;
            refreshes += tl.refreshes - r0;
            whole += tl.whole;
            if (w % 5 == 4) {                   /* now and then: everything into the heap, and the heap must be a heap over every row */
                tl_refresh(&tl, &os, 3 * K + 16);
                if (check_heap(&os, N)) { printf("after window %d of round %d (N %ld, K %d)\n", w, r, (long) N, K); return 1; }
            }
        }
        free(F); free(sorted); os_free(&os); tl_free(&tl);
    }
    if (!reinserted || !refreshes || !idle || !ties || !whole) { printf("the draws never reached a re-insertion (%llu), a refresh (%llu), an idle refresh (%llu), ties (%llu) or a list that is the whole population (%llu): the check checks too little\n", reinserted, refreshes, idle, ties, whole); return 1; }
    printf("ok %d populations, %llu windows, %llu accepted replacements (%llu back among the listed rows, %llu with tied values), %llu refreshes of the list (%llu while a window was in flight), %llu windows with the whole population listed\n",
           rounds, windows, accepts, reinserted, ties, refreshes, idle, whole);
    return 0;
}
