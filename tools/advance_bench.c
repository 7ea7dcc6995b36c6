/* advance_bench.c — microbenchmark of the CRS gather kernels through the kernel-level C-ABI
 * (include/nlopt_amd.h part 2): synthetic population and picks, tiling variants of the
 * resumable advance kernel (bitwise comparison between variants + HIP-event timing).  Development tool only.
 * build: gcc -O2 tools/advance_bench.c -Iinclude -Lnlopt_amd/lib -lnlopt_amd -Wl,-rpath,$PWD/nlopt_amd/lib -o /tmp/advance_bench */
#include "nlopt_amd.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rs = 88172645463325252ULL;
static uint64_t xr(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static int cmp32(const void *a, const void *b) { int32_t x = *(const int32_t *) a, y = *(const int32_t *) b; return (x > y) - (x < y); }

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const int64_t N = argc > 2 ? atoll(argv[2]) : 100000;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;   /* 1: consecutive rows (locality probe) */
    const int KMAX = 64, ld = (n + 1) & ~1, reps = 5;
    const int64_t i0 = N / 3;
    void *st;
    double *d_X, *d_TX, *d_TXo, *d_lb, *d_ub, *h_chunk, *h_a, *h_b;
    int32_t *d_jn, *d_pos, *d_last, *h_jn, *h_pos, *h_last, *d_t0, *d_t1, *h_t;
    int64_t *d_W, *h_W;
    void *e0, *e1;
    if (nla_dev_count() <= 0) { fprintf(stderr, "no device\n"); return 1; }
    st = nla_stream_create(); e0 = nla_event_create(); e1 = nla_event_create();
    d_X = nla_dev_malloc(sizeof(double) * (size_t) N * ld);
    d_TX = nla_dev_malloc(sizeof(double) * KMAX * ld); d_TXo = nla_dev_malloc(sizeof(double) * KMAX * ld);
    d_lb = nla_dev_malloc(sizeof(double) * ld); d_ub = nla_dev_malloc(sizeof(double) * ld);
    d_jn = nla_dev_malloc(4 * KMAX); d_last = nla_dev_malloc(4 * KMAX); d_pos = nla_dev_malloc(4 * (size_t) KMAX * n);
    d_t0 = nla_dev_malloc(4 * KMAX); d_t1 = nla_dev_malloc(4 * KMAX); d_W = nla_dev_malloc(8 * KMAX);
    h_jn = malloc(4 * KMAX); h_last = malloc(4 * KMAX); h_pos = malloc(4 * (size_t) KMAX * n); h_t = malloc(4 * KMAX);
    h_W = malloc(8 * KMAX); h_a = malloc(sizeof(double) * KMAX * ld); h_b = malloc(sizeof(double) * KMAX * ld);
    {   /* population */
        const size_t rows_per = 2048;
        h_chunk = malloc(sizeof(double) * rows_per * ld);
        for (int64_t r = 0; r < N; r += rows_per) {
            size_t nr = (size_t) (N - r < (int64_t) rows_per ? N - r : (int64_t) rows_per);
            for (size_t i = 0; i < nr * ld; ++i) h_chunk[i] = (double) (xr() >> 11) * (1.0 / 9007199254740992.0) * 1100.0 - 500.0;
            nla_memcpy_h2d(d_X + (size_t) r * ld, h_chunk, sizeof(double) * nr * ld, st);
            nla_stream_sync(st);
        }
        for (int i = 0; i < ld; ++i) h_chunk[i] = -500.0;
        nla_memcpy_h2d(d_lb, h_chunk, sizeof(double) * ld, st); nla_stream_sync(st);
        for (int i = 0; i < ld; ++i) h_chunk[i] = 600.0;
        nla_memcpy_h2d(d_ub, h_chunk, sizeof(double) * ld, st); nla_stream_sync(st);
    }
    for (int s = 0; s < KMAX; ++s) {    /* n distinct ascending reduced positions */
        int32_t *p = h_pos + (size_t) s * n;
        char *mark = calloc((size_t) N, 1);
        int got = 0;
        if (mode == 1) { int32_t r0 = (int32_t) (xr() % (uint64_t) (N - 2 - n)); for (got = 0; got < n; ++got) p[got] = r0 + got; }
        while (got < n) { int32_t r = (int32_t) (xr() % (uint64_t) (N - 2)); if (!mark[r]) { mark[r] = 1; p[got++] = r; } }
        free(mark);
        qsort(p, (size_t) n, 4, cmp32);
        h_last[s] = p[n - 1] - (p[n - 2] + 1);
        p[n - 1] = p[n - 2] + 1;
        h_jn[s] = (int32_t) (xr() % (uint64_t) n);
    }
    nla_memcpy_h2d(d_jn, h_jn, 4 * KMAX, st); nla_memcpy_h2d(d_last, h_last, 4 * KMAX, st);
    nla_memcpy_h2d(d_pos, h_pos, 4 * (size_t) KMAX * n, st); nla_stream_sync(st);

    static const int Ks[] = { 1, 2, 4, 6, 8, 16 };
    static const int variants[] = { 416, 432, 816, 832, 1616, 10832, 10864 };
    printf("n=%d N=%lld  bytes/trial=%.2f MB\n", n, (long long) N, 8.0 * n * (n + 1) / 1e6);
    for (size_t ki = 0; ki < sizeof Ks / sizeof *Ks; ++ki) {
        const int K = Ks[ki];
        const double gb = (double) K * 8.0 * n * (n + 1) / 1e9;
        float ms = 0, best = 1e30f;
        int have_ref = 0;
        for (size_t vi = 0; vi < sizeof variants / sizeof *variants; ++vi) {
            best = 1e30f;
            for (int r = 0; r < reps; ++r) {
                nla_event_record(e0, st);
                nla_memset(d_t0, 0, 4 * KMAX, st);
                if (nla_k_crs_advance(n, ld, d_X, i0, d_jn, d_pos, d_last, KMAX, 0, K, d_W, 0, d_t0, d_t1, KMAX - 1,
                                      d_lb, d_ub, d_TX, variants[vi], st)) { printf("launch failed\n"); return 1; }
                nla_event_record(e1, st); nla_stream_sync(st);
                ms = nla_event_elapsed_ms(e0, e1); if (ms < best) best = ms;
            }
            nla_memcpy_d2h(h_b, d_TX, sizeof(double) * (size_t) K * ld, st); nla_stream_sync(st);
            if (!have_ref) { memcpy(h_a, h_b, sizeof(double) * (size_t) K * ld); have_ref = 1; }
            printf("K=%2d  advance(v=%4d)  %8.3f ms %8.1f GB/s  %s\n", K, variants[vi], best, gb / (best * 1e-3),
                   memcmp(h_a, h_b, sizeof(double) * (size_t) K * ld) ? "MISMATCH" : "same bits as first variant");
        }
        /* two-pass: stop every slot a>0 at one of its picks, then finish */
        {
            int nW = K - 1 > 0 ? K - 1 : 0;
            for (int j = 0; j < nW; ++j) {          /* W[j] = a row sampled by slot j+1 (so slot j+1 must stop there) */
                int32_t rho = h_pos[(size_t) (j + 1) * n + (size_t) (xr() % (uint64_t) (n - 1))];
                h_W[j] = rho + (rho >= i0);
            }
            nla_memcpy_h2d(d_W, h_W, 8 * (size_t) (nW ? nW : 1), st);
            nla_event_record(e0, st);
            nla_memset(d_t0, 0, 4 * KMAX, st);
            nla_k_crs_advance(n, ld, d_X, i0, d_jn, d_pos, d_last, KMAX, 0, K, d_W, nW, d_t0, d_t1, KMAX - 1, d_lb, d_ub, d_TX, 0, st);
            nla_k_crs_advance(n, ld, d_X, i0, d_jn, d_pos, d_last, KMAX, 0, K, d_W, 0, d_t1, d_t0, KMAX - 1, d_lb, d_ub, d_TX, 0, st);
            nla_event_record(e1, st);
            nla_memcpy_d2h(h_t, d_t1, 4 * K, st);
            nla_memcpy_d2h(h_b, d_TX, sizeof(double) * (size_t) K * ld, st); nla_stream_sync(st);
            int partial = 0; for (int a = 0; a < K; ++a) partial += h_t[a] < n;
            printf("K=%2d  advance 2-pass   %8.3f ms  (%d slots stopped early in pass 1)  %s\n", K, nla_event_elapsed_ms(e0, e1), partial,
                   memcmp(h_a, h_b, sizeof(double) * (size_t) K * ld) ? "MISMATCH" : "bit-identical");
        }
    }
    return 0;
}
