/* hbm_ceiling.hip — what HBM delivers on this box for the two access shapes that bound crs_advance_kernel
 * (development / evidence tool, not part of the library):
 *   stream   every byte of an N x ld fp64 matrix once, 16 B per lane, fully coalesced
 *   gather   `rows` random rows of that matrix read in 1 KiB segments (64 lanes x 16 B = the advance kernel's
 *            (slot, 128-coordinate chunk) unit), U independent loads in flight per lane, NO accumulation order
 *            constraint — the ceiling for the gather-sum's access pattern without its serial fp64 chain
 * at launch sizes from one advance pass (~0.8 GB) up to the whole population.
 * build: hipcc --offload-arch=gfx950 -O3 tools/hbm_ceiling.hip -o /tmp/hbm_ceiling ; run: /tmp/hbm_ceiling [n] [N] */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>

__global__ __launch_bounds__(256) void stream_kernel(const double2 *__restrict__ p, size_t count, double *out)
{
    double2 acc = {0, 0};
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t) gridDim.x * 256) { const double2 v = p[i]; acc.x += v.x; acc.y += v.y; }
    if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

/* grid = slots x chunks; workgroup (8 waves) = one 1 KiB column segment of `rows_per_slot` random rows */
template <int U>
__global__ __launch_bounds__(512) void gather_kernel(const double *__restrict__ X, int ld, const int32_t *__restrict__ rows, int rows_per_slot_,
                                                     int chunks, double *out, const int32_t *__restrict__ slot_rows = nullptr)
{
    const int slot = blockIdx.x / chunks, chunk = blockIdx.x % chunks, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t *r = rows + (size_t) slot * rows_per_slot_;
    const int rows_per_slot = slot_rows ? slot_rows[slot] : rows_per_slot_;      /* ragged: this slot sums only its first rows */
    const size_t col = (size_t) (chunk * 64 + lane) * 2;
    double2 acc = {0, 0};
    for (int b = wave * U; b < rows_per_slot; b += 8 * U) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int rr = b + u < rows_per_slot ? r[b + u] : r[0]; v[u] = *reinterpret_cast<const double2 *>(X + (size_t) rr * ld + col); }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; }
    }
    if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    const long N = argc > 2 ? atol(argv[2]) : 100000;
    const int ld = n, chunks = n / 128;
    double *X, *out;
    hipMalloc(&X, sizeof(double) * (size_t) N * ld);
    hipMalloc(&out, 64);
    hipMemset(X, 0, sizeof(double) * (size_t) N * ld);
    const bool realistic = argc > 3 && atoi(argv[3]);           /* 3rd argument 1: random bits in memory, every slot's rows ascending (as Vitter's picks are) */
    if (realistic) {
        std::vector<uint64_t> blk((size_t) 1000 * ld);
        unsigned long long s = 1234567ULL;
        for (auto &v : blk) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (s >> 12) | 0x3ff0000000000000ULL; }
        for (long r0 = 0; r0 < N; r0 += 1000)
            hipMemcpy(X + (size_t) r0 * ld, blk.data(), sizeof(double) * (size_t) ld * (size_t) (N - r0 < 1000 ? N - r0 : 1000), hipMemcpyHostToDevice);
        printf("realistic: random doubles in memory, rows of a slot ascending\n");
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("matrix %ld x %d fp64 = %.2f GB\n", N, n, 8.0 * N * ld / 1e9);
    for (double frac : {0.25, 1.0}) {
        const size_t count = (size_t) (frac * N) * ld / 2;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            stream_kernel<<<256 * 16, 256>>>(reinterpret_cast<const double2 *>(X), count, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("stream  %7.1f MB  %8.3f ms  %7.1f GB/s\n", count * 16 / 1e6, best, count * 16 / 1e6 / best);
    }
    for (int slots : {6, 24, 96}) {
        std::vector<int32_t> h((size_t) slots * n);
        unsigned long long s = 88172645463325252ULL;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int32_t) (s % (unsigned long long) N); }
        if (realistic) for (int sl = 0; sl < slots; ++sl) std::sort(h.begin() + (size_t) sl * n, h.begin() + (size_t) (sl + 1) * n);
        int32_t *rows;
        hipMalloc(&rows, h.size() * 4);
        hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        const double mb = (double) slots * n * n * 8 / 1e6;
        float b16 = 1e30f, b32 = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            hipEventRecord(e0); gather_kernel<16><<<slots * chunks, 512>>>(X, ld, rows, n, chunks, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < b16) b16 = ms;
            hipEventRecord(e0); gather_kernel<32><<<slots * chunks, 512>>>(X, ld, rows, n, chunks, out); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < b32) b32 = ms;
        }
        printf("gather  %2d slots x %d rows x 1 KiB segments = %7.1f MB   U=16: %8.3f ms %7.1f GB/s   U=32: %8.3f ms %7.1f GB/s\n",
               slots, n, mb, b16, mb / b16, b32, mb / b32);
        hipFree(rows);
    }
    {   /* one advance pass as crs_driver sees it at n = 4096, N = 1e5: a window of 22 slots whose remaining work falls off with
         * their distance from the front (hazard stalls), longest pieces first — 6.2 slot-equivalents in total */
        const int slots = 22;
        std::vector<int32_t> h((size_t) slots * n), cnt(slots);
        unsigned long long s = 88172645463325252ULL;
        for (auto &v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int32_t) (s % (unsigned long long) N); }
        double total = 0;
        for (int i = 0; i < slots; ++i) { const int d = slots - 1 - i; cnt[i] = (int) (n * 0.52 / (1.0 + 0.041 * d) * (d == 0 ? 0.6 : 1.0)) + 1; total += cnt[i]; }
        int32_t *rows, *dc;
        hipMalloc(&rows, h.size() * 4); hipMalloc(&dc, slots * 4);
        hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dc, cnt.data(), slots * 4, hipMemcpyHostToDevice);
        const double mb = total * n * 8 / 1e6;
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            float ms;
            hipEventRecord(e0); gather_kernel<32><<<slots * chunks, 512>>>(X, ld, rows, n, chunks, out, dc); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("ragged  22 slots, %.1f slot-equivalents = %7.1f MB   U=32: %8.3f ms %7.1f GB/s   (no order constraint, no plan)\n", total / n, mb, best, mb / best);
    }
    return 0;
}
