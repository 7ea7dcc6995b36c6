/* crs_shard.hip — CRS2_LM over several GPUs: the population sharded BY COORDINATE.
 *
 * The reference's trial point is x = (2/n)(best + sum of n-1 rows - (n/2) row_jn), one accumulator per coordinate, rows in
 * ascending order (src/algs/crs/crs.c:63-121): the coordinates never mix.  So rank r keeps columns [r*colper, (r+1)*colper) of
 * EVERY row (N x colper doubles: 1/world of the population), runs the very same gather-sum (crs_advance_kernel, told how many
 * columns it has) on its slice — the same additions in the same order as the serial loop, bit for bit, with 1/world of the HBM
 * traffic and no accumulator travelling between ranks — applies the mutation to its slice (crs.c:139-146 is per coordinate too)
 * and writes accepted candidates into its slice of the replaced row (crs.c:153).  What crosses the ranks are the CANDIDATES of a
 * pass: the slices of the trial points that completed, and of their mutations, are ALL-GATHERED (north_star's "all-gather of elite
 * candidates": the exchange ships the slices of ALL K window slots of the pass — 16 n K / world + 16 bytes per rank, fixed size,
 * so that no count has to come back to the host before the collective; only the slots that completed in this pass are read
 * by the evaluation, ~6 of ~14 at the metric configuration) and every rank evaluates f of the assembled points with the same reduction the
 * single-GPU finish kernel uses.  Every rank therefore sees bit-identical f values — identical to a single-GPU run's — takes the
 * identical accept / reject decisions in its own replay of the chain, and no decision, row or index is ever sent.
 *
 * Kernels: the population slice from the MT stream (crs.c:211-219), the slice of the mutation + packing of a pass's candidates,
 * the evaluation of the gathered candidates + the pass's status records.  The gather itself stays crs_advance_kernel (HBM-bound,
 * 8 n (n+1) / world algorithmic bytes per trial and rank); everything here is O(n) per candidate. */
#include "dev_common.h"
#include "../../../include/nlopt_amd.h"

#define SH_WAVES 8                       /* = NLA_FIN_WAVES of crs_kernels.hip = CH_FWAVES of crs_chain.hip: the same f reduction */

/* rows [row_first, row_first + nrows) of the slice from the stream: local column i = global c0 + i, i < nc; pad columns [nc, ld)
 * are zero (the vectorised gather-sum may run over one of them).  One wavefront per row. */
__global__ __launch_bounds__(256) void crs_sh_init_rows_kernel(int n, int c0, int nc, int ld, const double *__restrict__ lb,
                                                                const double *__restrict__ ub, const uint32_t *__restrict__ words,
                                                                int64_t row_first, int64_t nrows, double *__restrict__ X)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const uint32_t *w = words + (size_t) r * 2 * (size_t) n + 2 * (size_t) c0;
    double *xr = X + (size_t) (row_first + r) * (size_t) ld;
    for (int i = lane; i < ld; i += 64) {
        double v = 0;
        if (i < nc) {                           /* k[1+j] = nlopt_urand(lb[j], ub[j]), crs.c:216-218 */
            const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
            v = nla_urand_from(lb[i], ub[i], ww.x, ww.y);
        }
        xr[i] = v;
    }
}

/* for every slot of the window that became complete in this pass: TM[q] = the slice of the local mutation that would follow the
 * trial's rejection (crs.c:139-146; w from the NEXT stream block at the slice's global coordinates), and the slices of both points
 * packed for the all-gather: SEND[(2a) * colper + i] = T_i, SEND[(2a+1) * colper + i] = M_i (i < nc; untouched for other slots);
 * SEND[2 K colper], [+1] = the rank's stop flags (the first one 2 instead of 0 / 1: this rank failed in the pass).  A rank's block of the all-gather is 2 K colper + 2 doubles. */
__global__ __launch_bounds__(256) void crs_sh_mutate_pack_kernel(
    int n, int c0, int nc, int ld, int colper, const double *__restrict__ X, int64_t i0, const double *__restrict__ TX, double *__restrict__ TM,
    const uint32_t *__restrict__ words_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *__restrict__ t_in,
    const int32_t *__restrict__ t_out, int slot_mask, const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ SEND,
    double flag0, double flag1)
{
    const int a = blockIdx.x;
    /* behind the 2K slices: what this rank sees of the per-process stop conditions (force_stop, clock) — agreed by the evaluation kernel */
    if (a == 0 && threadIdx.x == 0) { SEND[(size_t) 2 * K * colper] = flag0; SEND[(size_t) 2 * K * colper + 1] = flag1; }
    if (!(t_out[a] == n && t_in[a] != n)) return;                 /* uniform over the workgroup */
    const uint64_t block = first_block + (uint64_t) a;
    const int q = (int) (block & (uint64_t) slot_mask);
    const double *x = TX + (size_t) q * (size_t) ld, *xb = X + (size_t) i0 * (size_t) ld;
    const uint32_t *w = words_ring + (size_t) ((block + 1) % ring_blocks) * 2 * (size_t) n + 2 * (size_t) c0;
    double *m = TM + (size_t) q * (size_t) ld;
    double *sT = SEND + (size_t) (2 * a) * (size_t) colper, *sM = sT + colper;
    for (int i = threadIdx.x; i < ld; i += blockDim.x) {
        double mv = 0;
        if (i < nc) {                           /* p_i = best_i (1+w) - w p_i, clamp (crs.c:140-145) */
            const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
            const double wv = nla_urand_from(0., 1., ww.x, ww.y);
            const double xi = x[i];
            mv = nla_clamp_box(xb[i] * (1 + wv) - wv * xi, lb[i], ub[i]);
            sT[i] = xi; sM[i] = mv;
        }
        m[i] = mv;
    }
}

/* f of the gathered candidates + the status records of the pass (the evaluation half of crs_finish_kernel on assembled points):
 * RECV is rank-major, rank r's block holding 2K slices of colper doubles + its two stop flags; coordinate g of a point lives in rank
 * g / colper's block.  status[K] (one record behind the window's) = the flags OR-ed over the ranks, t = 1 if a rank reported a failure.
 * The reduction is nla_block_objective<OBJ, 8> over global coordinates — the order of crs_finish_kernel / crs_chain_kernel, so f is
 * what a single-GPU run computes, bit for bit. */
template <int OBJ>
__global__ __launch_bounds__(SH_WAVES * 64) void crs_sh_eval_kernel(
    int n, int colper, uint64_t first_block, int K, const int32_t *__restrict__ t_in, const int32_t *__restrict__ t_out, int slot_mask,
    const double *__restrict__ RECV, int world, double *__restrict__ fT_ring, double *__restrict__ fM_ring,
    nla_crs_slot_status *__restrict__ status, double sign)
{
    __shared__ double scratch[2 * SH_WAVES];
    const int tid = threadIdx.x;
    const int task = blockIdx.x / K, a = blockIdx.x - task * K;
    const int q = (int) ((first_block + (uint64_t) a) & (uint64_t) slot_mask);
    const int t1 = t_out[a];
    const bool newly = (t1 == n) && (t_in[a] != n);              /* uniform over the workgroup */
    double *ring = task == 0 ? fT_ring : fM_ring;
    double f = 0;
    if (newly) {
        const size_t rank_stride = (size_t) 2 * (size_t) K * (size_t) colper + 2, off = (size_t) (2 * a + task) * (size_t) colper;
        auto get = [&](int g) { return RECV[(size_t) (g / colper) * rank_stride + off + (size_t) (g % colper)]; };
        f = sign * nla_block_objective<OBJ, SH_WAVES>(n, get, scratch);
        if (tid == 0) ring[q] = f;
    } else if (t1 == n) f = ring[q];
    if (tid == 0) {
        if (task == 0) { status[a].fT = f; status[a].t = t1; status[a].pad = 0; }
        else status[a].fM = f;
        if (blockIdx.x == 0) {                   /* record K: the ranks' stop flags OR-ed (fT: force_stop, fM: clock) */
            const size_t rank_stride = (size_t) 2 * (size_t) K * (size_t) colper + 2;
            double f0 = 0, f1 = 0;
            int failed = 0;                      /* a rank whose first flag word is 2 failed in this pass (crs_engine.c): everyone leaves */
            for (int r = 0; r < world; ++r) {
                if (RECV[(size_t) r * rank_stride + rank_stride - 2] != 0.) f0 = 1.;
                if (RECV[(size_t) r * rank_stride + rank_stride - 2] == 2.) failed = 1;
                if (RECV[(size_t) r * rank_stride + rank_stride - 1] != 0.) f1 = 1.;
            }
            status[K].fT = f0; status[K].fM = f1; status[K].t = failed; status[K].pad = 0;
        }
    }
}

/* ---- launchers ---------------------------------------------------------------------------------------------------------------- */
extern "C" int nla_k_crs_sh_init_rows(int n, int c0, int nc, int ld, const double *lb, const double *ub, const uint32_t *words,
                                      int64_t row_first, int64_t nrows, double *X, void *stream)
{
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(crs_sh_init_rows_kernel, dim3((unsigned) ((nrows + 3) / 4)), dim3(256), 0, (hipStream_t) stream,
                       n, c0, nc, ld, lb, ub, words, row_first, nrows, X);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_sh_mutate_pack(int n, int c0, int nc, int ld, int colper, const double *X, int64_t i0, const double *TX, double *TM,
                                        const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *t_in,
                                        const int32_t *t_out, int slot_mask, const double *lb, const double *ub, double *SEND,
                                        int flag_forced, int flag_timed, void *stream)
{
    if (K <= 0) return 0;
    if (nc > colper || nc > ld) return (int) hipErrorInvalidValue;
    hipLaunchKernelGGL(crs_sh_mutate_pack_kernel, dim3((unsigned) K), dim3(256), 0, (hipStream_t) stream, n, c0, nc, ld, colper, X, i0, TX, TM,
                       words_ring, ring_blocks, first_block, K, t_in, t_out, slot_mask, lb, ub, SEND, (double) flag_forced, flag_timed ? 1. : 0.);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_sh_eval(int obj, int n, int colper, uint64_t first_block, int K, const int32_t *t_in, const int32_t *t_out,
                                 int slot_mask, const double *RECV, int world, double *fT_ring, double *fM_ring, nla_crs_slot_status *status,
                                 void *stream)
{
    if (K <= 0) return 0;
    const dim3 grid((unsigned) (2 * K)), block(SH_WAVES * 64);
    hipStream_t st = (hipStream_t) stream;
    const double sign = nla_obj_sign(&obj);
    if (colper < 1 || world < 1) return (int) hipErrorInvalidValue;
#define CALL(O) hipLaunchKernelGGL((crs_sh_eval_kernel<O>), grid, block, 0, st, n, colper, first_block, K, t_in, t_out, slot_mask, RECV, \
                                   world, fT_ring, fM_ring, status, sign)
    NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}
