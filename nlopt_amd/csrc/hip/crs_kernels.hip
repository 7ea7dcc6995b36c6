/* crs_kernels.hip — Controlled Random Search (CRS2_LM) on gfx950.
 *
 * Reference loops replaced: K1/K2 population init + evaluation
 * (src/algs/crs/crs.c:211-226), K3 Vitter method-A selection (crs.c:89-109), K4 the
 * centroid/reflection gather-sum (crs.c:101-120), K5 local mutation (crs.c:139-146), K6 the data
 * the in-order commit needs, and the row write-back (crs.c:153).
 *
 * Data layout in HBM: X is N x ld fp64, row-major, ld = n rounded up to even so every row starts
 * 16-byte aligned (the reference interleaves f as column 0 of an (n+1)-wide row, crs.c:43; f lives
 * in a separate host array here).  Trial points TX/TM are K x ld.  The MT word stream for a run of
 * 2n-word blocks is a flat uint32 array; the pre-digested selection of block b is pos[b*n ..].
 *
 * Roofline: crs_gather_kernel is the hot kernel — HBM-bound, 8*n*(n+1) algorithmic bytes per
 * trial (n random rows + the best row).  Parallelism: the n coordinates of one trial (one lane
 * per coordinate pair, rows accumulated in the reference's order so x is bit-identical) times K
 * speculative trials per launch; memory-level parallelism comes from U independent row loads in
 * flight per lane.  Everything else here is O(n) or O(N) per trial and off the roofline.
 */
#include "dev_common.h"
#include "../../../include/nlopt_amd.h"
#include <limits.h>

/* ------------------------------------------------------------------------------------------------
 * K1+K2: rows of the initial population from the MT stream, evaluated in the same pass.
 * One wavefront per row; lane l writes coordinates l, l+64, ... (coalesced 512 B stores).
 * ---------------------------------------------------------------------------------------------- */
template <int OBJ>
__global__ __launch_bounds__(256) void crs_init_rows_kernel(int n, int ld, const double *__restrict__ lb,
                                                             const double *__restrict__ ub,
                                                             const uint32_t *__restrict__ words, int64_t row_first,
                                                             int64_t nrows, double *__restrict__ X, double *__restrict__ F, double sign)
{
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const uint32_t *w = words + (size_t) r * 2 * (size_t) n;
    double *xr = X + (size_t) (row_first + r) * (size_t) ld;
    auto gen = [&](int i) {                      /* k[1+j] = nlopt_urand(lb[j], ub[j]), crs.c:216-218 */
        const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
        return nla_urand_from(lb[i], ub[i], ww.x, ww.y);
    };
    if (X) for (int i = lane; i < n; i += 64) xr[i] = gen(i);      /* X == NULL: the values only (a column-sharded run keeps slices, crs_shard.hip) */
    if (OBJ >= 0) {
        double f = nla_wave_objective<(OBJ >= 0 ? OBJ : 0)>(n, gen);
        if (lane == 0) F[row_first + r] = sign * f;
    }
}

/* generic batched evaluation: one wavefront per candidate */
template <int OBJ>
__global__ __launch_bounds__(256) void eval_kernel(int n, int ld, const double *__restrict__ P, int64_t count,
                                                    double *__restrict__ F, double sign)
{
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= count) return;
    const double *x = P + (size_t) c * (size_t) ld;
    double f = nla_wave_objective<OBJ>(n, [&](int i) { return x[i]; });
    if (lane == 0) F[c] = sign * f;
}

/* ------------------------------------------------------------------------------------------------
 * K3: Vitter method A, one lane per 2n-word stream block.
 * The reference's chain (crs.c:93-100) is, per visited row: q = (q*Nfree)/Nleft while q > v.  It
 * is inherently serial in fp64 (each q is a rounded mul + a rounded div of the previous one), so
 * the parallelism is across *future* blocks: every trial consumes exactly 2n words
 * [jn | v_0..v_{n-2} | last] regardless of history, hence block b's selection
 * is a pure function of its words and (n, N).  The skip loop and the pick step are fused into one
 * loop with exactly one IEEE division per visited row, so the lanes of a wavefront stay
 * converged (each runs ~N(n-1)/n iterations).  Positions are in "reduced" space (the best row
 * removed); the best's index is applied when the block is used.
 * ---------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void crs_vitter_kernel(int n, int64_t N, const uint32_t *__restrict__ words, int nblocks,
                                                         int32_t *__restrict__ jn_out, int32_t *__restrict__ pos,
                                                         int32_t *__restrict__ last)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t *w = words + (size_t) b * 2 * (size_t) n;
    int32_t *out = pos + (size_t) b * (size_t) n;
    jn_out[b] = (int32_t) (w[0] % (uint32_t) n);                 /* nlopt_iurand(n), crs.c:72 */
    int Nleft = (int) (N - 1), nleft = n, Nfree = Nleft - nleft; /* crs.c:90-91 */
    int r = 0, t = 0;
    if (n > 1) {
        double q = ((double) Nfree) / Nleft;
        double v = nla_res53(w[1], w[2]);
        double vn = (n > 2) ? nla_res53(w[3], w[4]) : 0.0;       /* prefetched uniform of the next pick */
        for (;;) {
            const bool take = !(q > v);
            if (take) {
                out[t] = r;
                ++t; --nleft;
                if (nleft == 1) { ++r; --Nleft; break; }
                v = vn;
                if (t + 1 < n - 1) vn = nla_res53(w[1 + 2 * (t + 1)], w[2 + 2 * (t + 1)]);
            }
            const double num = take ? (double) Nfree : q * (double) (Nfree - 1);
            Nfree -= take ? 0 : 1;
            --Nleft; ++r;
            q = num / (double) Nleft;
        }
    }
    out[n - 1] = r;
    last[b] = (int32_t) (w[2 * n - 1] % (uint32_t) Nleft);       /* nlopt_iurand(Nleft), crs.c:109 */
}

#include "crs_common.h"

/* ------------------------------------------------------------------------------------------------
 * K4': resumable gather-sum ("advance").  The reference accumulates the n sampled rows of a trial
 * in ascending row order into one accumulator per coordinate (crs.c:101-114); fp64 addition is
 * not associative, so that order is the contract.  A slot (= one stream block speculated as a
 * reflection trial) carries (t, acc): picks [0,t) are already summed into acc (kept in its TX
 * row).  One pass advances every slot of the window from its t to
 *     e = the first pick >= t whose row may still be overwritten before the slot's turn,
 * i.e. a row among W[0..d), the d worst rows at the time of the pass, d = the slot's distance
 * from the front of the window (each earlier block commits at most once, and every commit
 * overwrites the then-worst row, so the rows overwritten before the slot's turn are a subset of
 * that prefix).  Picks < e read rows whose content is already final for this slot, so no byte is
 * read twice and nothing speculative is ever discarded because of a write hazard.
 *
 * grid = window slots x coordinate chunks; one workgroup of WAVES wavefronts per (slot, chunk).
 * The per-coordinate chain is serial, the loads are not: the WAVES wavefronts take the batches of
 * U rows round-robin, each keeps its next batch in flight (U x 16 B per lane) while the
 * accumulator travels through LDS from wavefront to wavefront in batch order as a token (an LDS
 * turn counter; only the 2U fp64 adds of a batch and the hand-off are on the serial path) —
 * WAVES*U rows of one chunk are in flight instead of U.
 * ---------------------------------------------------------------------------------------------- */
/* The small per-pass lists can travel as KERNEL ARGUMENTS instead of through a host-to-device copy in front of the pass
 * (one dependent stream operation and its gap less per pass): inl != 0 -> use these, else the device pointers. */
#define NLA_KA_MAX 128
struct crs_lists { int inl; int32_t t_in[NLA_KA_MAX]; int64_t W[NLA_KA_MAX]; };
struct crs_commits { int inl; int32_t slot[NLA_KA_MAX], kind[NLA_KA_MAX]; int64_t row[NLA_KA_MAX]; };
/* The commits staged since the previous pass, done BY the advance launch instead of by a launch of their own in front of it (round 4:
 * one launch and its gap less per pass — 2.7 + 3.8 us of a 30 us pass at n = 512): nc extra workgroups at the end of the grid copy
 * the accepted points from their trial-point slots (src = kind << 16 | slot: TX if kind == 1, TM otherwise) into rows row[c] of X, and
 * every READ of such a row by the gather workgroups of the same launch — a pick, or the best row a fresh slot starts from — is
 * FORWARDED to the slot it is being copied from, so no workgroup reads a row that another one is writing.  The host guarantees the
 * rows are distinct and no source slot is one of this pass's own slots (crs_engine.c). */
#define NLA_KC_MAX 16
struct crs_fwd { int nc; int32_t src[NLA_KC_MAX]; int64_t row[NLA_KC_MAX]; };

typedef const __attribute__((address_space(4))) crs_fwd *crs_kernarg_fwd;
__device__ __forceinline__ int32_t crs_fwd_of(crs_kernarg_fwd Fk, int64_t r)
{
    int32_t s_ = -1;
    const int nc = Fk->nc;
    for (int c = 0; c < nc; ++c) if (Fk->row[c] == r) s_ = Fk->src[c];
    return s_;
}
__device__ __forceinline__ const char *crs_row_ptr(int32_t code, const double *X, const double *TX, const double *TM, int ld)      /* wave-uniform */
{
    if (code >= 0) return reinterpret_cast<const char *>(X + (size_t) code * (size_t) ld);
    const int sc = -(code + 1);
    return reinterpret_cast<const char *>(((sc >> 16) == 1 ? TX : TM) + (size_t) (sc & 0xffff) * (size_t) ld);
}

template <int VEC, int U, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void crs_advance_kernel(
    const crs_fwd F_first_kernel_argument,      /* read through the kernarg segment (offset 0), never by name: indexing the by-value copy with a
                                                 * run-time c makes the compiler move it to scratch memory (see crs_chain.hip) */
    int n, int ncol, int ld, const double *__restrict__ X, int64_t i0, const int32_t *__restrict__ jn_ring,
    const int32_t *__restrict__ pos_ring, const int32_t *__restrict__ last_ring, uint32_t ring_blocks,
    uint64_t first_block, int K, const int64_t *__restrict__ W, int nW,
    const int32_t *__restrict__ t_in, int32_t *__restrict__ t_out, int slot_mask, int chunks,
    const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ TX, const crs_lists L,
    double *__restrict__ Xw, const double *__restrict__ TM)
{
    typedef typename VecT<VEC>::T V;
    const crs_kernarg_fwd Fk = (crs_kernarg_fwd) __builtin_amdgcn_kernarg_segment_ptr();
    (void) F_first_kernel_argument;
#define F (*Fk)
    if ((int) blockIdx.x >= K * chunks) {       /* a staged commit (crs.c:153): accepted point -> its row of X */
        const int c = (int) blockIdx.x - K * chunks;
        const int sc = F.src[c];
        const double *src = ((sc >> 16) == 1 ? TX : TM) + (size_t) (sc & 0xffff) * (size_t) ld;
        double *dst = Xw + (size_t) F.row[c] * (size_t) ld;
        for (int i = threadIdx.x; i < ncol; i += WAVES * 64) dst[i] = src[i];
        return;
    }
    static_assert(U <= 64, "one lane per row of a batch");
    __shared__ V sacc[64];
    __shared__ int32_t srow[NLA_ADV_RCAP];
    __shared__ int s_e;
    __shared__ int s_turn;
    /* polled through an explicit LDS-address-space pointer: volatile accesses through a generic
     * pointer become flat loads, whose waits would also drain the global loads in flight */
    volatile __attribute__((address_space(3))) int *turn = (volatile __attribute__((address_space(3))) int *) &s_turn;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    /* back of the window first: fresh slots have the longest pieces, the front ones the shortest */
    const int a = K - 1 - (int) (blockIdx.x / chunks), chunk = blockIdx.x % chunks;
    const uint64_t block = first_block + (uint64_t) a;
    const uint32_t rb = (uint32_t) (block % ring_blocks);
    const int q = (int) (block & (uint64_t) slot_mask);
    const int32_t *p = pos_ring + (size_t) rb * (size_t) n;
    const int jn = jn_ring[rb];
    const int t0 = L.inl ? L.t_in[a] : t_in[a];
    /* last pick: i += iurand(Nleft); i += i == i0  (crs.c:109) */
    const int64_t rbase = p[n - 1];
    int64_t al = rbase + (rbase >= i0 ? 1 : 0) + (int64_t) last_ring[rb];
    al += (al == i0) ? 1 : 0;

    /* actual rows of picks t0 .. t0+cnt0-1 into LDS (ascending): one coalesced read of the pick
     * list serves both the plan below and the first segment of the sum */
    auto pick_row = [&](int t) -> int32_t {
        int64_t r;
        if (t < n - 1) { r = p[t]; r += (r >= i0 ? 1 : 0); } else r = al;
        return (int32_t) r;
    };
    /* (row r of X as this launch may read it: the row itself, or — if a commit of this launch is writing it — the slot it comes from,
     * encoded as -(1 + src) in the staged pick list AFTER the plan below has searched the list by row number: crs_fwd_of / crs_row_ptr) */
#define fwd_of(r_) crs_fwd_of(Fk, (int64_t) (r_))
#define row_ptr(code_) crs_row_ptr((code_), X, TX, TM, ld)
    const int cnt0 = (n - t0 < NLA_ADV_RCAP) ? n - t0 : NLA_ADV_RCAP;
    for (int i = threadIdx.x; i < cnt0; i += WAVES * 64) srow[i] = pick_row(t0 + i);
    __syncthreads();
    if (wave == 0) {                        /* plan: where must this slot stop in this pass? */
        int e = n;
        const int nun = a < nW ? a : nW;
        for (int j = lane; j < nun; j += 64) {
            const int64_t r = L.inl ? L.W[j] : W[j];
            if (r == i0) continue;          /* the best row is never sampled */
            int lo = 0, hi = cnt0 - 1;      /* binary search among the staged picks */
            bool found = false;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const int32_t pv = srow[mid];
                if (pv == (int32_t) r) { e = e < t0 + mid ? e : t0 + mid; found = true; break; }
                if (pv < (int32_t) r) lo = mid + 1; else hi = mid - 1;
            }
            if (!found && t0 + cnt0 < n) {  /* very long pick lists: the rest straight from memory */
                if (r == al) { e = e < n - 1 ? e : n - 1; continue; }
                const int32_t rho = (int32_t) (r - (r > i0 ? 1 : 0));
                lo = t0 + cnt0; hi = n - 2;
                while (lo <= hi) {
                    const int mid = (lo + hi) >> 1;
                    const int32_t pv = p[mid];
                    if (pv == rho) { e = e < mid ? e : mid; break; }
                    if (pv < rho) lo = mid + 1; else hi = mid - 1;
                }
            }
        }
        e = nla_wave_min_i32(e);
        if (e < t0) e = t0;
        if (lane == 0) s_e = e;
    }
    __syncthreads();
    const int e = s_e;
    if (e == t0) {
        if (chunk == 0 && threadIdx.x == 0) t_out[a] = t0;
        return;
    }
    if (F.nc > 0) {                         /* (uniform) forward the staged picks that a commit of this launch is writing */
        const int lim = (e - t0 < cnt0) ? e - t0 : cnt0;
        for (int i = threadIdx.x; i < lim; i += WAVES * 64) { const int32_t s_ = fwd_of(srow[i]); if (s_ >= 0) srow[i] = -(1 + s_); }
    }
    /* n rows are summed; ncol coordinates of them are held here (ncol == n: the whole rows; a column slice on several GPUs, crs_shard.hip) */
    const int col = (chunk * 64 + lane) * VEC;
    const bool active = col < ncol;
    const size_t colc = active ? (size_t) col : 0;
    const double *Xc = X + colc;
    double *accrow = TX + (size_t) q * (size_t) ld + colc;
    if (wave == 0) {                        /* x := best (crs.c:69), or resume */
        const int32_t sb_ = (t0 == 0 && F.nc > 0) ? fwd_of(i0) : -1;
        const double *bestc = sb_ >= 0 ? reinterpret_cast<const double *>(row_ptr(-(1 + sb_))) + colc : Xc + (size_t) i0 * (size_t) ld;
        sacc[lane] = (t0 == 0) ? ldv<VEC>(bestc) : ldv<VEC>(accrow);
    }

    const double hneg = -(0.5 * n);         /* x -= xi*(0.5*n)  ==  x += xi*(-(0.5*n)), exactly */
    const uint32_t lane_off = (uint32_t) (colc * sizeof(double));
    V v[U];
    for (int seg0 = t0; seg0 < e; seg0 += NLA_ADV_RCAP) {
        const int cnt = (e - seg0 < NLA_ADV_RCAP) ? e - seg0 : NLA_ADV_RCAP;
        nla_lds_barrier();                  /* the previous segment's row list is no longer needed */
        if (seg0 != t0)                     /* (the first segment was staged for the plan) */
            for (int i = threadIdx.x; i < cnt; i += WAVES * 64) {
                int32_t r_ = pick_row(seg0 + i);
                if (F.nc > 0) { const int32_t s_ = fwd_of(r_); if (s_ >= 0) r_ = -(1 + s_); }
                srow[i] = r_;
            }
        if (threadIdx.x == 0) *turn = 0;
        nla_lds_barrier();
        const int nb = (cnt + U - 1) / U;
        auto issue = [&](int b) {
            const int base = b * U;
            const int mine = base + lane < cnt ? base + lane : cnt - 1;
            const int32_t myrow = srow[mine];
#pragma unroll
            for (int u = 0; u < U; ++u) {        /* unconditional: lanes past the end hold the last row */
                const char *rowp = row_ptr(__builtin_amdgcn_readlane(myrow, u));   /* wave-uniform: the row of X, or the slot a commit of this launch copies it from */
                v[u] = *reinterpret_cast<const V *>(rowp + lane_off);
            }
        };
        /* The accumulator is a token: batch ph may only be added by its wavefront once batch ph-1 has
         * been (turn == ph).  Only the adds are on the serial path; a wavefront issues the loads of
         * its next batch after it has passed the token on. */
        if (wave < nb) issue(wave);
        for (int ph = wave; ph < nb; ph += WAVES) {
            while (*turn != ph) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            V acc = sacc[lane];
            const int base = ph * U, tb = seg0 + base;
            if (base + U <= cnt && !(jn >= tb && jn < tb + U)) {
#pragma unroll
                for (int u = 0; u < U; ++u) add_row(acc, v[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (base + u < cnt) acc_row(acc, v[u], (tb + u == jn) ? hneg : 1.0);
            }
            sacc[lane] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *turn = ph + 1;
            if (ph + WAVES < nb) issue(ph + WAVES);
        }
        if (wave == 0) {                    /* the segment is summed when the token has left its last batch */
            while (*turn != nb) __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    }
    if (wave == 0 && active) {
        V acc = sacc[lane];
        if (e == n) {                       /* x[k] *= 2.0 / n, then clamp (crs.c:116-120) */
            const double s = 2.0 / n;
            if constexpr (VEC == 1) {
                double a0 = *reinterpret_cast<double *>(&acc);
                *accrow = nla_clamp_box(a0 * s, lb[col], ub[col]);
            } else {
                double2 a2 = *reinterpret_cast<double2 *>(&acc), r2;
                r2.x = nla_clamp_box(a2.x * s, lb[col], ub[col]);
                r2.y = nla_clamp_box(a2.y * s, lb[col + 1], ub[col + 1]);
                *reinterpret_cast<double2 *>(accrow) = r2;
            }
        } else {
            *reinterpret_cast<V *>(accrow) = acc;
        }
    }
    if (chunk == 0 && threadIdx.x == 0) t_out[a] = e;
#undef F
#undef fwd_of
#undef row_ptr
}

/* finish kernel: for every slot of the window that became complete in this pass, f of the trial
 * and the local mutation that would follow its rejection (crs.c:139-146; w from the NEXT stream
 * block) with its f; for every slot, its status record for the host's in-order walk. */
#define NLA_FIN_WAVES 8
template <int OBJ>
__global__ __launch_bounds__(NLA_FIN_WAVES * 64) void crs_finish_kernel(
    int n, int ld, const double *__restrict__ X, int64_t i0, const double *__restrict__ TX, double *__restrict__ TM,
    const uint32_t *__restrict__ words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
    const int32_t *__restrict__ t_in, const int32_t *__restrict__ t_out, int slot_mask,
    const double *__restrict__ lb, const double *__restrict__ ub, double *__restrict__ fT_ring,
    double *__restrict__ fM_ring, nla_crs_slot_status *__restrict__ status, const crs_lists L, double sign,
    uint32_t *__restrict__ bell_count, uint32_t *__restrict__ bell, uint32_t bell_seq)
{
    __shared__ double scratch[2 * NLA_FIN_WAVES];
    const int tid = threadIdx.x;
    const int task = blockIdx.x / K, a = blockIdx.x - task * K;
    const uint64_t block = first_block + (uint64_t) a;
    const int q = (int) (block & (uint64_t) slot_mask);
    const int t1 = t_out[a];
    const bool was_done = (L.inl ? L.t_in[a] : t_in[a]) == n;
    const bool newly = (t1 == n) && !was_done;          /* uniform over the workgroup */
    const double *x = TX + (size_t) q * (size_t) ld;
    if (task == 0) {
        double f = 0;
        if (OBJ >= 0) {
            if (newly) {
                f = sign * nla_block_objective<(OBJ >= 0 ? OBJ : 0), NLA_FIN_WAVES>(n, [&](int i) { return x[i]; }, scratch);
                if (tid == 0) fT_ring[q] = f;
            } else if (t1 == n) f = fT_ring[q];
        }
        if (tid == 0) { status[a].fT = f; status[a].t = t1; }
    } else {
        double f = 0;
        if (OBJ >= 0 || OBJ == -2) {            /* -2: mutate only — the objective is a user-supplied kernel run by the caller */
            if (newly) {
                const double *xb = X + (size_t) i0 * (size_t) ld;
                const uint32_t *w = words_ring + (size_t) ((block + 1) % ring_blocks) * 2 * (size_t) n;
                double *m = TM + (size_t) q * (size_t) ld;
                auto mut = [&](int i) {        /* p_i = best_i (1+w) - w p_i, clamp (crs.c:140-145) */
                    const uint2 ww = *reinterpret_cast<const uint2 *>(w + 2 * i);
                    const double wv = nla_urand_from(0., 1., ww.x, ww.y);
                    return nla_clamp_box(xb[i] * (1 + wv) - wv * x[i], lb[i], ub[i]);
                };
                for (int i = tid; i < n; i += NLA_FIN_WAVES * 64) m[i] = mut(i);
                if (OBJ >= 0) {
                    f = sign * nla_block_objective<(OBJ >= 0 ? OBJ : 0), NLA_FIN_WAVES>(n, mut, scratch);
                    if (tid == 0) fM_ring[q] = f;
                }
            } else if (t1 == n && OBJ >= 0) f = fM_ring[q];
        }
        if (tid == 0) status[a].fM = f;
    }
    /* the doorbell (bell != NULL: status is pinned host memory and the host is spinning on *bell instead of sleeping in a stream
     * synchronisation, whose wake-up costs more than this kernel at small n): every workgroup's thread 0 — the one that wrote its
     * part of a status record — makes it visible system-wide and counts itself; the last one resets the count and rings */
    if (bell && tid == 0) {
        __threadfence_system();
        /* acquire-release on the count: the last workgroup's read of it happens-after every other workgroup's increment — and so after
         * their status writes — and the release store of the bell below carries all of that to the host's acquire load */
        if (__hip_atomic_fetch_add(bell_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            atomicExch(bell_count, 0u);
            __hip_atomic_store(bell, bell_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

/* write accepted candidates back into the population (crs.c:153) */
__global__ __launch_bounds__(256) void crs_commit_kernel(int n, int ld, double *__restrict__ X, const double *__restrict__ TX,
                                                          const double *__restrict__ TM, const int32_t *__restrict__ slot,
                                                          const int32_t *__restrict__ kind, const int64_t *__restrict__ row, const crs_commits C,
                                                          uint32_t *__restrict__ zero, int zero_words)
{
    const int c = blockIdx.x;
    /* the control block of the window launched behind this kernel (hip/crs_chain.hip) starts from zero: cleared here by the first
     * workgroup instead of by a fill operation of its own in front of every window (two fill kernels and their gaps per window) */
    if (c == 0) for (int i = threadIdx.x; i < zero_words; i += blockDim.x) zero[i] = 0u;
    const int kc = C.inl ? C.kind[c] : kind[c], sc = C.inl ? C.slot[c] : slot[c];
    const int64_t rc = C.inl ? C.row[c] : row[c];
    const double *src = (kc == 1 ? TX : TM) + (size_t) sc * (size_t) ld;
    double *dst = X + (size_t) rc * (size_t) ld;
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

/* the same on a column-sharded population whose trial points are whole (hip/crs_chain.hip, SH instance): row `rc` of the slice X (stride
 * ld, nc columns) := columns [c0, c0 + nc) of the slot's whole point in TX / TM (stride ldf).  Workgroup 0 also clears the next window's
 * control block, and copies a whole point into `xbest` when asked (best_slot >= 0: the window's new best point, which every later trial
 * starts from and every mutation is formed around — crs.c:69,141). */
__global__ __launch_bounds__(256) void crs_commit_sh_kernel(int nc, int ld, int ldf, int c0, double *__restrict__ X, const double *__restrict__ TX,
                                                             const double *__restrict__ TM, const crs_commits C, int ncommit,
                                                             uint32_t *__restrict__ zero, int zero_words, int n, int best_slot, int best_kind,
                                                             double *__restrict__ xbest)
{
    const int c = blockIdx.x;
    if (c == 0) {
        for (int i = threadIdx.x; i < zero_words; i += blockDim.x) zero[i] = 0u;
        if (best_slot >= 0) {
            const double *src = (best_kind == 1 ? TX : TM) + (size_t) best_slot * (size_t) ldf;
            for (int i = threadIdx.x; i < n; i += blockDim.x) xbest[i] = src[i];
        }
    }
    if (c >= ncommit) return;
    const double *src = (C.kind[c] == 1 ? TX : TM) + (size_t) C.slot[c] * (size_t) ldf + (size_t) c0;
    double *dst = X + (size_t) C.row[c] * (size_t) ld;
    for (int i = threadIdx.x; i < nc; i += blockDim.x) dst[i] = src[i];
}

/* one local mutation in place (host-callback mode): p := clamp(best(1+w) - w p) */
__global__ __launch_bounds__(256) void crs_mutate_kernel(int n, const double *__restrict__ best, double *__restrict__ p,
                                                          const uint32_t *__restrict__ w, const double *__restrict__ lb,
                                                          const double *__restrict__ ub)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double wv = nla_urand_from(0., 1., w[2 * i], w[2 * i + 1]);
    p[i] = nla_clamp_box(best[i] * (1 + wv) - wv * p[i], lb[i], ub[i]);
}

/* ------------------------------------------------------------------------------------------------
 * launchers
 * ---------------------------------------------------------------------------------------------- */
extern "C" int nla_k_crs_init_rows(int obj, int n, int ld, const double *lb, const double *ub, const uint32_t *words,
                                   int64_t row_first, int64_t nrows, double *X, double *F, void *stream)
{
    if (nrows <= 0) return 0;
    const dim3 grid((unsigned) ((nrows + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t) stream;
    const double sign = nla_obj_sign(&obj);
    if (obj < 0) {
        hipLaunchKernelGGL((crs_init_rows_kernel<-1>), grid, block, 0, st, n, ld, lb, ub, words, row_first, nrows, X, F, sign);
    } else {
#define CALL(O) hipLaunchKernelGGL((crs_init_rows_kernel<O>), grid, block, 0, st, n, ld, lb, ub, words, row_first, nrows, X, F, sign)
        NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    }
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_eval(int obj, int n, int ld, const double *P, int64_t count, double *F, void *stream)
{
    if (count <= 0) return 0;
    const dim3 grid((unsigned) ((count + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t) stream;
    const double sign = nla_obj_sign(&obj);
#define CALL(O) hipLaunchKernelGGL((eval_kernel<O>), grid, block, 0, st, n, ld, P, count, F, sign)
    NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_vitter(int n, int64_t N, const uint32_t *words, int nblocks,
                                int32_t *jn, int32_t *pos, int32_t *last, void *stream)
{
    if (nblocks <= 0) return 0;
    hipLaunchKernelGGL(crs_vitter_kernel, dim3((unsigned) ((nblocks + 63) / 64)), dim3(64), 0, (hipStream_t) stream,
                       n, N, words, nblocks, jn, pos, last);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_commit(int n, int ld, double *X, const double *TX, const double *TM, int ncommit,
                                const int32_t *slot, const int32_t *kind, const int64_t *row, void *stream)
{
    if (ncommit <= 0) return 0;
    crs_commits C;
    C.inl = 0;
    hipLaunchKernelGGL(crs_commit_kernel, dim3((unsigned) ncommit), dim3(256), 0, (hipStream_t) stream,
                       n, ld, X, TX, TM, slot, kind, row, C, nullptr, 0);
    NLA_LAUNCH_CHECK();
    return 0;
}

/* nla_k_crs_commit / nla_k_crs_commit_args (lists_on_host != 0: at most 128 commits, as kernel arguments) that also clear `zero_bytes`
 * (a multiple of 4) at `zero`: the control block of the device-resolved window launched next on the same stream */
extern "C" int nla_k_crs_commit_zero(int n, int ld, double *X, const double *TX, const double *TM, int ncommit, const int32_t *slot,
                                     const int32_t *kind, const int64_t *row, int lists_on_host, void *zero, size_t zero_bytes, void *stream)
{
    if (ncommit <= 0 || (zero_bytes & 3u) || (lists_on_host && ncommit > NLA_KA_MAX)) return (int) hipErrorInvalidValue;
    crs_commits C;
    C.inl = lists_on_host ? 1 : 0;
    if (lists_on_host) {
        for (int c = 0; c < ncommit; ++c) { C.slot[c] = slot[c]; C.kind[c] = kind[c]; C.row[c] = row[c]; }
        slot = nullptr; kind = nullptr; row = nullptr;
    }
    hipLaunchKernelGGL(crs_commit_kernel, dim3((unsigned) ncommit), dim3(256), 0, (hipStream_t) stream,
                       n, ld, X, TX, TM, slot, kind, row, C, (uint32_t *) zero, (int) (zero_bytes / 4));
    NLA_LAUNCH_CHECK();
    return 0;
}

/* column-sharded windows: commits (host lists, ncommit <= 128, may be 0) from whole trial points into the slice, the control block
 * cleared, the whole best row refreshed from a slot (best_slot < 0: left as it is) — one launch in front of the window */
extern "C" int nla_k_crs_commit_sh(int nc, int ld, int ldf, int c0, double *X, const double *TX, const double *TM, int ncommit,
                                   const int32_t *h_slot, const int32_t *h_kind, const int64_t *h_row, void *zero, size_t zero_bytes,
                                   int n, int best_slot, int best_kind, double *xbest, void *stream)
{
    if (ncommit < 0 || ncommit > NLA_KA_MAX || (zero_bytes & 3u)) return (int) hipErrorInvalidValue;
    crs_commits C;
    C.inl = 1;
    for (int c = 0; c < ncommit; ++c) { C.slot[c] = h_slot[c]; C.kind[c] = h_kind[c]; C.row[c] = h_row[c]; }
    hipLaunchKernelGGL(crs_commit_sh_kernel, dim3((unsigned) (ncommit > 0 ? ncommit : 1)), dim3(256), 0, (hipStream_t) stream,
                       nc, ld, ldf, c0, X, TX, TM, C, ncommit, (uint32_t *) zero, (int) (zero_bytes / 4), n, best_slot, best_kind, xbest);
    NLA_LAUNCH_CHECK();
    return 0;
}

/* the same with the commit list given as HOST arrays (ncommit <= 128): it travels as kernel arguments, no copy to the device */
extern "C" int nla_k_crs_commit_args(int n, int ld, double *X, const double *TX, const double *TM, int ncommit,
                                     const int32_t *h_slot, const int32_t *h_kind, const int64_t *h_row, void *stream)
{
    if (ncommit <= 0) return 0;
    if (ncommit > NLA_KA_MAX) return (int) hipErrorInvalidValue;
    crs_commits C;
    C.inl = 1;
    for (int c = 0; c < ncommit; ++c) { C.slot[c] = h_slot[c]; C.kind[c] = h_kind[c]; C.row[c] = h_row[c]; }
    hipLaunchKernelGGL(crs_commit_kernel, dim3((unsigned) ncommit), dim3(256), 0, (hipStream_t) stream,
                       n, ld, X, TX, TM, nullptr, nullptr, nullptr, C, nullptr, 0);
    NLA_LAUNCH_CHECK();
    return 0;
}

extern "C" int nla_k_crs_mutate(int n, const double *best, double *p, const uint32_t *words,
                                const double *lb, const double *ub, void *stream)
{
    hipLaunchKernelGGL(crs_mutate_kernel, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, (hipStream_t) stream,
                       n, best, p, words, lb, ub);
    NLA_LAUNCH_CHECK();
    return 0;
}

/* variant: 0 = automatic; otherwise WAVES*100 + U (tuning / microbenchmarks) */
static int crs_advance_launch(int n, int ncol, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                              const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                              uint64_t first_block, int K, const int64_t *W, int nW,
                              const int32_t *t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                              double *TX, int variant, const crs_lists &L, void *stream, const crs_fwd *fwd = nullptr, double *Xw = nullptr,
                              const double *TM = nullptr);
extern "C" int nla_k_crs_advance(int n, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                                 const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                                 uint64_t first_block, int K, const int64_t *W, int nW,
                                 const int32_t *t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                                 double *TX, int variant, void *stream)
{
    crs_lists L;
    L.inl = 0;
    return crs_advance_launch(n, n, ld, X, i0, jn_ring, pos_ring, last_ring, ring_blocks, first_block, K, W, nW, t_in, t_out, slot_mask, lb, ub,
                              TX, variant, L, stream);
}
/* the same on a column slice: n rows are summed (the stream blocks' picks), ncol coordinates of every row are held (X, TX, lb, ub
 * are the slice's: ld apart, local indices) — the gather-sum of one rank of a column-sharded run (crs_shard.hip) */
extern "C" int nla_k_crs_advance_cols(int n, int ncol, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                                      const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                                      uint64_t first_block, int K, const int64_t *W, int nW,
                                      const int32_t *t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                                      double *TX, int variant, void *stream)
{
    crs_lists L;
    L.inl = 0;
    if (ncol < 1 || ncol > ld) return (int) hipErrorInvalidValue;
    return crs_advance_launch(n, ncol, ld, X, i0, jn_ring, pos_ring, last_ring, ring_blocks, first_block, K, W, nW, t_in, t_out, slot_mask, lb, ub,
                              TX, variant, L, stream);
}
/* the same with W (nW <= 128) and t_in (K <= 128) given as HOST arrays: they travel as kernel arguments */
extern "C" int nla_k_crs_advance_args(int n, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                                      const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                                      uint64_t first_block, int K, const int64_t *h_W, int nW,
                                      const int32_t *h_t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                                      double *TX, int variant, void *stream)
{
    if (K > NLA_KA_MAX || nW > NLA_KA_MAX) return (int) hipErrorInvalidValue;
    crs_lists L;
    L.inl = 1;
    for (int a = 0; a < K; ++a) L.t_in[a] = h_t_in[a];
    for (int j = 0; j < nW; ++j) L.W[j] = h_W[j];
    return crs_advance_launch(n, n, ld, X, i0, jn_ring, pos_ring, last_ring, ring_blocks, first_block, K, nullptr, nW, nullptr, t_out, slot_mask,
                              lb, ub, TX, variant, L, stream);
}
/* nla_k_crs_advance_args with the staged commits (ncommit <= 16 distinct rows; no source slot among this pass's own slots) done
 * inside the same launch and the reads of those rows forwarded: replaces nla_k_crs_commit_args + nla_k_crs_advance_args */
extern "C" int nla_k_crs_advance_commit_args(int n, int ld, double *X, int64_t i0, const int32_t *jn_ring,
                                             const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                                             uint64_t first_block, int K, const int64_t *h_W, int nW,
                                             const int32_t *h_t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                                             double *TX, const double *TM, int ncommit, const int32_t *h_slot, const int32_t *h_kind,
                                             const int64_t *h_row, int variant, void *stream)
{
    if (K > NLA_KA_MAX || nW > NLA_KA_MAX || ncommit > NLA_KC_MAX || ncommit < 0) return (int) hipErrorInvalidValue;
    crs_lists L;
    crs_fwd F;
    L.inl = 1;
    for (int a = 0; a < K; ++a) L.t_in[a] = h_t_in[a];
    for (int j = 0; j < nW; ++j) L.W[j] = h_W[j];
    F.nc = ncommit;
    for (int c = 0; c < ncommit; ++c) { F.src[c] = (h_kind[c] << 16) | (h_slot[c] & 0xffff); F.row[c] = h_row[c]; }
    return crs_advance_launch(n, n, ld, X, i0, jn_ring, pos_ring, last_ring, ring_blocks, first_block, K, nullptr, nW, nullptr, t_out, slot_mask,
                              lb, ub, TX, variant, L, stream, &F, X, TM);
}
static int crs_advance_launch(int n, int ncol, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                              const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                              uint64_t first_block, int K, const int64_t *W, int nW,
                              const int32_t *t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                              double *TX, int variant, const crs_lists &L, void *stream, const crs_fwd *fwd, double *Xw, const double *TM)
{
    if (K <= 0) return 0;
    crs_fwd F;
    F.nc = 0;
    if (fwd) F = *fwd;
    hipStream_t st = (hipStream_t) stream;
    /* variant = [1 if thin][WAVES][U as two digits]; thin = one coordinate per lane (64-coordinate chunks:
     * twice the workgroups per slot, for when few slots must spread over the whole chip) */
    bool vec2 = (ncol % 2 == 0) && (ld % 2 == 0) && ncol >= 128;
    if (variant == 0) variant = n >= 2048 ? 832 : (n >= 512 ? 416 : (n >= 128 ? 216 : 116));
    if (variant >= 10000) { vec2 = false; variant -= 10000; }
    const int cpw = vec2 ? 128 : 64;
    const int chunks = (ncol + cpw - 1) / cpw;
    const dim3 grid((unsigned) ((long) chunks * K + F.nc));
#define ADV(VEC, UU, WV) hipLaunchKernelGGL((crs_advance_kernel<VEC, UU, WV>), grid, dim3(WV * 64), 0, st, F, n, ncol, ld, X, i0, jn_ring, \
        pos_ring, last_ring, ring_blocks, first_block, K, W, nW, t_in, t_out, slot_mask, chunks, lb, ub, TX, L, Xw, TM)
    if (vec2) {
        switch (variant) {
        case 116: ADV(2, 16, 1); break;
        case 132: ADV(2, 32, 1); break;
        case 216: ADV(2, 16, 2); break;
        case 416: ADV(2, 16, 4); break;
        case 432: ADV(2, 32, 4); break;
        case 816: ADV(2, 16, 8); break;
        case 832: ADV(2, 32, 8); break;
        case 1616: ADV(2, 16, 16); break;
        default: return (int) hipErrorInvalidValue;
        }
    } else {
        switch (variant) {
        case 116: ADV(1, 16, 1); break;
        case 216: ADV(1, 16, 2); break;
        case 416: ADV(1, 16, 4); break;
        case 432: ADV(1, 32, 4); break;
        case 816: ADV(1, 16, 8); break;
        case 832: ADV(1, 32, 8); break;
        case 864: ADV(1, 64, 8); break;
        case 1632: ADV(1, 32, 16); break;
        default: ADV(1, 16, 4); break;
        }
    }
#undef ADV
    NLA_LAUNCH_CHECK();
    return 0;
}

static int crs_finish_launch(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                             const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                             const int32_t *t_in, const int32_t *t_out, int slot_mask,
                             const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                             nla_crs_slot_status *status, const crs_lists &L, void *stream, uint32_t *bell_count = nullptr,
                             uint32_t *bell = nullptr, uint32_t bell_seq = 0);
extern "C" int nla_k_crs_finish(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                                const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                                const int32_t *t_in, const int32_t *t_out, int slot_mask,
                                const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                                nla_crs_slot_status *status, void *stream)
{
    crs_lists L;
    L.inl = 0;
    return crs_finish_launch(obj, n, ld, X, i0, TX, TM, words_ring, ring_blocks, first_block, K, t_in, t_out, slot_mask, lb, ub, fT_ring, fM_ring,
                             status, L, stream);
}
/* the same with t_in (K <= 128) given as a HOST array */
extern "C" int nla_k_crs_finish_args(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                                     const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                                     const int32_t *h_t_in, const int32_t *t_out, int slot_mask,
                                     const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                                     nla_crs_slot_status *status, void *stream)
{
    if (K > NLA_KA_MAX) return (int) hipErrorInvalidValue;
    crs_lists L;
    L.inl = 1;
    for (int a = 0; a < K; ++a) L.t_in[a] = h_t_in[a];
    return crs_finish_launch(obj, n, ld, X, i0, TX, TM, words_ring, ring_blocks, first_block, K, nullptr, t_out, slot_mask, lb, ub, fT_ring,
                             fM_ring, status, L, stream);
}
/* ... and with the doorbell: `status` and `bell` are pinned host memory, `bell_count` a zeroed device word; the last workgroup of
 * the launch stores bell_seq into *bell after every status record is visible to the host (crs_engine.c spins on it) */
extern "C" int nla_k_crs_finish_args_bell(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                                          const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                                          const int32_t *h_t_in, const int32_t *t_out, int slot_mask,
                                          const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                                          nla_crs_slot_status *status, uint32_t *bell_count, uint32_t *bell, uint32_t bell_seq, void *stream)
{
    if (K > NLA_KA_MAX || !bell_count || !bell) return (int) hipErrorInvalidValue;
    crs_lists L;
    L.inl = 1;
    for (int a = 0; a < K; ++a) L.t_in[a] = h_t_in[a];
    return crs_finish_launch(obj, n, ld, X, i0, TX, TM, words_ring, ring_blocks, first_block, K, nullptr, t_out, slot_mask, lb, ub, fT_ring,
                             fM_ring, status, L, stream, bell_count, bell, bell_seq);
}
static int crs_finish_launch(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                             const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                             const int32_t *t_in, const int32_t *t_out, int slot_mask,
                             const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                             nla_crs_slot_status *status, const crs_lists &L, void *stream, uint32_t *bell_count, uint32_t *bell,
                             uint32_t bell_seq)
{
    if (K <= 0) return 0;
    const dim3 grid((unsigned) (2 * K)), block(NLA_FIN_WAVES * 64);
    hipStream_t st = (hipStream_t) stream;
    const double sign = nla_obj_sign(&obj);
    if (obj == -2) {
        hipLaunchKernelGGL((crs_finish_kernel<-2>), grid, block, 0, st, n, ld, X, i0, TX, TM, words_ring, ring_blocks,
                           first_block, K, t_in, t_out, slot_mask, lb, ub, fT_ring, fM_ring, status, L, sign, bell_count, bell, bell_seq);
    } else if (obj < 0) {
        hipLaunchKernelGGL((crs_finish_kernel<-1>), grid, block, 0, st, n, ld, X, i0, TX, TM, words_ring, ring_blocks,
                           first_block, K, t_in, t_out, slot_mask, lb, ub, fT_ring, fM_ring, status, L, sign, bell_count, bell, bell_seq);
    } else {
#define CALL(O) hipLaunchKernelGGL((crs_finish_kernel<O>), grid, block, 0, st, n, ld, X, i0, TX, TM, words_ring, ring_blocks, \
                                   first_block, K, t_in, t_out, slot_mask, lb, ub, fT_ring, fM_ring, status, L, sign, bell_count, bell, bell_seq)
        NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    }
    NLA_LAUNCH_CHECK();
    return 0;
}
