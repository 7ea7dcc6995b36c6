#define _GNU_SOURCE            /* qsort_r */
/* mlsl_driver.c — Multi-Level Single-Linkage behind the reference's entry point
 *   mlsl_minimize(n, f, f_data, lb, ub, x, minf, stop, local_opt, Nsamples, lds)   (mlsl.h:34-41),
 * host side: the point / local-minimum bookkeeping of mlsl.c:251-438 (ordered by f, the trees'
 * tie rule kept), stop tests in the reference's order, stream accounting; device side through
 * the C-ABI launchers of include/nlopt_amd.h:
 *
 *   sampling phase (mlsl.c:349-374)   the N samples of an iteration are generated from the stream
 *       and evaluated in one launch; all pair distances new x (old + new) and new x minima in one
 *       tiled kernel (bit-identical distances); closest_pt_d / closest_lm_d by masked min kernels.
 *       The reference interleaves these per sample; the values it reads later (only in the local
 *       phase) are mins over the same pair sets, so batching changes nothing.
 *   local phase (mlsl.c:380-428)      the walk over the best ceil(gamma |pts|) points is serial in the
 *       reference because a new minimum can disqualify later candidates (closest_lm_d only ever
 *       decreases).  Here the next batch of currently-potential candidates is minimised
 *       concurrently — one workgroup each in the batched L-BFGS kernel — and committed in walk
 *       order; a candidate that an earlier commit of the same batch disqualified is dropped
 *       (its start point stays un-minimised, exactly as if it had never been started).
 *   multi-GPU (nlopt_amd_set_comm)    every rank runs this same driver on the same stream (sampling and
 *       bookkeeping are replicated, they are cheap); the candidates of a batch are dealt round-robin over
 *       the ranks — candidate c is minimised by rank c mod world — and the minimisers + their results
 *       are ALL-GATHERED (SURVEY.md §8e "all-gather of local minima"), after which every rank commits
 *       the whole batch in walk order.  world x BATCH_MAX searches are in flight per batch.
 *
 * Objectives: compiled-in and user-supplied device objectives run batched as described; an ordinary host callback
 * (nlopt_func) is served with the reference's contract — f is called on the caller's thread, one point at a time, in the
 * reference's order (samples in order with the stop tests between them, mlsl.c:360-366; then one local search after the
 * other, mlsl.c:404) — while sampling, distances and every vector operation of the local searches stay on the device.
 *
 * Provided: local optimisers NLOPT_LD_LBFGS and NLOPT_LD_MMA (the GD_MLSL default); pseudo-random sampling (the non-LDS
 * variants, and the LDS variants for n > 1111 where the reference's Sobol generator is NULL — sobolseq.c:143
 * — and mlsl.c:355-359 falls back to nlopt_urand) and Sobol sampling (LDS variants, n <= 1111; points by
 * index, sobol.c; no MT words are drawn in that mode).
 */
#include "nla_internal.h"
#include "nla_switches.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define NLA_MLSL_SEG_REGENS 64          /* segment length (in regenerations of 624 words) of an MLSL run's MT19937 device stream */

#define K2PI (6.2831853071795864769252867665590057683943388)
#define MLSL_SIGMA 2.
#define MLSL_GAMMA 0.3
#define BATCH_MAX 320                  /* local searches in flight per rank (one workgroup each) */

static double gam(int n) { double z = n / 2; return sqrt(pow(K2PI * z, 1.0 / n) * z) * exp(-0.5); }   /* mlsl.c:227-237 */

typedef struct {
    int n, ld, obj, N;
    nla_evaluator ev;
    double *h_rows;                 /* host objective: the iteration's samples, pinned (N x ld) */
    void *rs;                       /* the generator's stream: st itself, or (prefetch) a second one */
    void *ev_samples;               /* recorded behind the copy of the samples' values (the distance pass is enqueued behind it) */
    int prefetch; uint64_t prefetched_at;   /* device objectives: the next iteration's sample words generated beside the local phase; stream position they were generated for */
    void *st;
    nla_mtstream *mts;
    uint64_t words_used;
    /* points: row id = insertion order */
    size_t npts, cap;
    double *F, *cpd, *cld;          /* host */
    int32_t *minimized;
    size_t *ord;                    /* row ids sorted by f, equal keys newest first (redblack.c:120) */
    int ord_has_nan;                /* a NaN key went in: from then on rows are inserted one by one (ord_insert_run) */
    size_t nlms, lcap;
    double *LF; size_t *lord;
    double *d_lb, *d_ub, *d_P, *d_F, *d_cpd, *d_LM, *d_LF, *d_D, *d_tmp;
    double *d_LX;                   /* gathered minimisers of a batch: world x per rows of ld */
    nlopt_amd_comm *comm; int world, rank;
    int32_t *d_min;
    uint32_t *d_words;
    double *d_dx;                         /* LD_MMA local optimiser: the initial step on the device, or NULL */
    uint32_t *d_V; uint32_t sobol_next;   /* LDS mode: direction table on the device, index of the next point (1-based) */
    size_t dcap;                    /* doubles in d_D */
    nla_local_ctx *lb;
    double *h_D;                    /* pinned: one row of npts distances (the rerun of a single search) */
    size_t hcap;
    double *h_S, *d_S;              /* bmax x bmax: distance between minimiser c' and start point c of a batch — all the commit walk reads of the batch's matrix */
    double *h_lfall, *d_lfall;      /* bmax: f of the accepted minima by gathered row, +inf for the others (the device-side pts_update_newlm) */
    double *lf_cand;                /* the same by CANDIDATE (what the commit walk's inner loop reads: no index arithmetic per question) */
    int64_t *h_gi, *d_gi;           /* bmax: gathered row of candidate c */
    double *h_gather; size_t gcap;  /* several ranks: world x count doubles for the element-wise min of the distance minima */
    /* per-batch lists, pinned host side + device side: [idx of all candidates: bmax][idx of this rank's: BATCH_MAX]
     * [gathered-row index of the accepted minima: bmax] as int64, then the accepted minima's f (bmax doubles); flags: bmax int32 */
    int64_t *h_idx, *d_idx;
    double *h_lf;
    int32_t *h_flags, *d_flags;
    /* THE NEXT ITERATION'S SAMPLING PHASE, AHEAD (round 5; one rank, compiled-in objective): nothing of a sampling phase but the distances to the
     * minima of the local phase before it depends on that local phase — not the sample points (the local optimisers draw no random
     * numbers), not their values, not their distances to the points and to the minima known so far.  All of it is enqueued on the
     * generator's background stream right in front of the local searches' launch (mlsl_enqueue_ahead) and runs beside it: the searches
     * are bound by memory latency on a quarter of the chip's wavefront slots, the distance pass by fp64 arithmetic.  The results wait in
     * buffers of their own (closest-point minima in d_cpd2 / h_cpd2, closest-minimum distances in h_cld2) and are merged by `min` —
     * which is what both updates are (mlsl.c:155-194) — when the iteration they belong to starts. */
    int ahead, ahead_valid;
    size_t ahead_old, ahead_nlms;    /* rows old .. old + N - 1 were sampled for a point set of `old` rows; minima 0 .. ahead_nlms - 1 are accounted for */
    uint64_t ahead_words; uint32_t ahead_sobol;
    void *ev_ahead;
    double *Fnew2, *h_cld2;          /* pinned, N each */
    double *d_D2; size_t dcap2;
    double *d_cpd2, *h_cpd2, *h_inf, *d_cld2; size_t cap2;
    /* ... and it STARTS when half of the searches it runs beside have ended (nla_k_gate on the searches' counter of finished ones): a
     * launch lasts as long as its longest search (2.2 - 6.0 ms at config 4, 3.7 on average), and whatever else the device works on
     * slows the searches by a quarter while it runs — on other compute units too (CU-masked streams: same loss), so no question of issue
     * slots or LDS — behind the gate that is the launch's thinning second half, not all of it (config 4: 9.7 -> 9.4 ms per iteration) */
    const int32_t *gate_counter; int32_t gate_from, gate_need;
    nlopt_amd_stats *stats;
    int32_t *h_gate_gave_up; int gate_off;      /* pinned word a gate sets when it times out: no more gates in this run (nlopt_amd_stats.mlsl_gate_timeouts) */
    char err[200];
} mlsl_dev;

#ifndef NLA_MLSL_GATE_PCT
#define NLA_MLSL_GATE_PCT 50        /* the sampling phase enqueued ahead starts when this share of the batch's searches have ended (0: at once) */
#endif
#define MFAIL(d, ...) do { snprintf((d)->err, sizeof (d)->err, __VA_ARGS__); return -1; } while (0)
#define MCK(d, call) do { int rc_ = (call); if (rc_) MFAIL(d, "%s failed: %s", #call, nla_dev_error_string(rc_)); } while (0)

static void mfree(mlsl_dev *d)
{
    if (d->st) nla_stream_sync(d->st);
    if (d->rs) nla_stream_sync(d->rs);
    if (d->mts) { nla_mtstream_finish(d->mts, d->words_used); nla_mtstream_destroy(d->mts); }
    nla_local_ctx_destroy(d->lb);
    free(d->F); nla_host_free(d->cpd); nla_host_free(d->cld); nla_host_free(d->minimized); free(d->ord); free(d->LF); free(d->lord);
    nla_dev_free(d->d_lb); nla_dev_free(d->d_ub); nla_dev_free(d->d_P); nla_dev_free(d->d_F); nla_dev_free(d->d_cpd);
    nla_dev_free(d->d_LM); nla_dev_free(d->d_LF); nla_dev_free(d->d_D); nla_dev_free(d->d_tmp); nla_dev_free(d->d_min);
    nla_dev_free(d->d_words); nla_dev_free(d->d_LX); nla_dev_free(d->d_V); nla_dev_free(d->d_dx);
    nla_host_free(d->h_rows);
    free(d->h_gather);
    nla_host_free(d->h_D); nla_host_free(d->h_idx); nla_host_free(d->h_lf); nla_host_free(d->h_flags);
    nla_host_free(d->h_S); nla_host_free(d->h_lfall); nla_host_free(d->h_gi); free(d->lf_cand);
    nla_dev_free(d->d_idx); nla_dev_free(d->d_flags); nla_dev_free(d->d_S); nla_dev_free(d->d_lfall); nla_dev_free(d->d_gi);
    nla_event_destroy(d->ev_samples); nla_event_destroy(d->ev_ahead);
    nla_host_free(d->h_gate_gave_up); nla_host_free(d->Fnew2); nla_host_free(d->h_cld2); nla_host_free(d->h_cpd2); nla_host_free(d->h_inf);
    nla_dev_free(d->d_D2); nla_dev_free(d->d_cpd2); nla_dev_free(d->d_cld2);
    if (d->rs && d->rs != d->st) nla_stream_destroy(d->rs);
    if (d->st) nla_stream_destroy(d->st);
}

/* host arrays that travel to / from the device every iteration live in PINNED memory: a copy from pageable memory is staged by the
 * runtime and was seen to hold the main stream's work back until the generator's stream had drained (the sample prefetch did not
 * overlap anything, profiles/r04_mlsl_prefetch_timeline.txt) */
static void *pinned_regrow(void *old, size_t old_bytes, size_t new_bytes)
{
    void *p = nla_host_malloc(new_bytes);
    if (p && old && old_bytes) memcpy(p, old, old_bytes);
    if (p) nla_host_free(old);
    return p;
}
static int grow_pts(mlsl_dev *d, size_t need)
{
    size_t ncap = d->cap ? d->cap : 1024;
    double *nP, *nF, *nC;
    int32_t *nM;
    if (need <= d->cap) return 0;
    if (!d->cap) {
        /* the first allocation holds 16 iterations' samples (within 2 GiB of rows; 524 MB at config 4 of 288 GB): growing costs a
         * round of allocations, a copy of the whole point set and frees that wait for the device — 0.9 ms in the middle of one of
         * the first iterations each time the set doubled (profiles/r05_mlsl_timeline.txt) */
        const size_t want = 16 * (size_t) d->N + 1, budget = ((size_t) 2 << 30) / (sizeof(double) * (size_t) d->ld);
        const size_t first = want < budget ? want : budget;
        if (first > ncap) ncap = first;          /* exactly the budgeted rows: rounding up to a power of two could double the 2 GiB (advisor, round 5) */
    }
    while (ncap < need) ncap *= 2;
    d->F = (double *) realloc(d->F, sizeof(double) * ncap);
    {
        double *c1 = (double *) pinned_regrow(d->cpd, sizeof(double) * d->cap, sizeof(double) * ncap);
        double *c2 = (double *) pinned_regrow(d->cld, sizeof(double) * d->cap, sizeof(double) * ncap);
        int32_t *m1 = (int32_t *) pinned_regrow(d->minimized, sizeof(int32_t) * d->cap, sizeof(int32_t) * ncap);
        if (c1) d->cpd = c1;
        if (c2) d->cld = c2;
        if (m1) d->minimized = m1;
        if (!c1 || !c2 || !m1) MFAIL(d, "out of pinned memory growing the point set");
    }
    /* (with the point set, not in the local phase when a row of distances first exceeds it: freeing pinned memory waits for the device —
     * 0.53 ms + 0.04 ms for the new block between the searches' results and the commit walk, in every iteration that crossed the old
     * size; host-side API trace of config 4, round 5) */
    nla_host_free(d->h_D);
    d->h_D = (double *) nla_host_malloc(sizeof(double) * ncap);
    d->hcap = d->h_D ? ncap : 0;
    if (!d->h_D) MFAIL(d, "out of pinned memory growing the point set");
    d->ord = (size_t *) realloc(d->ord, sizeof(size_t) * ncap);
    nP = (double *) nla_dev_malloc(sizeof(double) * ncap * (size_t) d->ld);
    nF = (double *) nla_dev_malloc(sizeof(double) * ncap);
    nC = (double *) nla_dev_malloc(sizeof(double) * ncap);
    nM = (int32_t *) nla_dev_malloc(sizeof(int32_t) * ncap);
    if (!d->F || !d->cpd || !d->cld || !d->minimized || !d->ord || !nP || !nF || !nC || !nM) {
        nla_dev_free(nP); nla_dev_free(nF); nla_dev_free(nC); nla_dev_free(nM);
        MFAIL(d, "out of memory growing the point set");
    }
    if (d->npts && (nla_memcpy_d2d(nP, d->d_P, sizeof(double) * d->npts * (size_t) d->ld, d->st) ||
                    nla_memcpy_d2d(nF, d->d_F, sizeof(double) * d->npts, d->st) || nla_stream_sync(d->st))) {
        nla_dev_free(nP); nla_dev_free(nF); nla_dev_free(nC); nla_dev_free(nM);
        MFAIL(d, "copying the point set failed");
    }
    nla_dev_free(d->d_P); nla_dev_free(d->d_F); nla_dev_free(d->d_cpd); nla_dev_free(d->d_min);
    d->d_P = nP; d->d_F = nF; d->d_cpd = nC; d->d_min = nM;
    d->cap = ncap;
    return 0;
}

static int grow_lms(mlsl_dev *d, size_t need)
{
    size_t ncap = d->lcap ? d->lcap : 256;
    double *nL, *nF;
    if (need <= d->lcap) return 0;
    if (!d->lcap) {           /* (as the point set: room for 8 iterations' worth of minima from the start, within 1 GiB) */
        const size_t want = 8 * (size_t) d->N, budget = ((size_t) 1 << 30) / (sizeof(double) * (size_t) d->ld);
        const size_t first = want < budget ? want : budget;
        if (first > ncap) ncap = first;
    }
    while (ncap < need) ncap *= 2;
    d->LF = (double *) realloc(d->LF, sizeof(double) * ncap);
    d->lord = (size_t *) realloc(d->lord, sizeof(size_t) * ncap);
    nL = (double *) nla_dev_malloc(sizeof(double) * ncap * (size_t) d->ld);
    nF = (double *) nla_dev_malloc(sizeof(double) * ncap);
    if (!d->LF || !d->lord || !nL || !nF) { nla_dev_free(nL); nla_dev_free(nF); MFAIL(d, "out of memory growing the local-minimum set"); }
    if (d->nlms && (nla_memcpy_d2d(nL, d->d_LM, sizeof(double) * d->nlms * (size_t) d->ld, d->st) ||
                    nla_memcpy_d2d(nF, d->d_LF, sizeof(double) * d->nlms, d->st) || nla_stream_sync(d->st))) {
        nla_dev_free(nL); nla_dev_free(nF);
        MFAIL(d, "copying the local-minimum set failed");
    }
    nla_dev_free(d->d_LM); nla_dev_free(d->d_LF);
    d->d_LM = nL; d->d_LF = nF; d->lcap = ncap;
    return 0;
}

/* The distance scratch grows with the point set — N x npts doubles per sampling phase, npts larger by N every iteration — and it used
 * to be reallocated to the exact size EVERY iteration: a hipFree (which waits for every stream of the device: the generator's, when it
 * works ahead on a stream of its own — why round 4's sample prefetch never overlapped anything, profiles/r04_mlsl_prefetch_timeline.txt) and
 * a hipMalloc of tens of MB between the sampling kernel and the distance pass.  Now it doubles: a handful of reallocations per run. */
static int need_D(mlsl_dev *d, size_t doubles)
{
    if (doubles <= d->dcap) return 0;
    if (doubles < 2 * d->dcap) doubles = 2 * d->dcap;
    if (!d->dcap) {            /* the first allocation: the pass of the 8th iteration (N x 8 N doubles, within 1 GiB) */
        const size_t want = 8 * (size_t) d->N * (size_t) d->N, budget = ((size_t) 1 << 30) / sizeof(double);
        if (doubles < (want < budget ? want : budget)) doubles = want < budget ? want : budget;
    }
    nla_dev_free(d->d_D);
    d->d_D = (double *) nla_dev_malloc(sizeof(double) * doubles);
    d->dcap = d->d_D ? doubles : 0;
    if (!d->d_D) MFAIL(d, "out of device memory (distance matrix)");
    return 0;
}

/* ---- the next iteration's sampling phase, ahead (see mlsl_dev) ---- */
static int ahead_buffers(mlsl_dev *d, size_t doubles)
{
    if (d->cap2 < d->cap) {
        size_t i;
        nla_host_free(d->h_cpd2); nla_host_free(d->h_inf); nla_dev_free(d->d_cpd2);
        d->h_cpd2 = (double *) nla_host_malloc(sizeof(double) * d->cap);
        d->h_inf = (double *) nla_host_malloc(sizeof(double) * d->cap);
        d->d_cpd2 = (double *) nla_dev_malloc(sizeof(double) * d->cap);
        d->cap2 = (d->h_cpd2 && d->h_inf && d->d_cpd2) ? d->cap : 0;
        if (!d->cap2) MFAIL(d, "out of memory (sampling ahead)");
        for (i = 0; i < d->cap; ++i) d->h_inf[i] = HUGE_VAL;
    }
    if (!d->Fnew2) d->Fnew2 = (double *) nla_host_malloc(sizeof(double) * (size_t) d->N);
    if (!d->h_cld2) d->h_cld2 = (double *) nla_host_malloc(sizeof(double) * (size_t) d->N);
    if (!d->d_cld2) d->d_cld2 = (double *) nla_dev_malloc(sizeof(double) * (size_t) d->N);
    if (!d->Fnew2 || !d->h_cld2 || !d->d_cld2) MFAIL(d, "out of memory (sampling ahead)");
    if (doubles > d->dcap2) {
        if (doubles < 2 * d->dcap2) doubles = 2 * d->dcap2;
        if (!d->dcap2) {       /* as need_D: the pass of the 8th iteration from the start, within 1 GiB */
            const size_t want = 8 * (size_t) d->N * (size_t) d->N, budget = ((size_t) 1 << 30) / sizeof(double);
            if (doubles < (want < budget ? want : budget)) doubles = want < budget ? want : budget;
        }
        nla_dev_free(d->d_D2);
        d->d_D2 = (double *) nla_dev_malloc(sizeof(double) * doubles);
        d->dcap2 = d->d_D2 ? doubles : 0;
        if (!d->d_D2) MFAIL(d, "out of device memory (distance matrix of the samples ahead)");
    }
    return 0;
}

/* enqueue, on the generator's stream, the sampling of the iteration AFTER the current one: rows npts .. npts + N - 1 (the current
 * iteration's sampling is complete, its local phase adds no points), their values, closest_pt_d of and through them (find_closest_pt,
 * pts_update_newpt: mlsl.c:155-178) and their closest_lm_d over the minima known now (find_closest_lm, :139-153).  Called in front of
 * the local searches' launch, behind the generator's fill of the words these rows are made of.  Every allocation it may need is
 * made here, before anything is enqueued (freeing device memory waits for the whole device). */
static int mlsl_enqueue_ahead(mlsl_dev *d, int n)
{
    const size_t old = d->npts, nlms = d->nlms;
    const int N = d->N, nb = (int) (old + (size_t) N);
    const size_t cols = (size_t) nb > nlms ? (size_t) nb : nlms;
    const double *A, *FA;
    d->ahead_valid = 0;
    if (d->d_V && (uint64_t) d->sobol_next + (uint64_t) N >= 4294967295ULL) return 0;     /* (the iteration itself reports the exhausted sequence) */
    if ((size_t) N * cols > ((size_t) 1 << 29)) { d->ahead = 0; return 0; }               /* a second distance matrix above 4 GiB: not worth the memory */
    if (grow_pts(d, old + (size_t) N) || ahead_buffers(d, (size_t) N * cols)) return -1;
    A = d->d_P + old * (size_t) d->ld; FA = d->d_F + old;
    if (d->h_gate_gave_up && *(volatile int32_t *) d->h_gate_gave_up) {     /* (the launch it belonged to has long ended) */
        *(volatile int32_t *) d->h_gate_gave_up = 0; d->gate_off = 1;
        if (d->stats) ++d->stats->mlsl_gate_timeouts;
    }
    if (d->gate_need > 0 && !d->gate_off && nla_k_gate(d->gate_counter, d->gate_from, d->gate_need, 50.0, d->h_gate_gave_up, d->rs)) MFAIL(d, "sampling ahead failed");
    if (d->d_V) {
        if (nla_k_mlsl_sobol_rows(n, d->ld, d->d_lb, d->d_ub, d->d_V, d->sobol_next, N, d->d_P + old * (size_t) d->ld, d->rs) ||
            nla_k_eval(d->obj, n, d->ld, A, N, d->d_F + old, d->rs) ||
            (d->ev.sign < 0 && nla_k_mlsl_negate(d->d_F + old, N, d->rs))) MFAIL(d, "sampling ahead failed");
    } else if (nla_k_crs_init_rows(d->obj, n, d->ld, d->d_lb, d->d_ub, d->d_words, (int64_t) old, N, d->d_P, d->d_F, d->rs) ||
               (d->ev.sign < 0 && nla_k_mlsl_negate(d->d_F + old, N, d->rs))) MFAIL(d, "sampling ahead failed");
    if (nla_memcpy_d2h(d->Fnew2, d->d_F + old, sizeof(double) * (size_t) N, d->rs) ||
        nla_memcpy_h2d(d->d_cpd2, d->h_inf, sizeof(double) * (size_t) nb, d->rs) ||
        nla_k_mlsl_dist2(n, d->ld, A, N, d->d_P, nb, d->d_D2, d->rs) ||
        nla_k_mlsl_rowmin(d->d_D2, nb, N, nb, FA, d->d_F, NULL, d->d_cpd2 + old, d->rs) ||
        nla_k_mlsl_colmin(d->d_D2, nb, N, (int) old, FA, d->d_F, NULL, d->d_cpd2, d->rs) ||
        nla_memcpy_d2h(d->h_cpd2, d->d_cpd2, sizeof(double) * (size_t) nb, d->rs)) MFAIL(d, "distance pass ahead failed");
    if (nlms && (nla_k_mlsl_dist2(n, d->ld, A, N, d->d_LM, (int) nlms, d->d_D2, d->rs) ||
                 nla_k_mlsl_rowmin(d->d_D2, (int) nlms, N, (int) nlms, FA, d->d_LF, NULL, d->d_cld2, d->rs) ||
                 nla_memcpy_d2h(d->h_cld2, d->d_cld2, sizeof(double) * (size_t) N, d->rs))) MFAIL(d, "distance pass ahead failed");
    if (nla_event_record(d->ev_ahead, d->rs)) MFAIL(d, "distance pass ahead failed");
    d->ahead_old = old; d->ahead_nlms = nlms; d->ahead_words = d->words_used; d->ahead_sobol = d->sobol_next;
    d->ahead_valid = 1;
    return 0;
}

/* insert row id `r` (value f) into an order array: before the first element that is not smaller */
static void ord_insert(size_t *ord, size_t cnt, const double *F, size_t r)
{
    size_t lo = 0, hi = cnt;
    const double f = F[r];
    while (lo < hi) { size_t mid = (lo + hi) / 2; if (F[ord[mid]] < f) lo = mid + 1; else hi = mid; }
    memmove(ord + lo + 1, ord + lo, (cnt - lo) * sizeof *ord);
    ord[lo] = r;
}

/* the same for a run of nnew consecutive rows first .. first + nnew - 1 inserted one after the other (the sampling phase's 1000 points
 * per iteration: a binary search through F and a memmove of half the array each — 1 ms per iteration at 3000 points, growing with the
 * point set): sort the new rows, merge from the back.  The order is the one-by-one insertion's: an equal key goes in front of those
 * already there, so among equals new rows come before old ones and later new rows before earlier ones.  (Keys that do not order —
 * NaN — take the one-by-one path: the binary search's answer for them is whatever it happens to be, and that is what counts.) */
static int ord_cmp_new(const void *a_, const void *b_, void *F_)
{
    const double *F = (const double *) F_;
    const size_t a = *(const size_t *) a_, b = *(const size_t *) b_;
    if (F[a] < F[b]) return -1;
    if (F[a] > F[b]) return 1;
    return a > b ? -1 : (a < b ? 1 : 0);                 /* the later row first */
}
static void ord_insert_run(size_t *ord, size_t cnt, const double *F, size_t first, size_t nnew, int any_nan)
{
    size_t k;
    long i, j;
    for (k = 0; k < nnew && !any_nan; ++k) if (F[first + k] != F[first + k]) any_nan = 1;
    if (any_nan || nnew < 8) { for (k = 0; k < nnew; ++k) ord_insert(ord, cnt + k, F, first + k); return; }
    for (k = 0; k < nnew; ++k) ord[cnt + k] = first + k;                      /* the tail of the array is the merge's scratch */
    qsort_r(ord + cnt, nnew, sizeof *ord, ord_cmp_new, (void *) F);
    {
        /* merge in place from the back; the sorted new rows are first moved out of the way of the write position */
        size_t *tmp = (size_t *) malloc(sizeof *tmp * nnew);
        if (!tmp) { for (k = 0; k < nnew; ++k) ord[cnt + k] = 0; for (k = 0; k < nnew; ++k) ord_insert(ord, cnt + k, F, first + k); return; }
        memcpy(tmp, ord + cnt, sizeof *tmp * nnew);
        i = (long) cnt - 1; j = (long) nnew - 1;
        for (k = cnt + nnew; k-- > 0 && j >= 0; ) {
            if (i >= 0 && !(F[ord[i]] < F[tmp[j]])) ord[k] = ord[i--];       /* an old row with a key not smaller stays behind the new one */
            else ord[k] = tmp[j--];
        }
        free(tmp);
    }
}

/* the distances of a batch's minimisers to every point + the nb x nb of them the commit walk reads, enqueued on the stream (MLSL's
 * local phase; with a device objective on one rank: right behind the searches' kernel, nla_local_ctx_after_launch) */
typedef struct { mlsl_dev *d; int n, nb; size_t na; const double *ctx_X; size_t lx_bytes; } mlsl_pairs_job;
static int mlsl_enqueue_pairs(void *arg)
{
    mlsl_pairs_job *j = (mlsl_pairs_job *) arg;
    mlsl_dev *d = j->d;
    if (j->ctx_X && j->lx_bytes && nla_memcpy_d2d(d->d_LX, j->ctx_X, j->lx_bytes, d->st)) return -1;      /* (one rank: the all-gather of the minimisers is this copy) */
    return nla_memcpy_h2d(d->d_gi, d->h_gi, sizeof(int64_t) * (size_t) j->nb, d->st) ||
           nla_k_mlsl_dist2(j->n, d->ld, d->d_LX, (int) j->na, d->d_P, (int) d->npts, d->d_D, d->st) ||
           nla_k_mlsl_gather_pairs_t(d->d_D, (int) d->npts, d->d_gi, j->nb, d->d_idx, j->nb, d->d_S, d->st) ||
           nla_memcpy_d2h(d->h_S, d->d_S, sizeof(double) * (size_t) j->nb * (size_t) j->nb, d->st);
}

/* v[k] = min over the ranks of their v[k] (each rank holds the minima over ITS rows of the distance matrix) */
static int min_over_ranks(mlsl_dev *d, double *v, size_t count)
{
    size_t k;
    int r;
    if (count == 0) return 0;
    if ((size_t) d->world * count > d->gcap) {
        free(d->h_gather);
        d->gcap = 2 * (size_t) d->world * count;
        d->h_gather = (double *) malloc(sizeof(double) * d->gcap);
        if (!d->h_gather) { d->gcap = 0; return -1; }
    }
    if (nla_comm_allgather_host(d->comm, v, d->h_gather, sizeof(double) * count, d->st)) return -1;
    for (k = 0; k < count; ++k) {
        double m = d->h_gather[k];
        for (r = 1; r < d->world; ++r) { const double t = d->h_gather[(size_t) r * count + k]; if (t < m) m = t; }
        v[k] = m;
    }
    return 0;
}

/* the objective as MLSL's local optimiser sees it (fcount, mlsl.c:246-251): every call counted where the caller can see it */
typedef struct { nlopt_func f; void *f_data; double sign; int *nevals_p; } mlsl_count_wrap;
static double mlsl_counted_f(unsigned n, const double *x, double *grad, void *p)
{
    mlsl_count_wrap *w = (mlsl_count_wrap *) p;
    ++*w->nevals_p;
    return w->f(n, x, grad, w->f_data);
}
/* -f for a maximisation whose objective the dispatcher left unflipped (dev_sign) but which runs through host calls here */
static double mlsl_signed_f(unsigned n, const double *x, double *grad, void *p)
{
    mlsl_count_wrap *w = (mlsl_count_wrap *) p;
    (void) grad;
    return w->sign * w->f(n, x, NULL, w->f_data);
}

nlopt_result nla_mlsl_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x,
                               double *minf, nla_stopping *stop, nlopt_opt local_opt, int Nsamples, int lds)
{
    mlsl_dev D;
    nlopt_result ret = NLOPT_SUCCESS;
    nlopt_amd_stats *st = opt ? &opt->stats : NULL;
    nla_lbfgs_params prm;
    nla_lbfgs_result *res = NULL, *res_mine = NULL;
    size_t *cand = NULL;
    int bmax;
    double R_prefactor, *Fnew = NULL, best_f = HUGE_VAL;
    const double dlm = 1.0, dbound = 1e-6;
    const double *lbh = lb, *ubh = ub;
    int i, mf, best_is_lm = 0, loc_maxeval, use_mma = 0, use_cobyla = 0, cob_dev = 0, host, batch;
    mlsl_count_wrap cw, sw;
    nlopt_func lo_f = NULL; void *lo_fdata = NULL;
    nla_mma_params mma;
    nla_stopping lstop, agreed_view;
    const nla_stopping *sp = stop;
    int agreed_force = 0;
    size_t best_row = 0;
    double *cob_x = NULL;

    memset(&D, 0, sizeof D);
    D.N = Nsamples ? Nsamples : 4;                                             /* mlsl.c:283-286 */
    if (D.N < 1) { nla_stop_msg(stop, "population %d is too small", D.N); return NLOPT_INVALID_ARGS; }
    if (!local_opt || (local_opt->algorithm != NLOPT_LD_LBFGS && local_opt->algorithm != NLOPT_LD_MMA && local_opt->algorithm != NLOPT_LN_COBYLA)) {
        nla_stop_msg(stop, "nlopt_amd: MLSL is provided with NLOPT_LD_LBFGS, NLOPT_LD_MMA or NLOPT_LN_COBYLA as the local optimizer only (not %s)",
                     local_opt ? nlopt_algorithm_name(local_opt->algorithm) : "none");
        return NLOPT_INVALID_ARGS;
    }
    use_mma = local_opt->algorithm == NLOPT_LD_MMA;
    /* LN_COBYLA (GN_MLSL's default, optimize.c:763-768).  With a compiled-in device objective (round 6) the searches of a batch run
     * CONCURRENTLY on the device, one wavefront per start with the search's whole state in LDS (hip/cobyla_kernels.hip), like
     * LD_LBFGS / LD_MMA: `cob_dev`.  Otherwise — a host callback, a user kernel, a dimension whose simplex, inverse, models and LP
     * basis no longer fit a compute unit's LDS (n > 51), or "amd_cobyla_host" = 1 — it is a HOST algorithm (cobyla_host.c): the
     * searches run one at a time on the caller's thread through the library's own nlopt_optimize, exactly as mlsl.c:404-407 runs them;
     * samples, distances and the bookkeeping stay on the device, and the run takes the host-callback path throughout. */
    use_cobyla = local_opt->algorithm == NLOPT_LN_COBYLA;
    if (use_mma && (i = nla_mma_read_params(local_opt, &mma))) {
        nla_stop_msg(stop, "%s", local_opt->errmsg ? local_opt->errmsg : "invalid LD_MMA parameter");
        return (nlopt_result) i;
    }
    if (nla_dev_count() <= 0) {
        nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)");
        nla_comm_agree_ready((opt && !use_cobyla) ? opt->comm : NULL, 0);
        return NLOPT_FAILURE;
    }
    nla_evaluator_resolve(&D.ev, opt, f, f_data);
    D.obj = D.ev.kind == NLA_EVAL_DEVICE ? D.ev.obj : -1;
    cob_dev = use_cobyla && D.ev.kind == NLA_EVAL_DEVICE && nla_cobyla_fits(n) && !(opt && nlopt_get_param(opt, "amd_cobyla_host", 0) != 0) &&
              !nlopt_get_param(local_opt, "amd_cobyla_host", 0);
    if (cob_dev) use_cobyla = 0;              /* from here on `use_cobyla` means: the host algorithm */
    host = D.ev.kind == NLA_EVAL_HOST || use_cobyla;
    sw.f = f; sw.f_data = f_data; sw.sign = -1.; sw.nevals_p = NULL;
    if (use_cobyla && D.ev.kind != NLA_EVAL_HOST && D.ev.sign < 0) { f = mlsl_signed_f; f_data = &sw; }   /* a maximisation the dispatcher left unflipped (dev_sign): flip here */
    D.n = n; D.ld = (n + 1) & ~1;
    D.comm = (opt && !use_cobyla) ? opt->comm : NULL;              /* host searches: every rank runs the identical job, nothing to exchange */
    D.world = nlopt_amd_comm_world(D.comm); D.rank = nlopt_amd_comm_rank(D.comm);
    batch = host ? 1 : BATCH_MAX;             /* a host callback is called for one search at a time, in the reference's order */
    bmax = (host ? 1 : BATCH_MAX) * D.world;
    /* what a local search observes of the caller's stopping state: the force_stop flag and the run's clock
     * (nlopt_optimize_limited hands the remaining time down, optimize.c:1101-1104) */
    lstop = *stop;
    lstop.xtol_abs = local_opt->xtol_abs; lstop.x_weights = local_opt->x_weights;
    R_prefactor = sqrt(2. / K2PI) * pow(gam(n) * MLSL_SIGMA, 1.0 / n);            /* mlsl.c:313-317 */
    for (i = 0; i < n; ++i) R_prefactor *= pow(ub[i] - lb[i], 1.0 / n);

    /* the local optimiser as nlopt_optimize_limited(local_opt, ...) would configure it (mlsl.c:303-306,404-407) */
    memset(&prm, 0, sizeof prm);
    prm.minf_max = stop->minf_max; prm.ftol_rel = local_opt->ftol_rel; prm.ftol_abs = local_opt->ftol_abs;
    prm.xtol_rel = local_opt->xtol_rel; prm.tolg = nlopt_get_param(local_opt, "tolg", 0.);
    loc_maxeval = local_opt->maxeval;
    mf = nla_lbfgs_default_mf(n, (int) local_opt->vector_storage, 0);

    D.st = nla_stream_create();
    /* several ranks: set-up ends with an exchange of "ready" (comm.c, nla_comm_agree_ready) — a rank that fails below says so there
     * instead of leaving the others in the run's first collective */
    /* PREFETCH (always, with a device objective; measured on the MI355X in round 5 together with the short segments below: config 4
     * 15.9 -> 13.9 ms per iteration, profiles/r05_staged_ab.txt): pseudo-random sampling (every MLSL
     * variant at n > 1111, where the reference has no Sobol generator either) costs a generator pass per iteration that is bound by
     * latency, not throughput — 2 n N words are 13 segments = 13 wavefronts at config 4, 1.9 ms + 0.5-1 ms of segment jumps of the
     * 7.5 ms sampling phase.  Nothing between two sampling phases draws random numbers (the local optimisers are deterministic), so
     * the NEXT iteration's words are generated on a stream of their own beside the distance pass and the local searches.  Hand-over
     * = a host synchronisation of that stream before the sampling kernel reads them. */
    D.prefetch = !host;
    /* (not with the sums in the reference's order, amd_exact_dot = 1: that launch is one chain of dependent fp64 additions per search,
     * every issue slot the distance pass takes on its SIMD delays the chain — measured: 103 -> 117 ms per launch, round 5) */
    D.ahead = !host && D.world == 1 && D.ev.kind == NLA_EVAL_DEVICE && !use_cobyla && !cob_dev && !nla_exact_mode_for(opt, local_opt, &D.ev);
    D.prefetched_at = ~0ULL;
    D.stats = st;
    if (D.ahead) { D.h_gate_gave_up = (int32_t *) nla_host_malloc(sizeof(int32_t)); if (D.h_gate_gave_up) *D.h_gate_gave_up = 0; }     /* (its failure is the set-up's, below) */
    D.rs = (D.st && D.prefetch) ? nla_stream_create_background() : D.st;
    D.ev_samples = nla_event_create();
    D.ev_ahead = nla_event_create();
    {
        /* "amd_mlsl_seg_regens": the segment length of this run's stream (NLA_MT_SEG_REGENS = 1024 is the layout every other algorithm
         * uses) — 64 makes 205 wavefronts of the 13 above; measured at config 4 (MI355X, round 5): 15.9 ms per iteration at 1024, 14.7 at
         * 256, 14.4 at 64, 14.3 at 16.  The words are the same. */
        int seg = opt ? (int) nlopt_get_param(opt, "amd_mlsl_seg_regens", NLA_MLSL_SEG_REGENS) : NLA_MLSL_SEG_REGENS;
        if (seg < 1 || seg > NLA_MT_SEG_REGENS || (seg & (seg - 1))) seg = NLA_MT_SEG_REGENS;
        if (D.st && D.rs) D.mts = seg == NLA_MT_SEG_REGENS ? nla_mtstream_create(D.rs) : nla_mtstream_create_seg(D.rs, seg);
    }
    if (!D.st || !D.rs || !D.mts) { nla_stop_msg(stop, "nlopt_amd: could not create the device stream / generator state"); nla_comm_agree_ready(D.comm, 0); mfree(&D); return NLOPT_OUT_OF_MEMORY; }
    D.d_lb = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.ld);
    D.d_ub = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.ld);
    D.d_words = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * 2 * (size_t) n * (size_t) D.N);
    D.d_tmp = (double *) nla_dev_malloc(sizeof(double) * (size_t) (D.N > bmax ? D.N : bmax));
    D.d_LX = (double *) nla_dev_malloc(sizeof(double) * (size_t) bmax * (size_t) D.ld);
    D.h_idx = (int64_t *) nla_host_malloc(sizeof(int64_t) * (size_t) (2 * bmax + BATCH_MAX));
    D.d_idx = (int64_t *) nla_dev_malloc(sizeof(int64_t) * (size_t) (2 * bmax + BATCH_MAX));
    D.h_lf = (double *) nla_host_malloc(sizeof(double) * (size_t) bmax);
    D.h_S = (double *) nla_host_malloc(sizeof(double) * (size_t) bmax * (size_t) bmax);
    D.d_S = (double *) nla_dev_malloc(sizeof(double) * (size_t) bmax * (size_t) bmax);
    D.h_lfall = (double *) nla_host_malloc(sizeof(double) * (size_t) bmax);
    D.lf_cand = (double *) malloc(sizeof(double) * (size_t) bmax);
    D.d_lfall = (double *) nla_dev_malloc(sizeof(double) * (size_t) bmax);
    D.h_gi = (int64_t *) nla_host_malloc(sizeof(int64_t) * (size_t) bmax);
    D.d_gi = (int64_t *) nla_dev_malloc(sizeof(int64_t) * (size_t) bmax);
    D.h_flags = (int32_t *) nla_host_malloc(sizeof(int32_t) * (size_t) bmax);
    D.d_flags = (int32_t *) nla_dev_malloc(sizeof(int32_t) * (size_t) bmax);
    Fnew = (double *) nla_host_malloc(sizeof(double) * (size_t) D.N);        /* pinned: the samples' values come back every iteration */
    if (host) D.h_rows = (double *) nla_host_malloc(sizeof(double) * (size_t) D.N * (size_t) D.ld);
    res = (nla_lbfgs_result *) malloc(sizeof *res * (size_t) bmax);
    res_mine = (nla_lbfgs_result *) calloc(BATCH_MAX, sizeof *res_mine);
    cand = (size_t *) malloc(sizeof *cand * (size_t) bmax);
    if ((host && !D.h_rows) || !D.h_idx || !D.d_idx || !D.h_lf || !D.h_S || !D.d_S || !D.h_lfall || !D.lf_cand || !D.d_lfall || !D.h_gi || !D.d_gi || !D.h_flags || !D.d_flags || !D.d_lb || !D.d_ub || !D.d_words || !D.d_tmp || !D.d_LX || !D.ev_samples || !D.ev_ahead || (D.ahead && !D.h_gate_gave_up) || !Fnew || !res || !res_mine || !cand || grow_pts(&D, (size_t) D.N + 1) || grow_lms(&D, 1) ||
        nla_memcpy_h2d(D.d_lb, lbh, sizeof(double) * (size_t) n, D.st) || nla_memcpy_h2d(D.d_ub, ubh, sizeof(double) * (size_t) n, D.st)) {
        nla_stop_msg(stop, "nlopt_amd: could not create the MLSL device state");
        nla_comm_agree_ready(D.comm, 0);
        mfree(&D); nla_host_free(Fnew); free(res); free(res_mine); free(cand);
        return NLOPT_OUT_OF_MEMORY;
    }
    if (lds) {                                                                 /* d.s = nlopt_sobol_create(n), mlsl.c:306 */
        uint32_t *V = (uint32_t *) malloc(sizeof(uint32_t) * 32 * (size_t) n);
        if (V && nla_sobol_directions((unsigned) n, V)) {
            D.d_V = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * 32 * (size_t) n);
            if (!D.d_V || nla_memcpy_h2d(D.d_V, V, sizeof(uint32_t) * 32 * (size_t) n, D.st) || nla_stream_sync(D.st)) {
                nla_stop_msg(stop, "nlopt_amd: could not create the MLSL device state");
                nla_comm_agree_ready(D.comm, 0);
                free(V); mfree(&D); nla_host_free(Fnew); free(res); free(res_mine); free(cand);
                return NLOPT_OUT_OF_MEMORY;
            }
            D.sobol_next = nla_sobol_skip_count((unsigned) (10 * n + D.N)) + 1;  /* nlopt_sobol_skip(d.s, 10n+N, .), mlsl.c:332 */
        }                                                                      /* else: NULL generator -> nlopt_urand, as the reference */
        free(V);
    }
    if ((use_mma || cob_dev) && local_opt->dx) {                                            /* the initial step = MMA's sigma_init, optimize.c:829 */
        D.d_dx = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.ld);
        if (!D.d_dx || nla_memcpy_h2d(D.d_dx, local_opt->dx, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) {
            nla_stop_msg(stop, "nlopt_amd: could not create the MLSL device state");
            nla_comm_agree_ready(D.comm, 0);
            mfree(&D); nla_host_free(Fnew); free(res); free(res_mine); free(cand);
            return NLOPT_OUT_OF_MEMORY;
        }
    }
    if (use_cobyla) {
        /* the local optimiser as mlsl.c:303-306 configures it: counted objective, the box, the stop value */
        lo_f = local_opt->f; lo_fdata = local_opt->f_data;
        cw.f = f; cw.f_data = f_data; cw.sign = 1.; cw.nevals_p = stop->nevals_p;      /* (f is already the flipped one where that applies) */
        /* installed by assignment and taken out again at every exit below (opt->local_opt is the object's own copy whose objective
         * nlopt_set_local_optimizer cleared, options.c:824-846: nothing of the user's to munge, but nothing of this stack frame
         * may stay behind in it either) */
        local_opt->f = mlsl_counted_f; local_opt->f_data = &cw; local_opt->pre = NULL; local_opt->maximize = 0;
        nlopt_set_lower_bounds(local_opt, lb);
        nlopt_set_upper_bounds(local_opt, ub);
        nlopt_set_stopval(local_opt, stop->minf_max);
        cob_x = (double *) nla_host_malloc(sizeof(double) * (size_t) n);
        if (!cob_x) { nla_stop_msg(stop, "nlopt_amd: out of pinned memory"); ret = NLOPT_OUT_OF_MEMORY; goto done; }
    } else {
    D.lb = cob_dev ? nla_local_ctx_create_cobyla(&D.ev, n, batch, D.d_dx, D.d_lb, D.d_ub, D.st)
         : use_mma ? nla_local_ctx_create_mma(&D.ev, n, batch, &mma, D.d_dx, D.d_lb, D.d_ub, D.st)
                   : nla_local_ctx_create(&D.ev, n, batch, mf, D.d_lb, D.d_ub, D.st);
    if (D.lb && nla_local_ctx_set_options(D.lb, nla_exact_mode_for(opt, local_opt, &D.ev), local_opt->xtol_abs, local_opt->x_weights)) { nla_local_ctx_destroy(D.lb); D.lb = NULL; }
    /* (LD_LBFGS only: LD_MMA's launches are half as long as the distance pass beside them — behind a gate it ends after them and holds the
     * next iteration up: config 4 with LD_MMA 8.0 -> 8.3 ms per iteration, profiles/r05_mlsl_ahead_ab.txt) */
    if (D.lb && D.ahead && !use_mma && NLA_MLSL_GATE_PCT > 0 && nla_local_ctx_count_finished(D.lb)) { nla_local_ctx_destroy(D.lb); D.lb = NULL; }
    if (!D.lb) { nla_stop_msg(stop, "nlopt_amd: out of device memory (local-search batch)"); nla_comm_agree_ready(D.comm, 0); mfree(&D); nla_host_free(Fnew); free(res); free(res_mine); free(cand); return NLOPT_OUT_OF_MEMORY; }
    nla_local_ctx_set_stats(D.lb, st);
    if (cob_dev) nla_local_ctx_set_cobyla_min_batch(D.lb, (int) nlopt_get_param(opt ? opt : local_opt, "amd_cobyla_min_batch", nlopt_get_param(local_opt, "amd_cobyla_min_batch", 0)));
    }
    if (D.world > 1) {
        const int all = nla_comm_agree_same(D.comm, 1, nla_problem_fingerprint(lds ? NLOPT_G_MLSL_LDS : NLOPT_G_MLSL, n, D.N, D.obj + 100 * (int) local_opt->algorithm, lb, ub, x, stop) + nla_params_fingerprint(opt) + nla_params_fingerprint(local_opt));
        if (all < 0) { nla_stop_msg(stop, NLA_MSG_RANKS_DIFFER); ret = NLOPT_INVALID_ARGS; goto done; }
        if (all == 0) { nla_stop_msg(stop, "nlopt_amd: another rank could not set up its MLSL device state"); ret = NLOPT_FAILURE; goto done; }
    }
#define DEVFAIL() do { nla_stop_msg(stop, "device engine: %s", D.err); ret = NLOPT_FAILURE; goto done; } while (0)
#define PREFETCH_NOW() do { if (prefetch_due) { prefetch_due = 0; \
        if (nla_mtstream_fill(D.mts, D.words_used, 2ULL * (uint64_t) n * (uint64_t) D.N, D.d_words)) { snprintf(D.err, sizeof D.err, "MT stream fill failed"); DEVFAIL(); } \
        D.prefetched_at = D.words_used; } \
    if (ahead_due) { ahead_due = 0; if (mlsl_enqueue_ahead(&D, n)) DEVFAIL(); } } while (0)
/* the sampling phase's distance pass (find_closest_pt + pts_update_newpt, find_closest_lm) for the N new points in rows old .. old + N - 1:
 * everything it needs is on the device once the sampling kernel has run — the points, their f, the old points' closest_pt_d and flags
 * (uploaded here: the new rows' entries are set before) — so with a device objective on one rank it is ENQUEUED RIGHT BEHIND THE
 * SAMPLING KERNEL and runs while the host walks the N values (counts, stop tests, the order array: 0.3 ms per iteration that used to
 * stand between the sampling kernel and the distance kernel, profiles/r05_mlsl_timeline.txt).  A run that stops inside that walk
 * leaves; what the pass wrote is not looked at again. */
#define ENQUEUE_DISTANCES() do { \
            const int na_ = D.N, nb_ = (int) (old + (size_t) D.N); \
            const int per_r_ = (na_ + D.world - 1) / D.world; \
            const int r0_ = per_r_ * D.rank < na_ ? per_r_ * D.rank : na_; \
            const int mine_r_ = na_ - r0_ < per_r_ ? na_ - r0_ : per_r_; \
            const double *A_ = D.d_P + (old + (size_t) r0_) * (size_t) D.ld, *FA_ = D.d_F + old + r0_; \
            if (need_D(&D, (size_t) na_ * (size_t) nb_) || (D.nlms && need_D(&D, (size_t) na_ * D.nlms))) DEVFAIL(); \
            /* closest_pt_d of the new points: over every point with smaller f; of the old, not yet minimised points: over the new points \
             * with smaller f.  Several ranks: the ROWS of the new points are dealt in blocks over the ranks; every rank computes its rows' \
             * minima and its rows' share of the column minima, combined by an element-wise min over an all-gather (exact) */ \
            if (nla_memcpy_h2d(D.d_cpd, D.cpd, sizeof(double) * (size_t) nb_, D.st) || \
                nla_memcpy_h2d(D.d_min, D.minimized, sizeof(int32_t) * (size_t) nb_, D.st) || \
                (mine_r_ > 0 && (nla_k_mlsl_dist2(n, D.ld, A_, mine_r_, D.d_P, nb_, D.d_D, D.st) || \
                                 nla_k_mlsl_rowmin(D.d_D, nb_, mine_r_, nb_, FA_, D.d_F, NULL, D.d_cpd + old + r0_, D.st) || \
                                 nla_k_mlsl_colmin(D.d_D, nb_, mine_r_, (int) old, FA_, D.d_F, D.d_min, D.d_cpd, D.st))) || \
                nla_memcpy_d2h(D.cpd, D.d_cpd, sizeof(double) * (size_t) nb_, D.st)) { snprintf(D.err, sizeof D.err, "distance pass failed"); DEVFAIL(); } \
            if (D.nlms) {                                                      /* find_closest_lm */ \
                if (mine_r_ > 0 && (nla_k_mlsl_dist2(n, D.ld, A_, mine_r_, D.d_LM, (int) D.nlms, D.d_D, D.st) || \
                                    nla_k_mlsl_rowmin(D.d_D, (int) D.nlms, mine_r_, (int) D.nlms, FA_, D.d_LF, NULL, D.d_tmp, D.st) || \
                                    nla_memcpy_d2h(D.cld + old + r0_, D.d_tmp, sizeof(double) * (size_t) mine_r_, D.st))) { snprintf(D.err, sizeof D.err, "distance pass failed"); DEVFAIL(); } \
            } \
            dist_enqueued = 1; } while (0)
#define NEWPT_UNORDERED(row) do { (void) (row); ++D.npts; } while (0)     /* (entries initialised in front of the walk; ord_insert_run follows) */
#define NEWPT(row) do { D.minimized[row] = 0; D.cpd[row] = HUGE_VAL; D.cld[row] = HUGE_VAL; ord_insert(D.ord, D.npts, D.F, row); ++D.npts; } while (0)
    /* several ranks: the clock and the force_stop flag are decided by all ranks together at the start of every phase (comm.c);
     * sp is what those two tests look at until the next agreement */
#define AGREE() do { sp = nla_comm_agree_stop(D.comm, stop, &agreed_view, &agreed_force); \
        if (!sp) { snprintf(D.err, sizeof D.err, "stop agreement failed: %s", nlopt_amd_comm_error(D.comm)); DEVFAIL(); } } while (0)
#define STOPS(fv) do { if (nla_stop_forced(sp)) ret = NLOPT_FORCED_STOP; else if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED; \
        else if (nla_stop_time(sp)) ret = NLOPT_MAXTIME_REACHED; else if ((fv) < stop->minf_max) ret = NLOPT_MINF_MAX_REACHED; } while (0)
#define GET_MINF() do { if (D.npts) { best_f = D.F[D.ord[0]]; best_row = D.ord[0]; best_is_lm = 0; } \
        if (D.nlms && D.LF[D.lord[0]] < best_f) { best_f = D.LF[D.lord[0]]; best_row = D.lord[0]; best_is_lm = 1; } } while (0)

    /* f of `cnt` rows on the device: compiled-in objective, or the user's kernel (host objectives are called in the loops below) */
#define EVAL_ROWS(rows, cnt, dF) (D.ev.kind == NLA_EVAL_USER ? nla_userobj_eval_rows(D.ev.user, n, D.ld, (cnt), (rows), (dF), NULL, D.ev.sign, D.st) \
                                                              : (nla_k_eval(D.obj, n, D.ld, (rows), (cnt), (dF), D.st) || \
                                                                 (D.ev.sign < 0 && nla_k_mlsl_negate((dF), (int) (cnt), D.st))))
    /* the starting guess is the first point (mlsl.c:326-340) */
    if (host) {
        D.F[0] = f((unsigned) n, x, NULL, f_data);
        if (nla_memcpy_h2d(D.d_P, x, sizeof(double) * (size_t) n, D.st) || nla_memcpy_h2d(D.d_F, D.F, sizeof(double), D.st) ||
            nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "first evaluation failed"); DEVFAIL(); }
    } else if (nla_memcpy_h2d(D.d_P, x, sizeof(double) * (size_t) n, D.st) || EVAL_ROWS(D.d_P, 1, D.d_F) ||
        nla_memcpy_d2h(D.F, D.d_F, sizeof(double), D.st) || nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "first evaluation failed"); DEVFAIL(); }
    ++*stop->nevals_p;
    NEWPT(0);
    if (D.F[0] != D.F[0]) D.ord_has_nan = 1;
    AGREE();
    STOPS(D.F[0]);

    while (ret == NLOPT_SUCCESS) {
        double R, t0 = nla_seconds();
        size_t old = D.npts, used = 0, idx;
        int remaining, prefetch_due = 0, ahead_due = 0, dist_enqueued = 0, pairs_enqueued = 0, ahead_hit;
        mlsl_pairs_job pairs_job;
        GET_MINF();                                                            /* mlsl.c:347 */
        AGREE();
        if (opt && opt->progress) { opt->progress(opt->progress_data, st ? (long) st->generations : 0, (long) *stop->nevals_p); t0 = nla_seconds(); }

        /* ---- sampling phase (mlsl.c:349-374) ---- */
        if (grow_pts(&D, old + (size_t) D.N)) DEVFAIL();
        ahead_hit = D.ahead_valid && D.ahead_old == old && (D.d_V ? D.ahead_sobol == D.sobol_next : D.ahead_words == D.words_used);
        D.ahead_valid = 0;
        if (ahead_hit) {
            /* the samples were made beside the last local phase (mlsl_enqueue_ahead): nothing to launch for them */
        } else if (D.d_V) {                                                    /* nlopt_sobol_next, mlsl.c:355 */
            if ((uint64_t) D.sobol_next + (uint64_t) D.N >= 4294967295ULL) { snprintf(D.err, sizeof D.err, "Sobol sequence exhausted (2^32-1 points)"); DEVFAIL(); }
            if (nla_k_mlsl_sobol_rows(n, D.ld, D.d_lb, D.d_ub, D.d_V, D.sobol_next, D.N, D.d_P + old * (size_t) D.ld, D.st) ||
                (!host && EVAL_ROWS(D.d_P + old * (size_t) D.ld, D.N, D.d_F + old))) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
        } else {
            if (D.prefetched_at != D.words_used &&
                nla_mtstream_fill(D.mts, D.words_used, 2ULL * (uint64_t) n * (uint64_t) D.N, D.d_words)) { snprintf(D.err, sizeof D.err, "MT stream fill failed"); DEVFAIL(); }
            if (D.rs != D.st && nla_stream_sync(D.rs)) { snprintf(D.err, sizeof D.err, "MT stream fill failed"); DEVFAIL(); }   /* the words are there */
            if (nla_k_crs_init_rows(D.obj, n, D.ld, D.d_lb, D.d_ub, D.d_words, (int64_t) old, D.N, D.d_P, D.d_F, D.st) ||
                (D.ev.kind == NLA_EVAL_DEVICE && D.ev.sign < 0 && nla_k_mlsl_negate(D.d_F + old, D.N, D.st)) ||
                (D.ev.kind == NLA_EVAL_USER && EVAL_ROWS(D.d_P + old * (size_t) D.ld, D.N, D.d_F + old))) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
        }
        /* the new rows' closest-distance entries and flags start as "none yet" (what NEWPT does per point; here for all N at once so
         * that the distance pass can be enqueued before the host looks at a single value) */
        for (i = 0; i < D.N; ++i) { D.minimized[old + (size_t) i] = 0; D.cpd[old + (size_t) i] = HUGE_VAL; D.cld[old + (size_t) i] = HUGE_VAL; }
        dist_enqueued = 0;
        if (ahead_hit) {
            /* what is left of the distance pass: the new rows against the minima found since (this stream's work waits for the rows) */
            const size_t l0 = D.ahead_nlms, lc = D.nlms - D.ahead_nlms;
            size_t j;
            if (nla_stream_wait_event(D.st, D.ev_ahead)) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
            if (lc && (need_D(&D, (size_t) D.N * lc) ||
                       nla_k_mlsl_dist2(n, D.ld, D.d_P + old * (size_t) D.ld, D.N, D.d_LM + l0 * (size_t) D.ld, (int) lc, D.d_D, D.st) ||
                       nla_k_mlsl_rowmin(D.d_D, (int) lc, D.N, (int) lc, D.d_F + old, D.d_LF + l0, NULL, D.d_tmp, D.st) ||
                       nla_memcpy_d2h(D.cld + old, D.d_tmp, sizeof(double) * (size_t) D.N, D.st))) { snprintf(D.err, sizeof D.err, "distance pass failed"); DEVFAIL(); }
            if (nla_event_sync(D.ev_ahead)) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
            memcpy(Fnew, D.Fnew2, sizeof(double) * (size_t) D.N);
            if (st) ++st->mlsl_sampled_ahead;
            for (j = 0; j < old + (size_t) D.N; ++j) if (D.h_cpd2[j] < D.cpd[j]) D.cpd[j] = D.h_cpd2[j];
            dist_enqueued = 1;
        } else
        if (host ? nla_memcpy_d2h(D.h_rows, D.d_P + old * (size_t) D.ld, sizeof(double) * (size_t) D.N * (size_t) D.ld, D.st)
                 : nla_memcpy_d2h(Fnew, D.d_F + old, sizeof(double) * (size_t) D.N, D.st)) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
        if (ahead_hit) {
            /* (everything is enqueued or done) */
        } else if (!host && D.world == 1) {
            /* wait for the values only (an event behind their copy), with the distance pass already behind them on the stream */
            if (nla_event_record(D.ev_samples, D.st)) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
            ENQUEUE_DISTANCES();
            if (nla_event_sync(D.ev_samples)) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
        } else if (nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
        for (i = 0; i < D.N && ret == NLOPT_SUCCESS; ++i) {
            if (host) Fnew[i] = f((unsigned) n, D.h_rows + (size_t) i * (size_t) D.ld, NULL, f_data);   /* mlsl.c:360 */
            D.F[old + (size_t) i] = Fnew[i];
            ++*stop->nevals_p;
            if (st) ++st->evals_trial;
            NEWPT_UNORDERED(old + (size_t) i);
            used = (size_t) i + 1;
            if (opt && opt->trace) {
                if (opt->trace_len < opt->trace_cap) { nlopt_amd_trace_rec *tr = opt->trace + opt->trace_len; tr->f = Fnew[i]; tr->row = (int64_t) (old + (size_t) i); tr->kind = 3; tr->accepted = 0; }
                ++opt->trace_len;
            }
            STOPS(Fnew[i]);
        }
        {
            int any_nan = D.ord_has_nan;
            ord_insert_run(D.ord, old, D.F, old, used, any_nan);
            for (i = 0; i < (int) used && !D.ord_has_nan; ++i) if (D.F[old + (size_t) i] != D.F[old + (size_t) i]) D.ord_has_nan = 1;
        }
        if (D.d_V) D.sobol_next += (uint32_t) used;
        else D.words_used += 2ULL * (uint64_t) n * (uint64_t) used;
        if (host && (nla_memcpy_h2d(D.d_F + old, Fnew, sizeof(double) * used, D.st) || nla_stream_sync(D.st))) { snprintf(D.err, sizeof D.err, "sampling failed"); DEVFAIL(); }
        if (ret != NLOPT_SUCCESS) break;
        /* the sampling kernel has read the words (synchronised above): the next iteration's can be generated.  NOT here (round 5,
         * profiles/r05_mlsl_timeline.txt): enqueued now, the jump-ahead kernel — 33 k workgroups, 0.55 ms, once or twice per iteration —
         * and the generator had the device to themselves and the distance pass queued up behind them, 0.9-1.8 ms of every iteration.
         * They go out right in front of the local searches' launch instead (PREFETCH_NOW below), on a background-priority stream: the
         * searches' 300 workgroups leave most of every compute unit free for 6 ms */
        prefetch_due = D.prefetch && !D.d_V;
        ahead_due = D.ahead;
        if (!dist_enqueued) ENQUEUE_DISTANCES();
        {
            const int na = D.N;
            if (nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "distance pass failed"); DEVFAIL(); }
            if (ahead_hit && D.ahead_nlms)                                      /* closest_lm_d: the minima known when the rows were made | those found since */
                for (i = 0; i < na; ++i) if (D.h_cld2[i] < D.cld[old + (size_t) i]) D.cld[old + (size_t) i] = D.h_cld2[i];
            if (D.world > 1 && (min_over_ranks(&D, D.cpd, D.npts) || (D.nlms && min_over_ranks(&D, D.cld + old, (size_t) na)))) {
                snprintf(D.err, sizeof D.err, "all-gather of the distance minima failed: %s", nlopt_amd_comm_error(D.comm)); DEVFAIL();
            }
        }
        if (st) st->t_eval_s += nla_seconds() - t0;

        /* ---- local phase (mlsl.c:377-428) ---- */
        t0 = nla_seconds();
        R = R_prefactor * pow(log((double) D.npts) / D.npts, 1.0 / n);
        idx = 0;
        remaining = (int) (ceil(MLSL_GAMMA * D.npts) + 0.5);
        while (idx < D.npts && remaining > 0 && ret == NLOPT_SUCCESS) {
            int nb = 0, c, per, mine = 0, nacc = 0;
            size_t scan = idx, nlms0 = 0;
            int rem = remaining, eff;
            long limited;
            /* the next candidates that are potential minimisers right now (is_potential_minimizer, :196-221) */
            while (scan < D.npts && rem > 0 && nb < bmax) {
                const size_t r = D.ord[scan];
                ++scan; --rem;
                if (D.minimized[r] || D.cpd[r] <= R * R || D.cld[r] <= (dlm * R) * (dlm * R)) continue;
                cand[nb++] = scan - 1;
            }
            if (nb == 0) { idx = scan; remaining = rem; break; }
            AGREE();
            /* candidate c is minimised by rank c mod world in its slot c / world; gathered row of c: GI(c) */
            per = (nb + D.world - 1) / D.world;
#define GI(c) ((size_t) ((c) % D.world) * (size_t) per + (size_t) ((c) / D.world))
            /* start points of this rank's share into the batch (one gather launch), and the bound test of every candidate
             * (is_potential_minimizer, mlsl.c:211-218) on the device: only the flags come back */
            for (c = 0; c < nb; ++c) D.h_idx[c] = (int64_t) D.ord[cand[c]];
            for (c = D.rank; c < nb; c += D.world, ++mine) D.h_idx[bmax + mine] = (int64_t) D.ord[cand[c]];
            if (nla_memcpy_h2d(D.d_idx, D.h_idx, sizeof(int64_t) * (size_t) (bmax + mine), D.st) ||
                nla_k_mlsl_gather_rows(n, D.ld, D.d_P, D.d_idx + bmax, mine, use_cobyla ? D.d_LX : nla_local_ctx_X(D.lb), D.st) ||
                nla_k_mlsl_near_bound(n, D.ld, D.d_P, D.d_idx, nb, D.d_lb, D.d_ub, dbound * R, D.d_flags, D.st) ||
                nla_memcpy_d2h(D.h_flags, D.d_flags, sizeof(int32_t) * (size_t) nb, D.st)) { snprintf(D.err, sizeof D.err, "gather failed"); DEVFAIL(); }
            /* local searches of this rank's share, all-gather of the minimisers, then their distances to every point */
            if (host && D.world == 1) {
                /* a host objective sees its calls: the tests the reference makes in front of a search (mlsl.c:390-399) are made
                 * here, before the search's first callback, not at commit time */
                if (nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "gather failed"); DEVFAIL(); }
                if (D.h_flags[0]) mine = 0;                                         /* too close to a bound: not started */
                else if (nla_stop_forced(stop)) ret = NLOPT_FORCED_STOP;
                else if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED;
                else if (stop->maxtime > 0 && nla_seconds() - stop->start >= stop->maxtime) ret = NLOPT_MAXTIME_REACHED;
                if (ret != NLOPT_SUCCESS) break;
            }
            limited = (long) stop->maxeval - (long) *stop->nevals_p;             /* nlopt_optimize_limited, optimize.c:1097-1100 */
            eff = loc_maxeval;
            if (loc_maxeval <= 0 || (limited > 0 && limited < loc_maxeval)) eff = (int) limited;
            prm.maxeval = eff;
            if (use_cobyla) {
                memset(&res[0], 0, sizeof res[0]);
                if (mine > 0) {                                                     /* mlsl.c:400-407 */
                    const double t = nla_seconds();
                    double lf = HUGE_VAL;
                    nlopt_result lret;
                    if (nla_memcpy_d2h(cob_x, D.d_LX, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "gather failed"); DEVFAIL(); }
                    lret = nla_optimize_limited(local_opt, cob_x, &lf, stop->maxeval - *stop->nevals_p, stop->maxtime - (t - stop->start));
                    if (nla_memcpy_h2d(D.d_LX, cob_x, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "minimum store failed"); DEVFAIL(); }
                    res[0].ret = (int) lret; res[0].f = lf;
                    res[0].nevals = 0; res[0].iterm = 0;                            /* the calls were counted one by one (mlsl_counted_f) */
                }
            } else {
            pairs_enqueued = 0;
            if (!host && D.world == 1 && mine > 0) {
                /* one rank, device objective: the minimisers' distances go out right behind the searches' kernel (they need nothing the
                 * host decides), instead of after the host has woken up, copied the results and come back: 0.2 ms per iteration */
                pairs_job.d = &D; pairs_job.n = n; pairs_job.nb = nb; pairs_job.na = (size_t) per; pairs_job.ctx_X = nla_local_ctx_X(D.lb);
                pairs_job.lx_bytes = sizeof(double) * (size_t) per * (size_t) D.ld;
                if (need_D(&D, (size_t) per * D.npts)) DEVFAIL();
                for (c = 0; c < nb; ++c) D.h_gi[c] = (int64_t) GI(c);
                nla_local_ctx_after_launch(D.lb, mlsl_enqueue_pairs, &pairs_job);
                pairs_enqueued = 1;
            }
            /* (in front of the launch, behind everything that may allocate: freeing device memory waits for every stream, the gate included) */
            D.gate_need = 0;
            if (D.ahead && mine > 0 && (D.gate_counter = nla_local_ctx_finished_counter(D.lb, &D.gate_from))) D.gate_need = (mine * NLA_MLSL_GATE_PCT + 99) / 100;
            PREFETCH_NOW();
            D.gate_need = 0;
            if (mine > 0 && nla_local_ctx_run(D.lb, mine, &prm, res_mine, &lstop, NULL)) { snprintf(D.err, sizeof D.err, "local-search batch failed"); DEVFAIL(); }
            if ((!pairs_enqueued && nla_comm_allgather_dev(D.comm, nla_local_ctx_X(D.lb), D.d_LX, sizeof(double) * (size_t) per * (size_t) D.ld, D.st)) ||
                nla_comm_allgather_host(D.comm, res_mine, res, sizeof *res * (size_t) per, D.st)) {
                snprintf(D.err, sizeof D.err, "all-gather of the local minima failed: %s", nlopt_amd_comm_error(D.comm)); DEVFAIL();
            }
            }
            {
                /* the distances of the batch's minimisers to every point stay on the device (na x npts: 12 MB at config 4); the commit walk
                 * below only needs those to the batch's OWN start points (whether an earlier commit of the batch disqualified a later
                 * candidate, mlsl.c:180-194 seen from :196-221) — nb x nb values — and pts_update_newlm for all the other points is an
                 * order-independent min that one launch computes once the accepted set is known */
                const size_t na = (size_t) per * (size_t) D.world;
                if (need_D(&D, na * D.npts)) DEVFAIL();
                if (!pairs_enqueued) {
                    for (c = 0; c < nb; ++c) D.h_gi[c] = (int64_t) GI(c);
                    pairs_job.d = &D; pairs_job.n = n; pairs_job.nb = nb; pairs_job.na = na; pairs_job.ctx_X = NULL; pairs_job.lx_bytes = 0;
                    if (mlsl_enqueue_pairs(&pairs_job)) { snprintf(D.err, sizeof D.err, "distance pass failed"); DEVFAIL(); }
                }
                if (nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "distance pass failed"); DEVFAIL(); }
                for (c = 0; c < (int) na; ++c) D.h_lfall[c] = HUGE_VAL;
                for (c = 0; c < nb; ++c) D.lf_cand[c] = HUGE_VAL;
            }
            if (grow_lms(&D, D.nlms + (size_t) nb)) DEVFAIL();
            nacc = 0; nlms0 = D.nlms;
            /* commit in walk order */
            for (c = 0; c < nb && ret == NLOPT_SUCCESS; ++c) {
                const size_t r = D.ord[cand[c]];
                double lf, cl = D.cld[r];
                int calls, pot, cp;
                const size_t g = GI(c);
                /* closest_lm_d of this start as the serial order would see it now: the minima committed earlier in this batch count
                 * (pts_update_newlm, mlsl.c:180-194: those with smaller f than the point's, if closer) */
                /* (h_S[c][cp] = distance of start c to minimiser cp: transposed on the device.  nb^2 / 2 questions per batch — 46 k at config 4:
                 * with the gathered-row index GI(cp), a division and a remainder, inside the loop they were the better part of the 0.6 ms the
                 * device waited for this walk) */
                {
                    const double Fr = D.F[r], *Sc = D.h_S + (size_t) c * (size_t) nb, *lfc = D.lf_cand;
                    for (cp = 0; cp < c; ++cp)
                        if (lfc[cp] < Fr && Sc[cp] < cl) cl = Sc[cp];
                }
                pot = !(cl <= (dlm * R) * (dlm * R));
                /* nodes between the previous candidate and this one were visited and skipped */
                remaining -= (int) (cand[c] + 1 - idx);
                idx = cand[c] + 1;
                if (pot && D.h_flags[c]) pot = 0;                               /* too close to a bound (mlsl.c:211-218) */
                if (!pot) continue;
                if (!(host && D.world == 1)) {
                    if (nla_stop_forced(sp)) { ret = NLOPT_FORCED_STOP; break; }
                    if (nla_stop_evals(stop)) { ret = NLOPT_MAXEVAL_REACHED; break; }
                    if (nla_stop_time(sp)) { ret = NLOPT_MAXTIME_REACHED; break; }
                }
                /* did this search run under the evaluation limit it would have had in the serial order? */
                limited = (long) stop->maxeval - (long) *stop->nevals_p;
                eff = loc_maxeval;
                if (loc_maxeval <= 0 || (limited > 0 && limited < loc_maxeval)) eff = (int) limited;
                if (eff > 0 && eff != prm.maxeval && res[g].nevals >= eff) {
                    /* the limit binds differently than assumed: redo this one search alone with the exact limit
                     * (every rank does, identically; the gathered rows of the other candidates are untouched) */
                    nla_lbfgs_params p1 = prm;
                    nla_lbfgs_result r1;
                    p1.maxeval = eff;
                    if (nla_memcpy_d2d(nla_local_ctx_X(D.lb), D.d_P + r * (size_t) D.ld, sizeof(double) * (size_t) n, D.st) ||
                        nla_local_ctx_run(D.lb, 1, &p1, &r1, &lstop, NULL) ||
                        nla_memcpy_d2d(D.d_LX + g * (size_t) D.ld, nla_local_ctx_X(D.lb), sizeof(double) * (size_t) n, D.st) ||
                        nla_k_mlsl_dist2(n, D.ld, D.d_LX + g * (size_t) D.ld, 1, D.d_P, (int) D.npts, D.d_D + g * D.npts, D.st) ||
                        nla_memcpy_d2h(D.h_D, D.d_D + g * D.npts, sizeof(double) * D.npts, D.st) || nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "local-search rerun failed"); DEVFAIL(); }
                    for (cp = 0; cp < nb; ++cp) D.h_S[(size_t) cp * (size_t) nb + (size_t) c] = D.h_D[D.ord[cand[cp]]];     /* this minimiser moved: its distances to the batch's start points */
                    res[g] = r1;
                }
                calls = use_mma ? res[g].iterm : res[g].nevals;                /* objective calls the search made (MMA: iterm) */
                *stop->nevals_p += calls;                                       /* fcount, mlsl.c:246-251 */
                if (st) { st->evals_mutation += (uint64_t) calls; ++st->accepted; }
                D.minimized[r] = 1;
                if (opt && opt->trace) {
                    if (opt->trace_len < opt->trace_cap) { nlopt_amd_trace_rec *tr = opt->trace + opt->trace_len; tr->f = res[g].f; tr->row = (int64_t) r; tr->kind = 4; tr->accepted = calls; }
                    ++opt->trace_len;
                }
                if (res[g].ret < 0) { ret = (nlopt_result) res[g].ret; goto done_noget; }
                lf = res[g].f;
                /* the minimum joins the device-side set; stream-ordered, no synchronisation per minimum (room for the whole
                 * batch was made before the walk; res[] stays valid until the batch's closing synchronisation) */
                D.h_idx[bmax + BATCH_MAX + nacc] = (int64_t) g;
                D.h_lf[nacc] = lf;
                D.h_lfall[g] = lf;
                D.lf_cand[c] = lf;
                ++nacc;
                D.LF[D.nlms] = lf;
                ord_insert(D.lord, D.nlms, D.LF, D.nlms);
                ++D.nlms;
                if (nla_stop_forced(sp)) ret = NLOPT_FORCED_STOP;
                else if (lf < stop->minf_max) ret = NLOPT_MINF_MAX_REACHED;
                else if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED;
                else if (nla_stop_time(sp)) ret = NLOPT_MAXTIME_REACHED;
                /* (pts_update_newlm for this minimum: with the batch's other accepted minima, in one launch below.  A minimum whose commit
                 * ends the run updated nothing in the reference either — the stop tests come first, mlsl.c:417-425 — so it is left out.) */
                if (ret != NLOPT_SUCCESS) { D.h_lfall[g] = HUGE_VAL; D.lf_cand[c] = HUGE_VAL; }
            }
            /* pts_update_newlm (mlsl.c:180-194) for every accepted minimum of the batch at once: closest_lm_d[k] = min over the accepted
             * minima with smaller f than point k of their distance to k — a min, so the order of the commits does not matter.  (The
             * reference skips minimised points; their closest_lm_d is never read again, mlsl.c:200, so no flag is consulted here.) */
            if (nacc > 0 && (nla_memcpy_h2d(D.d_lfall, D.h_lfall, sizeof(double) * (size_t) per * (size_t) D.world, D.st) ||
                             nla_memcpy_h2d(D.d_cpd, D.cld, sizeof(double) * D.npts, D.st) ||
                             nla_k_mlsl_colmin(D.d_D, (int) D.npts, per * D.world, (int) D.npts, D.d_lfall, D.d_F, NULL, D.d_cpd, D.st) ||
                             nla_memcpy_d2h(D.cld, D.d_cpd, sizeof(double) * D.npts, D.st))) { snprintf(D.err, sizeof D.err, "closest-minimum update failed"); DEVFAIL(); }
            /* the batch's accepted minima join the device-side set in acceptance order: one gather launch */
            if (nacc > 0 && (nla_memcpy_h2d(D.d_idx + bmax + BATCH_MAX, D.h_idx + bmax + BATCH_MAX, sizeof(int64_t) * (size_t) nacc, D.st) ||
                             nla_k_mlsl_gather_rows(n, D.ld, D.d_LX, D.d_idx + bmax + BATCH_MAX, nacc, D.d_LM + nlms0 * (size_t) D.ld, D.st) ||
                             nla_memcpy_h2d(D.d_LF + nlms0, D.h_lf, sizeof(double) * (size_t) nacc, D.st))) { snprintf(D.err, sizeof D.err, "minimum store failed"); DEVFAIL(); }
            if (nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "minimum store failed"); DEVFAIL(); }
            if (ret == NLOPT_SUCCESS && c == nb) {
                /* nodes scanned after the last candidate of the batch (none qualified) are visited too */
                remaining -= (int) (scan - idx);
                idx = scan;
            }
        }
        PREFETCH_NOW();                                                        /* (an iteration without a local search) */
        if (st) { st->t_evolve_s += nla_seconds() - t0; ++st->generations; }
    }
    GET_MINF();                                                                /* mlsl.c:431 */
done_noget:
    if (ret != NLOPT_FAILURE || best_f < HUGE_VAL) {
        const double *src = best_is_lm ? D.d_LM + best_row * (size_t) D.ld : D.d_P + best_row * (size_t) D.ld;
        if (best_f < HUGE_VAL) {
            *minf = best_f;
            if (nla_memcpy_d2h(x, src, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) { nla_stop_msg(stop, "device engine: result read-back failed"); ret = NLOPT_FAILURE; }
        }
    }
done:
    if (st) st->mt_words = D.words_used;
    if (use_cobyla) { local_opt->f = lo_f; local_opt->f_data = lo_fdata; nla_host_free(cob_x); }   /* (the wrapper's data live on this stack) */
    mfree(&D);
    nla_host_free(Fnew); free(res); free(res_mine); free(cand);
    return ret;
}
