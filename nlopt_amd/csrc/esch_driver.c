#define _GNU_SOURCE            /* qsort_r */
/* esch_driver.c — NLOPT_GN_ESCH behind the reference's entry point
 *   chevolutionarystrategy(n, f, f_data, lb, ub, x, minf, stop, np, no)   (src/algs/esch/esch.h; dispatched at
 *   optimize.c:946-949 with np = population, no = (unsigned)(population * 1.5); 0 -> 40 / 60, esch.c:96-97),
 * host side: the generation loop, the best-point / stop tests the reference makes after EVERY candidate (esch.c:168-183,
 * 222-238, replayed in candidate order over the device's fitness array), the stream accounting and the kernel
 * sequencing (hip/esch_kernels.hip).  SURVEY.md §8f.1.
 *
 * Stream use per phase (the thread's MT19937 generator is left where the reference's would be):
 *   initial populations   2 words per randcauchy attempt, (np + no) n accepted values needed
 *   generation            3 no words (crossover), then the mutation chain: per step 2 words + 2 per attempt
 * A device objective runs entirely on the GPU; any other callback is called on the caller's thread, candidate by
 * candidate in the reference's order, on rows copied back (the evolution itself stays on the device).
 */
#include "nla_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, ld, obj;                /* obj: compiled-in objective id, -2: user-supplied kernel (ev), -1: host callback */
    nla_evaluator ev;
    int64_t np, no, P;
    void *st;
    nla_mtstream *mts;
    uint64_t words_used;
    double *d_lb, *d_ub, *d_R, *d_G, *d_F, *d_fit[2], *d_v;
    int32_t *d_slot[2], *d_last, *d_counts;
    int64_t *d_vatt, *d_out;
    uint32_t *d_words;
    void *d_mscratch, *d_sscratch;
    size_t wcap, mscratch_bytes, sscratch_bytes;
    double *h_fit, *h_G;
    char err[200];
} esch_dev;

#define EFAIL(d, ...) do { snprintf((d)->err, sizeof (d)->err, __VA_ARGS__); return -1; } while (0)
#define ECK(d, call) do { int rc_ = (call); if (rc_) EFAIL(d, "%.90s failed: %.60s", #call, nla_dev_error_string(rc_)); } while (0)

static void efree(esch_dev *d)
{
    if (d->st) nla_stream_sync(d->st);
    if (d->mts) { nla_mtstream_finish(d->mts, d->words_used); nla_mtstream_destroy(d->mts); }
    nla_dev_free(d->d_lb); nla_dev_free(d->d_ub); nla_dev_free(d->d_R); nla_dev_free(d->d_G); nla_dev_free(d->d_F);
    nla_dev_free(d->d_fit[0]); nla_dev_free(d->d_fit[1]); nla_dev_free(d->d_v); nla_dev_free(d->d_slot[0]); nla_dev_free(d->d_slot[1]);
    nla_dev_free(d->d_last); nla_dev_free(d->d_counts); nla_dev_free(d->d_vatt); nla_dev_free(d->d_out); nla_dev_free(d->d_words);
    nla_dev_free(d->d_mscratch); nla_dev_free(d->d_sscratch);
    nla_host_free(d->h_fit); nla_host_free(d->h_G);
    if (d->st) nla_stream_destroy(d->st);
}

static int need_words(esch_dev *d, size_t words)
{
    if (words <= d->wcap) return 0;
    nla_dev_free(d->d_words); nla_dev_free(d->d_counts);
    d->wcap = words + words / 4;
    d->d_words = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * d->wcap);
    d->d_counts = (int32_t *) nla_dev_malloc(sizeof(int32_t) * (d->wcap / 2 / 1024 + 16));
    if (!d->d_words || !d->d_counts) { d->wcap = 0; EFAIL(d, "out of device memory (stream words)"); }
    return 0;
}

/* the (np + no) n accepted Cauchy values of the initial populations, in draw order (esch.c:133-164) */
static int init_rows(esch_dev *d, const double *x0)
{
    const int64_t E = d->P * (int64_t) d->n;
    int64_t have = 0, attempts_done = 0, last_att;
    ECK(d, nla_memset(d->d_out, 0, 2 * sizeof(int64_t), d->st));
    while (have < E) {
        int64_t a = (int64_t) (1.2 * (double) (E - have) / 0.874) + 2048;
        if (a > (1LL << 28)) a = 1LL << 28;
        if (need_words(d, (size_t) (2 * a))) return -1;
        if (nla_mtstream_fill(d->mts, d->words_used + 2ULL * (uint64_t) attempts_done, 2ULL * (uint64_t) a, d->d_words)) EFAIL(d, "MT stream fill failed");
        ECK(d, nla_k_esch_cauchy(d->d_words, a, attempts_done, d->d_counts, d->d_out, have, E, d->d_v, d->d_vatt, d->st));
        ECK(d, nla_memcpy_d2h(&have, d->d_out, sizeof have, d->st));
        ECK(d, nla_stream_sync(d->st));
        attempts_done += a;
    }
    ECK(d, nla_memcpy_d2h(&last_att, d->d_vatt + (E - 1), sizeof last_att, d->st));
    ECK(d, nla_k_esch_fill_rows(d->n, d->ld, d->d_lb, d->d_ub, d->d_v, 0, E, d->d_R, d->st));
    ECK(d, nla_memcpy_h2d(d->d_R, x0, sizeof(double) * (size_t) d->n, d->st));          /* parent 0 := x (esch.c:148) */
    ECK(d, nla_stream_sync(d->st));
    d->words_used += 2ULL * (uint64_t) (last_att + 1);
    return 0;
}

/* individuals [i0, i0 + count): rows gathered; device objective: their fitness in h_fit; otherwise the rows in h_G for the
 * callback, which the caller invokes candidate by candidate (the reference stops calling f the moment a stop test fires) */
static int evaluate(esch_dev *d, int cur, int64_t i0, int64_t count)
{
    ECK(d, nla_k_esch_gather_rows(d->n, d->ld, d->d_slot[cur], i0, count, d->d_R, d->d_G, d->st));
    if (d->obj != -1) {
        if (d->obj >= 0) ECK(d, nla_k_eval(d->ev.sign < 0 ? (d->obj | NLA_OBJ_NEGATE) : d->obj, d->n, d->ld, d->d_G, count, d->d_F, d->st));
        else ECK(d, nla_userobj_eval_rows(d->ev.user, d->n, d->ld, count, d->d_G, d->d_F, NULL, d->ev.sign, d->st));
        ECK(d, nla_memcpy_d2h(d->h_fit + i0, d->d_F, sizeof(double) * (size_t) count, d->st));
    } else
        ECK(d, nla_memcpy_d2h(d->h_G, d->d_G, sizeof(double) * (size_t) count * (size_t) d->ld, d->st));
    ECK(d, nla_stream_sync(d->st));
    return 0;
}

/* Selection of a generation with NaN fitness values, on the host and literally as the reference does it: the device's stable sort
 * works on order-preserving keys (a total order) and a NaN has none; the reference's comparator (esch.c:59-64) calls a NaN equal to
 * everything and its result is what glibc's qsort_r (nlopt_qsort_r on Linux, util/qsort_r.c:164-170) makes of those answers on the
 * np + no records of 16 bytes (esch.c:243).  Same libc, same comparator, records of the same size in the same order: same result. */
typedef struct { int64_t slot; double fitness; } esch_rec;
static int esch_rec_compare(const void *a_, const void *b_, void *unused)               /* CompareIndividuals, esch.c:59-64 */
{
    const esch_rec *a = (const esch_rec *) a_, *b = (const esch_rec *) b_;
    (void) unused;
    return a->fitness < b->fitness ? -1 : (a->fitness > b->fitness ? +1 : 0);
}
static int select_with_nan(esch_dev *d, int cur)
{
    const int64_t P = d->P;
    esch_rec *rec = (esch_rec *) malloc(sizeof(esch_rec) * (size_t) P);
    int32_t *slot = (int32_t *) malloc(sizeof(int32_t) * (size_t) P);
    int64_t i;
    int rc = -1;
    if (!rec || !slot) goto out;
    if (nla_memcpy_d2h(slot, d->d_slot[cur], sizeof(int32_t) * (size_t) P, d->st) || nla_stream_sync(d->st)) goto out;
    for (i = 0; i < P; ++i) { rec[i].slot = slot[i]; rec[i].fitness = d->h_fit[i]; }
    qsort_r(rec, (size_t) P, sizeof(esch_rec), esch_rec_compare, NULL);
    for (i = 0; i < P; ++i) { slot[i] = (int32_t) rec[i].slot; d->h_fit[i] = rec[i].fitness; }
    if (nla_memcpy_h2d(d->d_slot[cur ^ 1], slot, sizeof(int32_t) * (size_t) P, d->st) ||
        nla_memcpy_h2d(d->d_fit[cur ^ 1], d->h_fit, sizeof(double) * (size_t) P, d->st) || nla_stream_sync(d->st)) goto out;
    rc = 0;
out:
    free(rec); free(slot);
    return rc;
}

nlopt_result nla_esch_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                               nla_stopping *stop, unsigned np_, unsigned no_)
{
    esch_dev D;
    nlopt_result ret = NLOPT_SUCCESS;
    nlopt_amd_stats *st = opt ? &opt->stats : NULL;
    int cur = 0, total;
    int64_t i, kbest;
    memset(&D, 0, sizeof D);
    D.np = np_ ? np_ : 40; D.no = no_ ? no_ : 60;                                  /* esch.c:96-97 */
    if (D.np < 1 || D.no < 1) { nla_stop_msg(stop, "populations %d, %d are too small", (int) D.np, (int) D.no); return NLOPT_INVALID_ARGS; }
    if (nla_dev_count() <= 0) { nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
    D.P = D.np + D.no; D.n = n; D.ld = (n + 1) & ~1;
    nla_evaluator_resolve(&D.ev, opt, f, f_data);
    D.obj = D.ev.kind == NLA_EVAL_DEVICE ? D.ev.obj : (D.ev.kind == NLA_EVAL_USER ? -2 : -1);
    if (opt && nlopt_get_param(opt, "amd_host_eval", 0) != 0) D.obj = -1;
    if ((uint64_t) D.no * (uint64_t) n >= (1ULL << 31)) { nla_stop_msg(stop, "nlopt_amd: ESCH with offspring x dimension >= 2^31 is not supported"); return NLOPT_INVALID_ARGS; }
    total = (int) (((unsigned) D.no * (unsigned) n) / 10);                          /* esch.c:207 */
    if (total < 1) total = 1;

    D.st = nla_stream_create();
    if (!D.st || !(D.mts = nla_mtstream_create(D.st))) { nla_stop_msg(stop, "nlopt_amd: could not create the device stream / generator state"); efree(&D); return NLOPT_OUT_OF_MEMORY; }
    D.sscratch_bytes = nla_esch_sort_scratch_bytes(D.P);
    D.d_lb = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.ld);
    D.d_ub = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.ld);
    D.d_R = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.P * (size_t) D.ld);
    D.d_G = (double *) nla_dev_malloc(sizeof(double) * (size_t) (D.np > D.no ? D.np : D.no) * (size_t) D.ld);
    D.d_F = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.P);
    D.d_fit[0] = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.P);
    D.d_fit[1] = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.P);
    D.d_slot[0] = (int32_t *) nla_dev_malloc(sizeof(int32_t) * (size_t) D.P);
    D.d_slot[1] = (int32_t *) nla_dev_malloc(sizeof(int32_t) * (size_t) D.P);
    D.d_v = (double *) nla_dev_malloc(sizeof(double) * (size_t) D.P * (size_t) n);
    D.d_vatt = (int64_t *) nla_dev_malloc(sizeof(int64_t) * (size_t) D.P * (size_t) n);
    D.d_last = (int32_t *) nla_dev_malloc(sizeof(int32_t) * (size_t) D.no * (size_t) n);
    D.d_out = (int64_t *) nla_dev_malloc(2 * sizeof(int64_t));
    D.d_sscratch = nla_dev_malloc(D.sscratch_bytes);
    D.h_fit = (double *) nla_host_malloc(sizeof(double) * (size_t) D.P);
    D.h_G = (double *) nla_host_malloc(sizeof(double) * (D.obj != -1 ? (size_t) D.ld : (size_t) (D.np > D.no ? D.np : D.no) * (size_t) D.ld));
    if (!D.d_lb || !D.d_ub || !D.d_R || !D.d_G || !D.d_F || !D.d_fit[0] || !D.d_fit[1] || !D.d_slot[0] || !D.d_slot[1] || !D.d_v || !D.d_vatt ||
        !D.d_last || !D.d_out || !D.d_sscratch || !D.h_fit || !D.h_G) {
        nla_stop_msg(stop, "nlopt_amd: could not create the ESCH device state (out of device memory?)");
        efree(&D);
        return NLOPT_OUT_OF_MEMORY;
    }
#define DEVFAIL() do { nla_stop_msg(stop, "device engine: %s", D.err); ret = NLOPT_FAILURE; goto done; } while (0)
    {
        int32_t *ident = (int32_t *) malloc(sizeof(int32_t) * (size_t) D.P);
        if (!ident) { nla_stop_msg(stop, "nlopt_amd: out of memory"); efree(&D); return NLOPT_OUT_OF_MEMORY; }
        for (i = 0; i < D.P; ++i) ident[i] = (int32_t) i;
        if (nla_memcpy_h2d(D.d_lb, lb, sizeof(double) * (size_t) n, D.st) || nla_memcpy_h2d(D.d_ub, ub, sizeof(double) * (size_t) n, D.st) ||
            nla_memcpy_h2d(D.d_slot[0], ident, sizeof(int32_t) * (size_t) D.P, D.st) || nla_stream_sync(D.st)) {
            free(ident); snprintf(D.err, sizeof D.err, "upload failed"); DEVFAIL();
        }
        free(ident);
    }
    if (init_rows(&D, x)) DEVFAIL();

    /* best point / stop tests of one evaluated candidate, esch.c:173-182 = :227-237 */
#define AFTER_EVAL(idx, knd, off) do { \
        if (D.obj == -1) D.h_fit[idx] = f((unsigned) n, D.h_G + (size_t) ((idx) - (off)) * (size_t) D.ld, NULL, f_data); \
        const double fv_ = D.h_fit[idx]; \
        ++*stop->nevals_p; \
        if (st) { if ((knd) == 0) ++st->evals_init; else ++st->evals_trial; } \
        if (opt && opt->trace) { \
            if (opt->trace_len < opt->trace_cap) { nlopt_amd_trace_rec *r_ = opt->trace + opt->trace_len; r_->f = fv_; r_->row = (idx); r_->kind = (knd); r_->accepted = 0; } \
            ++opt->trace_len; } \
        if (*minf > fv_) { *minf = fv_; kbest = (idx); } \
        if (nla_stop_forced(stop)) ret = NLOPT_FORCED_STOP; \
        else if (*minf < stop->minf_max) ret = NLOPT_MINF_MAX_REACHED; \
        else if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED; \
        else if (nla_stop_time(stop)) ret = NLOPT_MAXTIME_REACHED; \
    } while (0)
    /* memcpy(x, best row): once per pass over a population — rows do not change while they are being evaluated */
#define FETCH_BEST() do { if (kbest >= 0) { \
        int32_t s_; \
        if (nla_memcpy_d2h(&s_, D.d_slot[cur] + kbest, sizeof s_, D.st) || nla_stream_sync(D.st) || \
            nla_memcpy_d2h(x, D.d_R + (size_t) s_ * (size_t) D.ld, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) { \
            snprintf(D.err, sizeof D.err, "best-point read-back failed"); DEVFAIL(); } } } while (0)

    kbest = -1;
    if (evaluate(&D, cur, 0, D.np)) DEVFAIL();
    for (i = 0; i < D.np && ret == NLOPT_SUCCESS; ++i) AFTER_EVAL(i, 0, 0);
    FETCH_BEST();
    if (nla_memcpy_h2d(D.d_fit[cur], D.h_fit, sizeof(double) * (size_t) D.np, D.st)) { snprintf(D.err, sizeof D.err, "upload failed"); DEVFAIL(); }

    while (ret == NLOPT_SUCCESS) {                             /* one generation (esch.c:187-251) */
        double t0 = nla_seconds();
        int64_t out[2] = { 0, 0 };
        size_t M = (size_t) ((double) total * 4.6) + 8192;     /* expected 2 + 2/0.874 = 4.29 words per mutation step */
        int tries;
        if (opt && opt->progress) opt->progress(opt->progress_data, st ? (long) st->generations : 0, (long) *stop->nevals_p);
        /* crossover */
        if (need_words(&D, 3 * (size_t) D.no > M ? 3 * (size_t) D.no : M)) DEVFAIL();
        if (nla_mtstream_fill(D.mts, D.words_used, 3ULL * (uint64_t) D.no, D.d_words)) { snprintf(D.err, sizeof D.err, "MT stream fill failed"); DEVFAIL(); }
        if (nla_k_esch_crossover(n, D.ld, D.np, D.no, D.d_words, D.d_slot[cur], D.d_R, D.st)) { snprintf(D.err, sizeof D.err, "crossover launch failed"); DEVFAIL(); }
        D.words_used += 3ULL * (uint64_t) D.no;
        /* point mutations: the chain must fit into the generated segment */
        for (tries = 0;; ++tries) {
            if (need_words(&D, M)) DEVFAIL();
            if (nla_esch_mut_scratch_bytes((int64_t) M) > D.mscratch_bytes) {
                nla_dev_free(D.d_mscratch);
                D.mscratch_bytes = nla_esch_mut_scratch_bytes((int64_t) M) * 2;
                D.d_mscratch = nla_dev_malloc(D.mscratch_bytes);
                if (!D.d_mscratch) { D.mscratch_bytes = 0; snprintf(D.err, sizeof D.err, "out of device memory (mutation scratch)"); DEVFAIL(); }
            }
            if (nla_mtstream_fill(D.mts, D.words_used, (uint64_t) M, D.d_words)) { snprintf(D.err, sizeof D.err, "MT stream fill failed"); DEVFAIL(); }
            if (nla_k_esch_mutate(D.d_words, (int64_t) M, total, n, D.ld, D.np, D.no, D.d_lb, D.d_ub, D.d_slot[cur], D.d_R, D.d_last, D.d_mscratch,
                                  D.d_out, D.st) ||
                nla_memcpy_d2h(out, D.d_out, sizeof out, D.st) || nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "mutation pass failed"); DEVFAIL(); }
            if (out[0] >= total) break;
            /* the segment ended before the last step did: nothing beyond the first out[0] steps was applied wrongly — later
             * steps only overwrite; redo the whole pass on a longer segment (rows touched so far get the same values again) */
            if (tries >= 6) { snprintf(D.err, sizeof D.err, "mutation chain did not fit the stream segment"); DEVFAIL(); }
            M *= 2;
        }
        D.words_used += (uint64_t) out[1];
        if (st) st->t_evolve_s += nla_seconds() - t0;
        /* offspring evaluation */
        t0 = nla_seconds();
        kbest = -1;
        if (evaluate(&D, cur, D.np, D.no)) DEVFAIL();
        if (st) st->t_eval_s += nla_seconds() - t0;
        for (i = 0; i < D.no && ret == NLOPT_SUCCESS; ++i) AFTER_EVAL(D.np + i, 1, D.np);
        FETCH_BEST();
        if (ret != NLOPT_SUCCESS) break;
        /* selection */
        t0 = nla_seconds();
        {
            int has_nan = 0;
            for (i = 0; i < D.P && !has_nan; ++i) has_nan = D.h_fit[i] != D.h_fit[i];
            if (has_nan) { if (select_with_nan(&D, cur)) { snprintf(D.err, sizeof D.err, "selection failed"); DEVFAIL(); } }
            else
        if (nla_memcpy_h2d(D.d_fit[cur] + D.np, D.h_fit + D.np, sizeof(double) * (size_t) D.no, D.st) ||
            nla_k_esch_select(D.P, D.d_slot[cur], D.d_fit[cur], D.d_slot[cur ^ 1], D.d_fit[cur ^ 1], D.d_sscratch, D.sscratch_bytes, D.st) ||
            nla_memcpy_d2h(D.h_fit, D.d_fit[cur ^ 1], sizeof(double) * (size_t) D.P, D.st) || nla_stream_sync(D.st)) { snprintf(D.err, sizeof D.err, "selection failed"); DEVFAIL(); }
        }
        cur ^= 1;
        if (st) { st->t_rank_s += nla_seconds() - t0; ++st->generations; st->mt_words = D.words_used; }
    }
done:
    if (st) st->mt_words = D.words_used;
    efree(&D);
    return ret;
}
