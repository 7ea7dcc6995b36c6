/* crs_driver.c — CRS2_LM, the algorithm side: speculate on the device, commit in program order.
 *
 * The reference (src/algs/crs/crs.c) is one serial chain: build ONE trial point from the current
 * best + n random rows, evaluate it, accept it over the current worst or reject it, maybe mutate,
 * repeat (crs_trial :125-156, crs_minimize :250-270).  Its only data parallelism is inside a
 * trial.  To feed a GPU we exploit two facts:
 *   (i)  every reflection trial and every mutation consumes exactly one 2n-word block of the MT
 *        stream, so the random content of block m is known before its role is;
 *   (ii) a block is a mutation iff the previous block was a rejected reflection (:129-151), and
 *        acceptance is the common case, so "every block is a reflection trial" is a good guess.
 * Each round the engine computes K consecutive blocks as reflection trials against one snapshot
 * of the population (plus, for each, the mutation that would follow its rejection), and this file
 * then replays the reference's accept/reject chain over the results *in block order*: a
 * speculative trial is used only if nothing it read has changed since the snapshot —
 *   (a) its block still has the reflection role,
 *   (b) the best row is still the same row,
 *   (c) none of the rows it sampled was overwritten by an earlier commit of the same round.
 * Since every commit overwrites the then-worst row, the rows overwritten so far always form a
 * prefix of the round's initial worst-first list W; (c) is therefore one integer compare against
 * the smallest W-rank the trial touched (computed on the device).  The first unusable slot ends
 * the round; the next round starts at that block.  The result is the reference's exact sequence
 * of (candidate, accept/reject, replaced row), including its quirks: maxeval is only tested
 * after a rejection (:136-137), ftol compares successive bests (:256), ties in f break by row
 * index (:51-56, here in key_less), and a reached tolerance is overridden by MAXEVAL (:263-268).
 */
#include "nla_internal.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- ordered set over rows keyed (f, row): max-heap for the worst + tracked argmin ----------- */
typedef struct {
    const double *F;
    int64_t *heap, nheap, best;
    int64_t *cand;      /* scratch for top-k */
} ordset;

static inline int key_less(const double *F, int64_t a, int64_t b)       /* crs_compare, crs.c:51-56 */
{
    if (F[a] < F[b]) return 1;
    if (F[a] > F[b]) return 0;
    return a < b;
}

static void os_push(ordset *s, int64_t row)
{
    int64_t pos = s->nheap++;
    while (pos > 0) {
        int64_t par = (pos - 1) / 2;
        if (!key_less(s->F, s->heap[par], row)) break;
        s->heap[pos] = s->heap[par];
        pos = par;
    }
    s->heap[pos] = row;
    if (s->nheap == 1 || key_less(s->F, row, s->best)) s->best = row;
}

static void os_top_changed(ordset *s)        /* key of heap[0] decreased: restore the heap */
{
    int64_t pos = 0, v = s->heap[0];
    for (;;) {
        int64_t c = 2 * pos + 1;
        if (c >= s->nheap) break;
        if (c + 1 < s->nheap && key_less(s->F, s->heap[c], s->heap[c + 1])) ++c;
        if (!key_less(s->F, v, s->heap[c])) break;
        s->heap[pos] = s->heap[c];
        pos = c;
    }
    s->heap[pos] = v;
}

/* the k largest rows, worst first, without disturbing the heap: best-first walk over heap nodes */
static int os_topk(ordset *s, int k, int64_t *out)
{
    int64_t *c = s->cand;      /* small max-heap of heap positions */
    int nc = 0, got = 0;
    if (s->nheap == 0) return 0;
    c[nc++] = 0;
    while (got < k && nc > 0) {
        int64_t p = c[0];
        out[got++] = s->heap[p];
        /* pop */
        int64_t lastp = c[--nc];
        if (nc > 0) {
            int pos = 0;
            for (;;) {
                int ch = 2 * pos + 1;
                if (ch >= nc) break;
                if (ch + 1 < nc && key_less(s->F, s->heap[c[ch]], s->heap[c[ch + 1]])) ++ch;
                if (!key_less(s->F, s->heap[lastp], s->heap[c[ch]])) break;
                c[pos] = c[ch];
                pos = ch;
            }
            c[pos] = lastp;
        }
        for (int64_t child = 2 * p + 1; child <= 2 * p + 2; ++child) {
            if (child >= s->nheap) break;
            int pos = nc++;
            while (pos > 0) {
                int par = (pos - 1) / 2;
                if (!key_less(s->F, s->heap[c[par]], s->heap[child])) break;
                c[pos] = c[par];
                pos = par;
            }
            c[pos] = child;
        }
    }
    return got;
}

/* ---------------------------------------------------------------------------------------------- */
typedef struct {
    const nla_crs_engine_ops *ops; void *e;
    const nla_crs_problem *pb;
    ordset os;
    double *F;
    double *x;            /* caller's x: previous best point (crs.c:247,258) */
    double *minf;
    double *xtmp;
    int need_x;
    nlopt_result ret;
} run_state;

static void trace_add(const nla_crs_problem *pb, double f, int64_t row, int kind, int accepted)
{
    if (!pb->trace || !pb->trace_len) return;
    if (*pb->trace_len < pb->trace_cap) {
        nlopt_amd_trace_rec *r = pb->trace + *pb->trace_len;
        r->f = f; r->row = row; r->kind = kind; r->accepted = accepted;
    }
    ++*pb->trace_len;
}

/* crs_minimize's bookkeeping after an accepted replacement (crs.c:252-269).  The accepted point
 * sits in speculation slot `slot` (buffer `kind`); it has not been written to the population yet. */
static int after_accept(run_state *rs, int slot, int kind)
{
    const nla_crs_problem *pb = rs->pb;
    nla_stopping *stop = pb->stop;
    const int64_t b = rs->os.best;
    nlopt_result ret = NLOPT_SUCCESS;
    if (rs->F[b] < *rs->minf) {
        if (rs->F[b] < stop->minf_max) ret = NLOPT_MINF_MAX_REACHED;
        else if (nla_stop_f(stop, rs->F[b], *rs->minf)) ret = NLOPT_FTOL_REACHED;
        else if (rs->need_x) {
            if (rs->ops->read_slot(rs->e, slot, kind, rs->xtmp)) return -1;
            if (nla_stop_x(stop, rs->xtmp, rs->x)) ret = NLOPT_XTOL_REACHED;
            memcpy(rs->x, rs->xtmp, sizeof(double) * (size_t) pb->n);
        }
        *rs->minf = rs->F[b];
    }
    if (ret != NLOPT_SUCCESS) {                 /* quirk kept: crs.c:263-268 */
        if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED;
        else if (nla_stop_time(stop)) ret = NLOPT_MAXTIME_REACHED;
    }
    rs->ret = ret;
    return 0;
}

/* ---- resumable run: begin (= crs_init), advance (= rounds of the trial loop), end -------------- */
struct nla_crs_session {
    run_state rs;
    nla_crs_problem pb;
    nlopt_result ret;
    uint64_t block, init_words;
    int Kmax, host_eval;
    double runlen;
    double *fT, *fM;
    int32_t *minhz, *cslot, *ckind;
    int64_t *W, *crow;
};

static void session_free(nla_crs_session *S)
{
    if (!S) return;
    free(S->rs.F); free(S->rs.os.heap); free(S->rs.os.cand); free(S->rs.xtmp);
    free(S->fT); free(S->fM); free(S->minhz); free(S->W); free(S->cslot); free(S->ckind); free(S->crow);
    free(S);
}

static void engine_failed(nla_crs_session *S)
{
    nla_stopping *stop = S->pb.stop;
    if (stop->stop_msg && S->rs.ops->last_error) nla_stop_msg(stop, "device engine: %s", S->rs.ops->last_error(S->rs.e));
    S->ret = NLOPT_FAILURE;
}

/* crs_init (crs.c:165-229) + the first assignments of crs_minimize (crs.c:246-248).  On return
 * *ret_out is NLOPT_SUCCESS if the trial loop may start, else the final result. */
nla_crs_session *nla_crs_begin(const nla_crs_engine_ops *ops, void *e, const nla_crs_problem *pb,
                               double *x, double *minf, nlopt_result *ret_out)
{
    const int n = pb->n;
    const int64_t N = pb->N;
    nla_stopping *stop = pb->stop;
    nlopt_amd_stats *st = pb->stats;
    nla_crs_session *S = (nla_crs_session *) calloc(1, sizeof *S);
    run_state *rs;
    nlopt_result ret = NLOPT_SUCCESS;
    int64_t rows_done = 0, i;
    double t0 = nla_seconds();
    if (!S) { *ret_out = NLOPT_OUT_OF_MEMORY; return NULL; }
    rs = &S->rs;
    S->pb = *pb;
    S->host_eval = pb->obj < 0;
    S->Kmax = pb->max_spec > 0 ? pb->max_spec : 1024;
    if (S->Kmax > 1024) S->Kmax = 1024;
    if (S->host_eval) S->Kmax = 1;
    S->runlen = 4.0;
    rs->ops = ops; rs->e = e; rs->pb = &S->pb; rs->x = x; rs->minf = minf;
    rs->need_x = (stop->xtol_rel > 0 || stop->xtol_abs != NULL);

    rs->F = (double *) malloc(sizeof(double) * (size_t) N);
    rs->os.heap = (int64_t *) malloc(sizeof(int64_t) * (size_t) N);
    rs->os.cand = (int64_t *) malloc(sizeof(int64_t) * (size_t) (2 * S->Kmax + 8));
    rs->xtmp = (double *) malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));
    S->fT = (double *) malloc(sizeof(double) * (size_t) S->Kmax);
    S->fM = (double *) malloc(sizeof(double) * (size_t) S->Kmax);
    S->minhz = (int32_t *) malloc(sizeof(int32_t) * (size_t) S->Kmax);
    S->W = (int64_t *) malloc(sizeof(int64_t) * (size_t) S->Kmax);
    S->cslot = (int32_t *) malloc(sizeof(int32_t) * (size_t) S->Kmax);
    S->ckind = (int32_t *) malloc(sizeof(int32_t) * (size_t) S->Kmax);
    S->crow = (int64_t *) malloc(sizeof(int64_t) * (size_t) S->Kmax);
    if (!rs->F || !rs->os.heap || !rs->os.cand || !rs->xtmp || !S->fT || !S->fM || !S->minhz || !S->W || !S->cslot ||
        !S->ckind || !S->crow) {
        session_free(S);
        *ret_out = NLOPT_OUT_OF_MEMORY;
        return NULL;
    }
    rs->os.F = rs->F;

    /* the device generates and (if it can) evaluates all N rows; the reference's stop tests run
     * after *every* evaluation, so replay them in row order and forget rows past the first stop */
    if (ops->init_population(e, x, rs->F)) { engine_failed(S); *ret_out = S->ret; return S; }
    for (i = 0; i < N && ret == NLOPT_SUCCESS; ++i) {
        if (S->host_eval) {
            if (i == 0) rs->F[0] = pb->f((unsigned) n, x, NULL, pb->f_data);
            else {
                if (ops->read_row(e, i, rs->xtmp)) { engine_failed(S); *ret_out = S->ret; return S; }
                rs->F[i] = pb->f((unsigned) n, rs->xtmp, NULL, pb->f_data);
            }
        }
        ++*stop->nevals_p;
        if (st) ++st->evals_init;
        os_push(&rs->os, i);
        trace_add(pb, rs->F[i], i, 0, 1);
        rows_done = i + 1;
        if (rs->F[i] < stop->minf_max) ret = NLOPT_MINF_MAX_REACHED;
        else if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED;
        else if (nla_stop_time(stop)) ret = NLOPT_MAXTIME_REACHED;
    }
    S->init_words = 2ULL * (uint64_t) n * (uint64_t) (rows_done - 1);
    *minf = rs->F[rs->os.best];                               /* crs.c:246-248 */
    if (ops->read_row(e, rs->os.best, x)) { engine_failed(S); *ret_out = S->ret; return S; }
    if (st) st->t_init_s = nla_seconds() - t0;
    S->ret = ret;
    *ret_out = ret;
    return S;
}

/* Run rounds of the trial loop (crs.c:250-270) until the algorithm stops or at least
 * `eval_budget` more evaluations have been made (<= 0: no budget).  Pausing between rounds does
 * not change the sequence: a round boundary is only a speculation boundary. */
nlopt_result nla_crs_advance(nla_crs_session *S, int64_t eval_budget)
{
    run_state *rs = &S->rs;
    const nla_crs_engine_ops *ops = rs->ops;
    void *e = rs->e;
    const nla_crs_problem *pb = &S->pb;
    nla_stopping *stop = pb->stop;
    nlopt_amd_stats *st = pb->stats;
    const int n = pb->n, host_eval = S->host_eval;
    const int64_t N = pb->N;
    double *fT = S->fT, *fM = S->fM;
    int32_t *minhz = S->minhz, *cslot = S->cslot, *ckind = S->ckind;
    int64_t *W = S->W, *crow = S->crow;
    nlopt_result ret = S->ret;
    const int evals_at_entry = *stop->nevals_p;
    double t0 = nla_seconds();

    while (ret == NLOPT_SUCCESS) {
        int K, nW, j = 0, c = 0, ncommit = 0, best_changed = 0, cap;
        if (eval_budget > 0 && (int64_t) (*stop->nevals_p - evals_at_entry) >= eval_budget) break;
        cap = ops->max_slots(e, S->block);
        if (cap <= 0) { engine_failed(S); return S->ret; }
        K = (int) ceil(1.25 * S->runlen) + 1;
        if (K > S->Kmax) K = S->Kmax;
        if (K > cap) K = cap;
        if (K < 1) K = 1;
        nW = K < N ? K : (int) N;
        nW = os_topk(&rs->os, nW, W);
        if (ops->speculate(e, S->block, K, rs->os.best, W, nW, fT, fM, minhz)) { engine_failed(S); return S->ret; }
        if (st) { ++st->rounds; st->slots_launched += (uint64_t) K; }

        while (j < K && ret == NLOPT_SUCCESS) {
            int64_t worst;
            int kind = 1, accepted = 0;
            double fcand;
            if (best_changed) { if (st) st->slots_newbest += (uint64_t) (K - j); break; }
            if (minhz[j] < c) { if (st) ++st->slots_invalid; break; }
            if (st) ++st->slots_used;
            worst = rs->os.heap[0];
            /* reflection trial of block+j */
            if (host_eval) {
                if (ops->read_slot(e, j, 1, rs->xtmp)) { engine_failed(S); return S->ret; }
                fT[j] = pb->f((unsigned) n, rs->xtmp, NULL, pb->f_data);
            }
            fcand = fT[j];
            ++*stop->nevals_p;
            if (st) ++st->evals_trial;
            if (nla_stop_forced(stop)) { trace_add(pb, fcand, -1, 1, 0); ret = NLOPT_FORCED_STOP; ++j; break; }
            if (fcand < rs->F[worst]) accepted = 1;
            else {
                trace_add(pb, fcand, -1, 1, 0);
                if (nla_stop_evals(stop)) { ret = NLOPT_MAXEVAL_REACHED; ++j; break; }   /* only after a rejection */
                if (nla_stop_time(stop)) { ret = NLOPT_MAXTIME_REACHED; ++j; break; }
                /* local mutation: consumes block+j+1 (crs.c:139-146) */
                kind = 2;
                if (host_eval) {
                    if (ops->mutate_slot(e, j, S->block + (uint64_t) j + 1, rs->os.best) ||
                        ops->read_slot(e, j, 1, rs->xtmp)) { engine_failed(S); return S->ret; }
                    fM[j] = pb->f((unsigned) n, rs->xtmp, NULL, pb->f_data);
                }
                fcand = fM[j];
                ++*stop->nevals_p;
                if (st) { ++st->evals_mutation; if (j + 1 < K) ++st->slots_role; }
                if (nla_stop_forced(stop)) { trace_add(pb, fcand, -1, 2, 0); ret = NLOPT_FORCED_STOP; j += 2; break; }
                if (fcand < rs->F[worst]) accepted = 1;
                else {
                    trace_add(pb, fcand, -1, 2, 0);
                    if (nla_stop_evals(stop)) { ret = NLOPT_MAXEVAL_REACHED; j += 2; break; }
                    if (nla_stop_time(stop)) { ret = NLOPT_MAXTIME_REACHED; j += 2; break; }
                }
            }
            if (accepted) {
                /* memcpy(worst->k, d->p) + resort (crs.c:153-154); the row write is deferred */
                if (c < nW && W[c] == worst) ++c;            /* else: a row of the prefix, again */
                rs->F[worst] = fcand;
                os_top_changed(&rs->os);
                trace_add(pb, fcand, worst, kind, 1);
                if (st) ++st->accepted;
                cslot[ncommit] = j; ckind[ncommit] = host_eval ? 1 : kind; crow[ncommit] = worst; ++ncommit;
                if (key_less(rs->F, worst, rs->os.best)) { rs->os.best = worst; best_changed = 1; }
                if (after_accept(rs, j, host_eval ? 1 : kind)) { engine_failed(S); return S->ret; }
                ret = rs->ret;
            }
            j += (kind == 2) ? 2 : 1;
        }
        S->block += (uint64_t) j;
        {   /* adapt the speculation depth to the observed usable run length */
            double obs = (j >= K) ? 2.0 * K : (double) j;
            S->runlen = 0.7 * S->runlen + 0.3 * obs;
            if (S->runlen < 1.0) S->runlen = 1.0;
        }
        if (ncommit > 0) {
            /* a row replaced twice in one round keeps only its last content */
            int k, m = 0;
            for (k = 0; k < ncommit; ++k) {
                int later = 0, q;
                for (q = k + 1; q < ncommit; ++q) if (crow[q] == crow[k]) { later = 1; break; }
                if (!later) { cslot[m] = cslot[k]; ckind[m] = ckind[k]; crow[m] = crow[k]; ++m; }
            }
            if (ops->commit(e, m, cslot, ckind, crow)) { engine_failed(S); return S->ret; }
        }
    }
    if (st) st->t_trial_s += nla_seconds() - t0;
    S->ret = ret;
    return ret;
}

/* final x (crs.c:258 keeps x = best point), stream accounting, release */
nlopt_result nla_crs_end(nla_crs_session *S, uint64_t *words_used)
{
    nlopt_result ret;
    if (!S) return NLOPT_INVALID_ARGS;
    ret = S->ret;
    if (words_used) *words_used = S->init_words + 2ULL * (uint64_t) S->pb.n * S->block;
    if (ret != NLOPT_FAILURE && S->rs.os.nheap > 0 && S->rs.ops->read_row(S->rs.e, S->rs.os.best, S->rs.x)) {
        engine_failed(S);
        ret = S->ret;
    }
    session_free(S);
    return ret;
}

nlopt_result nla_crs_run(const nla_crs_engine_ops *ops, void *e, const nla_crs_problem *pb,
                         double *x, double *minf, uint64_t *words_used)
{
    nlopt_result ret;
    nla_crs_session *S = nla_crs_begin(ops, e, pb, x, minf, &ret);
    if (!S) return ret;
    if (ret == NLOPT_SUCCESS) nla_crs_advance(S, 0);
    return nla_crs_end(S, words_used);
}
