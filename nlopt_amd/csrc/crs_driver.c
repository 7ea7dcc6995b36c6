/* crs_driver.c — CRS2_LM, the algorithm side: a window of stream blocks in flight on the device,
 * committed in program order on the host.
 *
 * The reference (src/algs/crs/crs.c) is one serial chain: build ONE trial point from the current
 * best + n random rows, evaluate it, accept it over the current worst or reject it, maybe mutate,
 * repeat (crs_trial :125-156, crs_minimize :250-270).  Its only data parallelism is inside a
 * trial.  To feed a GPU we exploit:
 *   (i)   every reflection trial and every mutation consumes exactly one 2n-word block of the MT
 *         stream, so the random content of block m is known before its role is;
 *   (ii)  a block is a mutation iff the previous block was a rejected reflection (:129-151), and
 *         acceptance is the common case, so "every block is a reflection trial" is a good guess;
 *   (iii) a trial's gather-sum visits its n rows in ascending row order, and between "now" and the
 *         trial's own turn the only rows that can change are the ones the intervening blocks may
 *         commit to: each block commits at most once and every commit overwrites the then-worst
 *         row, so after d more blocks the overwritten rows are a subset of the current d worst.
 * The device keeps a window of K consecutive blocks in flight, each as a *resumable* gather-sum
 * (picks summed so far + accumulator).  One pass (ops->advance) pushes every slot forward until
 * its next pick is one of the rows in its hazard prefix W[0..d) — never across it — so no row is
 * read before its content is final for that slot, no byte is read twice and no finished trial is
 * ever thrown away because of a write hazard.  The front slot has d = 0 and always finishes.  This
 * file then replays the reference's accept/reject chain over the finished slots at the window
 * front *in block order*, until it meets an unfinished one; commits are written back and the next
 * pass recomputes W from the updated order.  What is still discarded (and counted in the stats):
 *   (a) a slot whose block turned out to be a mutation block (its predecessor was rejected),
 *   (b) everything in flight when the best point changes (the sum starts from the best row).
 * DEVICE-RESOLVED WINDOWS (the engine's chain op, hip/crs_chain.hip; default where the engine has it): the conservative
 * passes above let the chain consume only the first ~sqrt(2N/n) blocks of a window.  The chain kernel instead computes every
 * slot of the window to the end in one launch and resolves the dependences itself — it evaluates each finished trial, replays
 * crs_trial's accept / reject decisions in block order on the window's worst rows, and a slot whose pick is one of those
 * rows reads what the row holds AT ITS TURN: the point of the block that overwrote it, or the row as it is.  This file stays
 * the authority: per slot it gets one record per such pick (row, what was read: the row itself / block b's trial point /
 * block b's mutation) and consumes the slot only if, at that moment of the replayed chain, the row's last writer is exactly
 * that (lastw[]); otherwise the window ends there and the slot is recomputed at the front of the next one (stats.slots_invalid).
 * What the chain consumes is therefore always a trial point built from the rows the reference reads.
 * The result is the reference's exact sequence of (candidate, accept/reject, replaced row),
 * including its quirks: maxeval is only tested after a rejection (:136-137), ftol compares
 * successive bests (:256), ties in f break by row index (:51-56, here in key_less), and a reached
 * tolerance is overridden by MAXEVAL (:263-268).
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
    const nla_stopping *sp;   /* what the clock / force_stop tests look at: pb->stop, or the view all ranks agreed on for this pass */
    int64_t xrow;         /* row whose content is the x of the last STRICT improvement of minf (crs.c:253-259), -1: rs->x holds it */
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
 * sits in the slot of stream block `block` (buffer `kind`); it has not been written to the
 * population yet. */
static int after_accept(run_state *rs, uint64_t block, int kind)
{
    const nla_crs_problem *pb = rs->pb;
    nla_stopping *stop = pb->stop;
    const int64_t b = rs->os.best;
    nlopt_result ret = NLOPT_SUCCESS;
    int have_x = 0;
    if (rs->F[b] < *rs->minf) {
        if (rs->F[b] < stop->minf_max) ret = NLOPT_MINF_MAX_REACHED;
        else if (nla_stop_f(stop, rs->F[b], *rs->minf)) ret = NLOPT_FTOL_REACHED;
        else if (rs->need_x) {
            if (rs->ops->read_slot(rs->e, block, kind, rs->xtmp)) return -1;
            if (nla_stop_x(stop, rs->xtmp, rs->x)) ret = NLOPT_XTOL_REACHED;
            memcpy(rs->x, rs->xtmp, sizeof(double) * (size_t) pb->n);
            have_x = 1;
        }
        rs->xrow = have_x ? -1 : b;             /* memcpy(x, best->k + 1) deferred: the row keeps this content while it is the best */
        *rs->minf = rs->F[b];
    }
    if (ret != NLOPT_SUCCESS) {                 /* quirk kept: crs.c:263-268 */
        if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED;
        else if (nla_stop_time(rs->sp)) ret = NLOPT_MAXTIME_REACHED;
    }
    rs->ret = ret;
    return 0;
}

/* ---- resumable run: begin (= crs_init), advance (= rounds of the trial loop), end -------------- */
#define TRING 2048                 /* host mirror of per-slot progress, indexed by block % TRING */
#define FWCAP 48                   /* records per slot the chain kernel keeps; a slot with more is recomputed at the front */

struct nla_crs_session {
    run_state rs;
    nla_crs_problem pb;
    nlopt_result ret;
    uint64_t block;                /* first stream block not yet consumed by the chain */
    uint64_t fresh_from;           /* blocks >= this have no device state yet */
    uint64_t init_words;
    int Kmax, host_eval;
    double kmult;                  /* window = kmult x (blocks consumed per pass, smoothed) + 4 */
    double runlen;
    nla_crs_slot_status *status;
    int32_t *tprev;                /* picks already summed, per in-flight block (stats only) */
    uint64_t *cblock;
    int32_t *ckind;
    int64_t *W, *crow;
    /* device-resolved windows */
    int forward;
    double *Wf;                    /* f of the rows W */
    uint32_t *fwcnt, *fwrec;       /* Kmax, Kmax x FWCAP: the last window's records */
    uint64_t *lastw;               /* N: (block + 1) << 1 | (reflection trial ? 1 : 0) of the row's last writer, 0 = initial row */
    nla_stopping view; int agreed_force;      /* the stop view all ranks agreed on (collective passes only) */
};

static void session_free(nla_crs_session *S)
{
    if (!S) return;
    free(S->rs.F); free(S->rs.os.heap); free(S->rs.os.cand); free(S->rs.xtmp);
    free(S->status); free(S->tprev); free(S->W); free(S->cblock); free(S->ckind); free(S->crow);
    free(S->Wf); free(S->fwcnt); free(S->fwrec); free(S->lastw);
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
    S->host_eval = pb->obj == -1;               /* (-2: a user-supplied device kernel, evaluated by the engine) */
    S->Kmax = pb->max_spec > 0 ? pb->max_spec : 256;
    /* measured (MI355X, pop 1e5): the chain consumes the same number of blocks per pass whether the window is 1.5x or 12x
     * that number wide — the slots behind only add workgroups, status records and hazard rows: 1.5 is as fast as 3 at
     * n = 4096 and 3 % / 10 % faster at n = 512 / 64; 12 is 15 % slower */
    S->kmult = pb->window_factor > 0 ? pb->window_factor : 1.5;
    if (S->Kmax > 1024) S->Kmax = 1024;
    if (S->host_eval) S->Kmax = 1;
    S->forward = pb->forward && ops->chain && pb->obj >= 0 && S->Kmax > 1;
    if (S->forward && S->Kmax > 256) S->Kmax = 256;
    /* measured (MI355X, n = 4096, N = 1e5, bench.py --max-spec): 39.9 k evals/s at 40 slots per launch, 40.7 k at 48, 41.6 k at 64,
     * 42.5 k at 96, 43.0 k at 128, 43.1 k at 160 / 192, 42.3 k at 256 — a longer window amortises the launch ramp and the 53 us of
     * host turnaround between windows; past 128 the slots recomputed because a value landed among the worst rows twice eat the gain */
    if (S->forward && pb->max_spec <= 0 && S->Kmax > 128) S->Kmax = 128;
    S->runlen = 4.0;
    rs->ops = ops; rs->e = e; rs->pb = &S->pb; rs->x = x; rs->minf = minf;
    rs->need_x = (stop->xtol_rel > 0 || stop->xtol_abs != NULL);

    rs->F = (double *) malloc(sizeof(double) * (size_t) N);
    rs->os.heap = (int64_t *) malloc(sizeof(int64_t) * (size_t) N);
    rs->os.cand = (int64_t *) malloc(sizeof(int64_t) * (size_t) (2 * S->Kmax + 8));
    rs->xtmp = (double *) malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));
    S->status = (nla_crs_slot_status *) malloc(sizeof(nla_crs_slot_status) * (size_t) S->Kmax);
    S->tprev = (int32_t *) calloc(TRING, sizeof(int32_t));
    S->W = (int64_t *) malloc(sizeof(int64_t) * (size_t) S->Kmax);
    S->cblock = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) S->Kmax);
    S->ckind = (int32_t *) malloc(sizeof(int32_t) * (size_t) S->Kmax);
    S->crow = (int64_t *) malloc(sizeof(int64_t) * (size_t) S->Kmax);
    if (S->forward) {
        S->Wf = (double *) malloc(sizeof(double) * (size_t) S->Kmax);
        S->fwcnt = (uint32_t *) calloc((size_t) S->Kmax, sizeof(uint32_t));
        S->fwrec = (uint32_t *) calloc((size_t) S->Kmax * FWCAP, sizeof(uint32_t));
        S->lastw = (uint64_t *) calloc((size_t) N, sizeof(uint64_t));
    }
    if (!rs->F || !rs->os.heap || !rs->os.cand || !rs->xtmp || !S->status || !S->tprev || !S->W || !S->cblock ||
        !S->ckind || !S->crow || (S->forward && (!S->Wf || !S->fwcnt || !S->fwrec || !S->lastw))) {
        session_free(S);
        *ret_out = NLOPT_OUT_OF_MEMORY;
        return NULL;
    }
    rs->os.F = rs->F;

    /* the device generates and (if it can) evaluates all N rows; the reference's stop tests run
     * after *every* evaluation, so replay them in row order and forget rows past the first stop */
    rs->sp = stop;
    if (ops->init_population(e, x, rs->F)) { engine_failed(S); *ret_out = S->ret; return S; }
    if (pb->comm) {            /* collective passes follow: every rank must leave the replay below at the same row */
        rs->sp = nla_comm_agree_stop(pb->comm, stop, &S->view, &S->agreed_force);
        if (!rs->sp) { nla_stop_msg(stop, "stop agreement failed: %s", nlopt_amd_comm_error(pb->comm)); S->ret = NLOPT_FAILURE; *ret_out = S->ret; return S; }
    }
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
        else if (nla_stop_time(rs->sp)) ret = NLOPT_MAXTIME_REACHED;
    }
    S->init_words = 2ULL * (uint64_t) n * (uint64_t) (rows_done - 1);
    *minf = rs->F[rs->os.best];                               /* crs.c:246-248 */
    if (ops->read_row(e, rs->os.best, x)) { engine_failed(S); *ret_out = S->ret; return S; }
    rs->xrow = -1;
    if (st) st->t_init_s = nla_seconds() - t0;
    S->ret = ret;
    *ret_out = ret;
    return S;
}

/* Run passes of the trial loop (crs.c:250-270) until the algorithm stops or at least
 * `eval_budget` more evaluations have been made (<= 0: no budget).  Pausing between passes does
 * not change the sequence: the in-flight slots simply stay in flight. */
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
    nla_crs_slot_status *status = S->status;
    uint64_t *cblock = S->cblock;
    int32_t *ckind = S->ckind;
    int64_t *W = S->W, *crow = S->crow;
    nlopt_result ret = S->ret;
    const int evals_at_entry = *stop->nevals_p;
    double t0 = nla_seconds();

    while (ret == NLOPT_SUCCESS) {
        int K, nW, j = 0, ncommit = 0, best_changed = 0, cap, a, timed_pass = 1;
        uint64_t wend;
        if (eval_budget > 0 && (int64_t) (*stop->nevals_p - evals_at_entry) >= eval_budget) break;
        if (pb->comm) {
            /* the clock and the force_stop flag are decided once per pass, by all ranks together: this rank's view travels with the
             * pass's candidates and comes back OR-ed with the status (no collective of its own); an engine without that: comm.c */
            if (ops->stop_flags_in && ops->stop_flags_out) ops->stop_flags_in(e, nla_stop_forced(stop), nla_stop_time(stop));
            else {
                rs->sp = nla_comm_agree_stop(pb->comm, stop, &S->view, &S->agreed_force);
                if (!rs->sp) { nla_stop_msg(stop, "stop agreement failed: %s", nlopt_amd_comm_error(pb->comm)); S->ret = NLOPT_FAILURE; return S->ret; }
            }
        }
        cap = ops->max_slots(e, S->block);
        if (cap <= 0) { engine_failed(S); return S->ret; }
        K = (int) ceil(S->kmult * S->runlen) + 4;
        if (K > S->Kmax) K = S->Kmax;
        if (K > cap) K = cap;
        if (K < 1) K = 1;
        /* never shrink the window below what is already in flight (their state would be lost) */
        if (!S->forward && S->fresh_from > S->block && (uint64_t) K < S->fresh_from - S->block && S->fresh_from - S->block <= (uint64_t) cap &&
            S->fresh_from - S->block <= (uint64_t) S->Kmax)
            K = (int) (S->fresh_from - S->block);
        nW = K < N ? K : (int) N;
        nW = os_topk(&rs->os, nW, W);
        const double t_eng0 = st ? nla_seconds() : 0.;
        const uint64_t gl0 = st ? st->gather_launches : 0;
        if (S->forward) {
            /* every slot of the window is computed in this launch; what an earlier window left unconsumed is dropped */
            if (st && S->fresh_from > S->block) st->slots_invalid += S->fresh_from - S->block;
            S->fresh_from = S->block;
            for (a = 0; a < nW; ++a) S->Wf[a] = rs->F[W[a]];
            if (ops->chain(e, S->block, K, rs->os.best, rs->F[rs->os.best], W, S->Wf, nW, status, S->fwcnt, S->fwrec, FWCAP)) { engine_failed(S); return S->ret; }
        } else
        {
            if (!S->forward && ops->advance(e, S->block, K, S->fresh_from, rs->os.best, W, nW, status)) { engine_failed(S); return S->ret; }
            if (pb->comm && ops->stop_flags_in && ops->stop_flags_out) {
                int forced = 0, timed = 0;
                ops->stop_flags_out(e, &forced, &timed);
                nla_stop_view(stop, forced, timed, &S->view, &S->agreed_force);
                rs->sp = &S->view;
            }
        }
        timed_pass = !st || st->gather_launches != gl0;            /* the engine times the gather of every pass, or of a sample of them: bytes follow */
        wend = S->block + (uint64_t) K;
        const double t_walk0 = st ? nla_seconds() : 0.;
        if (st) {
            st->t_engine_s += t_walk0 - t_eng0;
            ++st->rounds;
            for (a = 0; a < K; ++a) {          /* algorithmic bytes this pass moved: 8n per row summed */
                const uint64_t b = S->block + (uint64_t) a;
                const int32_t told = b >= S->fresh_from ? 0 : S->tprev[b % TRING];
                const int32_t tnew = status[a].t;
                if (b >= S->fresh_from) ++st->slots_launched;
                if (tnew > told && timed_pass) st->gather_bytes += 8ULL * (uint64_t) n * (uint64_t) (tnew - told + (told == 0 ? 1 : 0));
                S->tprev[b % TRING] = tnew;
            }
        }
        if (wend > S->fresh_from) S->fresh_from = wend;

        while (j < K && ret == NLOPT_SUCCESS) {
            int64_t worst;
            int kind = 1, accepted = 0;
            double fcand;
            const uint64_t blk = S->block + (uint64_t) j;
            if (best_changed) break;
            if (status[j].t < n) break;             /* not finished yet: next pass */
            if (S->forward) {
                /* what this slot read for its picks among the window's worst rows must be what the chain, replayed up to
                 * here, says those rows hold: the row untouched since the window started, or last written by exactly the
                 * block and point (trial / mutation) the slot took it from */
                const uint32_t cnt = S->fwcnt[j];
                int ok = cnt <= FWCAP;
                uint32_t c;
                for (c = 0; ok && c < cnt; ++c) {
                    const uint32_t rec = S->fwrec[(size_t) j * FWCAP + c];
                    const int64_t r = W[rec & 0xffu];
                    const uint64_t pbk = S->block + (uint64_t) ((rec >> 8) & 0xffu);
                    const unsigned kind = (rec >> 16) & 3u;
                    if (kind == 0) ok = (S->lastw[r] >> 1) <= S->block;          /* last written by a block before this window, or never */
                    else ok = S->lastw[r] == (((pbk + 1) << 1) | (kind == 1 ? 1u : 0u));
                }
                if (!ok) {
                    if (st) ++st->slots_invalid;
                    break;                              /* recomputed as the front slot of the next window */
                }
            }
            if (st) ++st->slots_used;
            worst = rs->os.heap[0];
            /* reflection trial of block blk */
            if (host_eval) {
                if (ops->read_slot(e, blk, 1, rs->xtmp)) { engine_failed(S); return S->ret; }
                status[j].fT = pb->f((unsigned) n, rs->xtmp, NULL, pb->f_data);
            }
            fcand = status[j].fT;
            ++*stop->nevals_p;
            if (st) ++st->evals_trial;
            if (nla_stop_forced(rs->sp)) { trace_add(pb, fcand, -1, 1, 0); ret = NLOPT_FORCED_STOP; ++j; break; }
            if (fcand < rs->F[worst]) accepted = 1;
            else {
                trace_add(pb, fcand, -1, 1, 0);
                if (nla_stop_evals(stop)) { ret = NLOPT_MAXEVAL_REACHED; ++j; break; }   /* only after a rejection */
                if (nla_stop_time(rs->sp)) { ret = NLOPT_MAXTIME_REACHED; ++j; break; }
                /* local mutation: consumes block blk+1 (crs.c:139-146) */
                kind = 2;
                if (host_eval) {
                    if (ops->mutate_slot(e, blk, rs->os.best) || ops->read_slot(e, blk, 1, rs->xtmp)) { engine_failed(S); return S->ret; }
                    status[j].fM = pb->f((unsigned) n, rs->xtmp, NULL, pb->f_data);
                }
                fcand = status[j].fM;
                ++*stop->nevals_p;
                if (st) { ++st->evals_mutation; if (blk + 1 < S->fresh_from) ++st->slots_role; }
                if (nla_stop_forced(rs->sp)) { trace_add(pb, fcand, -1, 2, 0); ret = NLOPT_FORCED_STOP; j += 2; break; }
                if (fcand < rs->F[worst]) accepted = 1;
                else {
                    trace_add(pb, fcand, -1, 2, 0);
                    if (nla_stop_evals(stop)) { ret = NLOPT_MAXEVAL_REACHED; j += 2; break; }
                    if (nla_stop_time(rs->sp)) { ret = NLOPT_MAXTIME_REACHED; j += 2; break; }
                }
            }
            if (accepted) {
                /* the reference's x is the point of the last strict improvement, not whatever row is best at the end (an
                 * equal-f row with a smaller address can take over the tree's minimum without x being copied, crs.c:253):
                 * if that row is about to be overwritten (only possible when the whole population has one f), save it now */
                if (worst == rs->xrow) {
                    if (ops->read_row(e, rs->xrow, rs->x)) { engine_failed(S); return S->ret; }
                    rs->xrow = -1;
                }
                /* memcpy(worst->k, d->p) + resort (crs.c:153-154); the row write is deferred */
                rs->F[worst] = fcand;
                os_top_changed(&rs->os);
                trace_add(pb, fcand, worst, kind, 1);
                if (st) ++st->accepted;
                cblock[ncommit] = blk; ckind[ncommit] = host_eval ? 1 : kind; crow[ncommit] = worst; ++ncommit;
                if (S->forward) S->lastw[worst] = ((blk + 1) << 1) | (kind == 1 ? 1u : 0u);
                if (key_less(rs->F, worst, rs->os.best)) { rs->os.best = worst; best_changed = 1; }
                if (after_accept(rs, blk, host_eval ? 1 : kind)) { engine_failed(S); return S->ret; }
                ret = rs->ret;
            }
            j += (kind == 2) ? 2 : 1;
        }
        S->block += (uint64_t) j;
        if (best_changed) {
            /* every slot in flight started from the old best row: forget them all */
            if (st && S->fresh_from > S->block) st->slots_newbest += S->fresh_from - S->block;
            S->fresh_from = S->block;
        }
        if (S->fresh_from < S->block) S->fresh_from = S->block;   /* a mutation consumed the block past the window */
        {   /* adapt the window to the observed number of blocks consumed per pass */
            double obs = (j >= K) ? 2.0 * K : (double) j;
            S->runlen = 0.7 * S->runlen + 0.3 * obs;
            if (S->runlen < 1.0) S->runlen = 1.0;
        }
        if (ncommit > 0) {
            /* a row replaced twice in one pass keeps only its last content */
            int k, m = 0;
            for (k = 0; k < ncommit; ++k) {
                int later = 0, q;
                for (q = k + 1; q < ncommit; ++q) if (crow[q] == crow[k]) { later = 1; break; }
                if (!later) { cblock[m] = cblock[k]; ckind[m] = ckind[k]; crow[m] = crow[k]; ++m; }
            }
            if (ops->commit(e, m, cblock, ckind, crow)) { engine_failed(S); return S->ret; }
        }
        if (st) st->t_walk_s += nla_seconds() - t_walk0;
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
    if (ret != NLOPT_FAILURE && S->rs.os.nheap > 0 && S->rs.xrow >= 0 && S->rs.ops->read_row(S->rs.e, S->rs.xrow, S->rs.x)) {
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
