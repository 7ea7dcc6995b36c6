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
/* What the chain asks of it: the worst row (crs.c:134 reads the tree's maximum), "the worst row got a smaller value" once per accepted
 * trial (crs.c:153-154: the node is re-inserted), the K worst rows in order once per window, the best row.  The walk below runs on the
 * host between two launches — the device idles meanwhile — and with a binary heap of row indices keyed through F[] the sift-down
 * after an acceptance was most of it: 17 levels x three dependent cache misses (two child indices, two values) = 0.5 us per accepted
 * trial at N = 1e5, 61 of the 73 us between two 128-slot windows (measured over the emulated device, round 4).  So: a 4-ARY heap with
 * the KEY STORED IN THE NODE, the four children of a node in ONE 64-byte line (node i lives in slot i + 3 of a 64-byte-aligned array:
 * children 4i+1 .. 4i+4 are slots 4i+4 .. 4i+7), and the grandchildren's lines prefetched while the children are compared: 9 levels,
 * one miss each, overlapped.  The order is the same total order (f, then row index: crs_compare, crs.c:51-56), so the worst row, the
 * K worst in order and every decision are what the binary heap — and the reference's red-black tree — give. */
typedef struct { double f; int64_t row; } osnode;
typedef struct {
    const double *F;     /* the rows' values (the copy everything else reads); a node's f is F[row] */
    osnode *node;        /* 64-byte aligned; heap position i is node[i + 3] */
    void *mem;
    int64_t nheap, best;
    void *cand;          /* scratch for top-k: 2 K + 2 candidates ... */
    int64_t *gpos;       /* ... and 4 (K + 1) positions: the sorted children of every visited node */
} ordset;
#define OSN(s, i) ((s)->node[(i) + 3])

static inline int key_less(const double *F, int64_t a, int64_t b)       /* crs_compare, crs.c:51-56 */
{
    if (F[a] < F[b]) return 1;
    if (F[a] > F[b]) return 0;
    return a < b;
}
static inline int node_less(const osnode a, const osnode b)              /* the same on keys held in the nodes */
{
    if (a.f < b.f) return 1;
    if (a.f > b.f) return 0;
    return a.row < b.row;
}

static int os_alloc(ordset *s, int64_t N, int kmax)
{
    s->mem = malloc(sizeof(osnode) * (size_t) (N + 8) + 64);
    s->cand = malloc(32 * (size_t) (2 * kmax + 4));          /* (sizeof(oscand) = 24) */
    s->gpos = (int64_t *) malloc(sizeof(int64_t) * 4 * (size_t) (kmax + 2));
    if (!s->mem || !s->cand || !s->gpos) return -1;
    s->node = (osnode *) (((uintptr_t) s->mem + 63u) & ~(uintptr_t) 63u);
    s->nheap = 0; s->best = 0;
    return 0;
}
static void os_free(ordset *s) { free(s->mem); free(s->cand); free(s->gpos); s->mem = NULL; s->cand = NULL; s->gpos = NULL; s->node = NULL; }

static void os_push(ordset *s, int64_t row)
{
    const osnode v = { s->F[row], row };
    int64_t pos = s->nheap++;
    while (pos > 0) {
        const int64_t par = (pos - 1) / 4;
        if (!node_less(OSN(s, par), v)) break;
        OSN(s, pos) = OSN(s, par);
        pos = par;
    }
    OSN(s, pos) = v;
    if (s->nheap == 1 || key_less(s->F, row, s->best)) s->best = row;
}

/* the k largest rows, worst first, without disturbing the heap: best-first walk over heap nodes.  The candidates are kept in a small
 * binary max-heap with their keys inline; the four children of a visited node enter it ONE AT A TIME — sorted once, the largest now,
 * each next one when its elder sibling is taken — so a visit costs two insertions, not four. */
typedef struct { osnode key; int32_t grp, idx; } oscand;
static int os_topk(ordset *s, int k, int64_t *out, int64_t *out_pos, double *out_f)
{
    oscand *c = (oscand *) s->cand;          /* 2 k + 2 candidates, then the sibling groups: 4 positions per visited node */
    int64_t *gpos = s->gpos;
    int nc = 0, got = 0, ng = 0;
    if (s->nheap == 0 || k <= 0) return 0;
#define CAND_PUSH(KEY, G, I) do { const osnode kv_ = (KEY); int pos_ = nc++;                                    \
        while (pos_ > 0) { const int par_ = (pos_ - 1) / 2; if (!node_less(c[par_].key, kv_)) break; c[pos_] = c[par_]; pos_ = par_; } \
        c[pos_].key = kv_; c[pos_].grp = (G); c[pos_].idx = (I); } while (0)
    gpos[0] = 0; gpos[1] = gpos[2] = gpos[3] = -1; ng = 1;       /* group 0: the root alone */
    CAND_PUSH(OSN(s, 0), 0, 0);
    while (got < k && nc > 0) {
        const oscand top = c[0];
        const int64_t p = gpos[4 * top.grp + top.idx];
        if (out_pos) out_pos[got] = p;
        if (out_f) out_f[got] = top.key.f;
        out[got++] = top.key.row;
        /* pop */
        {
            const oscand last = c[--nc];
            if (nc > 0) {
                int pos = 0;
                for (;;) {
                    int ch = 2 * pos + 1;
                    if (ch >= nc) break;
                    if (ch + 1 < nc && node_less(c[ch].key, c[ch + 1].key)) ++ch;
                    if (!node_less(last.key, c[ch].key)) break;
                    c[pos] = c[ch];
                    pos = ch;
                }
                c[pos] = last;
            }
        }
        if (got == k) break;
        if (top.idx < 3 && gpos[4 * top.grp + top.idx + 1] >= 0)                       /* its next sibling */
            CAND_PUSH(OSN(s, gpos[4 * top.grp + top.idx + 1]), top.grp, top.idx + 1);
        if (4 * p + 1 < s->nheap) {                                                    /* its children, largest first */
            int64_t *g = gpos + 4 * ng;
            const int64_t c0 = 4 * p + 1, end = c0 + 4 < s->nheap ? c0 + 4 : s->nheap;
            int m = 0, i, j;
            for (int64_t q = c0; q < end; ++q) {
                for (i = m; i > 0 && node_less(OSN(s, g[i - 1]), OSN(s, q)); --i) g[i] = g[i - 1];
                g[i] = q; ++m;
            }
            for (j = m; j < 4; ++j) g[j] = -1;
            CAND_PUSH(OSN(s, g[0]), ng, 0);
            ++ng;
        }
    }
#undef CAND_PUSH
    return got;
}

/* Heap positions whose rows got smaller values (F[] holds them), `cnt` of them ordered by DECREASING LEVEL: restore the heap.
 * Floyd's order — deeper nodes first — makes every sift-down meet valid sub-heaps; nodes of one level have disjoint subtrees, so
 * their sift-downs are run INTERLEAVED, one level per turn each, the children's line of a hole prefetched a turn ahead: the misses of
 * up to 64 walks overlap instead of queueing behind one another (a lone sift-down is nine dependent misses). */
static void os_repair(ordset *s, const int64_t *posdesc, int cnt)
{
    const int64_t n = s->nheap;
    int i = 0;
    while (i < cnt) {
        int64_t hole[64], lo = 0, hi = 1;
        osnode v[64];
        int nb = 0, b;
        while (posdesc[i] >= hi) { lo = hi; hi = 4 * hi + 1; }          /* level of the deepest position left: [lo, hi) */
        while (i < cnt && nb < 64 && posdesc[i] >= lo) {
            hole[nb] = posdesc[i]; v[nb] = OSN(s, posdesc[i]); v[nb].f = s->F[v[nb].row];
            if (4 * hole[nb] + 1 < n) __builtin_prefetch(&OSN(s, 4 * hole[nb] + 1));
            ++nb; ++i;
        }
        while (nb > 0) {
            for (b = 0; b < nb; ) {
                const int64_t c = 4 * hole[b] + 1, end = c + 4 < n ? c + 4 : n;
                int64_t m = c, q;
                int settled = c >= n;
                if (!settled) {
                    for (q = c + 1; q < end; ++q) if (node_less(OSN(s, m), OSN(s, q))) m = q;
                    settled = !node_less(v[b], OSN(s, m));
                }
                if (settled) { OSN(s, hole[b]) = v[b]; --nb; hole[b] = hole[nb]; v[b] = v[nb]; continue; }
                OSN(s, hole[b]) = OSN(s, m);
                hole[b] = m;
                if (4 * m + 1 < n) __builtin_prefetch(&OSN(s, 4 * m + 1));
                ++b;
            }
        }
    }
}

/* ---- the worst rows between two looks at the heap ------------------------------------------------------------------------------
 * Only worst rows are ever replaced.  So after the heap has given its `cnt` worst rows in order (the RESERVOIR: a few windows' worth),
 * the chain can be followed without touching it: the current worst row is the head of a sorted list T that starts as the reservoir;
 * an accepted trial removes the head and, if its value still lies above the reservoir's last entry as it was when it was drawn (thr),
 * puts the row back at its place in T — every row outside T is <= thr, and rows outside T do not change.  A window's list of worst
 * rows (what the device is given, crs.c:134's tree maximum for every trial of the window) is T's first K entries.  The heap learns of
 * the replacements in ONE batch (os_repair over the positions the rows had when the reservoir was drawn: the heap is not touched
 * in between, so they still hold) right before the next reservoir is drawn — and that refresh is done while the device is busy with a
 * window (tl_idle, called by the engine between its launch and its wait) whenever the list would not carry the window after: the
 * sift-downs and the best-first extraction that used to sit between two launches are off the serial path.  A list that runs short
 * anyway (window sizes jump) is refreshed on the spot. */
typedef struct { double f; int64_t row, pos; int recorded; } topent;
typedef struct {
    topent *T; int head, tail, cap;         /* sorted worst first: T[head .. tail) */
    double thr_f; int64_t thr_row;          /* the reservoir's last entry when it was drawn */
    int whole;                              /* the reservoir was the whole population: nothing lies outside T */
    int64_t *chg; int nchg, chgcap;         /* heap positions of the rows replaced since the reservoir was drawn (each once) */
    int64_t *rows, *pos; double *fs;        /* extraction buffers (cap entries) */
    int64_t *scratch; signed char *lev;     /* repair order (chgcap entries) */
    int inflight;                           /* slots of the window the device is working on (tl_idle) */
    uint64_t refreshes, refreshes_idle;
    nlopt_amd_stats *st;                    /* or NULL */
} toplist;

static void tl_free(toplist *t) { free(t->T); free(t->chg); free(t->rows); free(t->pos); free(t->fs); free(t->scratch); free(t->lev); memset(t, 0, sizeof *t); }
static int tl_alloc(toplist *t, int kmax)
{
    memset(t, 0, sizeof *t);
    t->cap = 3 * kmax + 64;                 /* a reservoir of up to 3 windows */
    t->chgcap = 4 * kmax + 128;
    t->T = (topent *) malloc(sizeof(topent) * (size_t) (2 * t->cap));        /* room to slide: entries live in [head, tail) of 2 cap */
    t->chg = (int64_t *) malloc(sizeof(int64_t) * (size_t) t->chgcap);
    t->rows = (int64_t *) malloc(sizeof(int64_t) * (size_t) t->cap);
    t->pos = (int64_t *) malloc(sizeof(int64_t) * (size_t) t->cap);
    t->fs = (double *) malloc(sizeof(double) * (size_t) t->cap);
    t->scratch = (int64_t *) malloc(sizeof(int64_t) * (size_t) t->chgcap);
    t->lev = (signed char *) malloc((size_t) t->chgcap);
    return (t->T && t->chg && t->rows && t->pos && t->fs && t->scratch && t->lev) ? 0 : -1;
}

/* the heap takes the replacements made since the last reservoir, then gives the `want` worst rows (at most cap) */
static void tl_refresh(toplist *t, ordset *s, int want)
{
    int i, j, cnt = t->nchg, got;
    const double t_r0 = t->st ? nla_seconds() : 0.;
    if (cnt > 0) {   /* decreasing LEVEL is all os_repair needs (the nodes of a level are independent): a counting sort over the levels */
        int lvcnt[40] = { 0 }, lvoff[40], nl = 0;
        for (i = 0; i < cnt; ++i) { int64_t hi = 1; int l = 0; while (t->chg[i] >= hi) { hi = 4 * hi + 1; ++l; } t->lev[i] = (signed char) l; ++lvcnt[l]; if (l + 1 > nl) nl = l + 1; }
        for (j = nl - 1, i = 0; j >= 0; --j) { lvoff[j] = i; i += lvcnt[j]; }
        for (i = 0; i < cnt; ++i) t->scratch[lvoff[(int) t->lev[i]]++] = t->chg[i];
        os_repair(s, t->scratch, cnt);
        t->nchg = 0;
    }
    if (want > t->cap) want = t->cap;
    if ((int64_t) want > s->nheap) want = (int) s->nheap;
    got = os_topk(s, want, t->rows, t->pos, t->fs);
    t->head = 0; t->tail = got;
    for (i = 0; i < got; ++i) { t->T[i].f = t->fs[i]; t->T[i].row = t->rows[i]; t->T[i].pos = t->pos[i]; t->T[i].recorded = 0; }
    t->whole = (int64_t) got == s->nheap;
    if (got > 0) { t->thr_f = t->T[got - 1].f; t->thr_row = t->T[got - 1].row; }
    ++t->refreshes;
    if (t->st) { ++t->st->list_refreshes; t->st->list_refreshes_beside_device += t->inflight > 0; t->st->t_list_refresh_s += nla_seconds() - t_r0; }
}

/* make sure the list holds the `need` worst rows (need <= N); `ahead`: how many to draw if it has to be drawn again */
static void tl_need(toplist *t, ordset *s, int need, int ahead)
{
    if (t->tail - t->head >= need || (t->whole && t->tail - t->head >= (int) s->nheap)) return;
    tl_refresh(t, s, ahead > need ? ahead : need);
}

static inline int64_t tl_worst(const toplist *t) { return t->T[t->head].row; }

/* the device is working on a window of t->inflight slots whose list is T's head: if what will be left of T cannot carry the window
 * after it, the heap takes the replacements made so far and gives the next reservoir NOW, beside the device.  The head of the new list
 * is the window's list again (both are the worst rows in order), so the walk that follows consumes it as it would have the old one. */
static void tl_idle(toplist *t, ordset *s)
{
    const int k = t->inflight;
    if (k <= 0 || t->whole || t->tail - t->head - k >= k) return;
    { const uint64_t r0 = t->refreshes; tl_refresh(t, s, 3 * k + 16); t->refreshes_idle += t->refreshes - r0; }
}

/* the head row has just been given a smaller value in F[] */
static void tl_accepted(toplist *t, ordset *s)
{
    topent e = t->T[t->head++];
    const double f = s->F[e.row];
    if (!e.recorded) {
        t->chg[t->nchg++] = e.pos;
        e.recorded = 1;
    }
    /* still above the reservoir's last entry as it stood when it was drawn (key order: f, then row)?  then it is among the worst again */
    if (t->whole || t->thr_f < f || (!(t->thr_f > f) && t->thr_row < e.row)) {
        int lo = t->head, hi = t->tail;       /* first entry that is smaller than the new key: insert in front of it */
        osnode k; k.f = f; k.row = e.row;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            osnode m; m.f = t->T[mid].f; m.row = t->T[mid].row;
            if (node_less(m, k)) hi = mid; else lo = mid + 1;
        }
        if (t->tail == 2 * t->cap) {          /* slide the live part back to the front */
            memmove(t->T, t->T + t->head, sizeof(topent) * (size_t) (t->tail - t->head));
            lo -= t->head; t->tail -= t->head; t->head = 0;
        }
        memmove(t->T + lo + 1, t->T + lo, sizeof(topent) * (size_t) (t->tail - lo));
        e.f = f;
        t->T[lo] = e;
        ++t->tail;
    }
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
    toplist tl;                    /* the worst rows between two looks at the heap (W is a window's copy of its head) */
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
    free(S->rs.F); os_free(&S->rs.os); free(S->rs.xtmp);
    free(S->status); free(S->tprev); free(S->W); tl_free(&S->tl); free(S->cblock); free(S->ckind); free(S->crow);
    free(S->Wf); free(S->fwcnt); free(S->fwrec); free(S->lastw);
    free(S);
}

static void crs_idle(void *arg)            /* from the engine, between handing a pass to the device and waiting for it */
{
    nla_crs_session *S = (nla_crs_session *) arg;
    tl_idle(&S->tl, &S->rs.os);
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
    /* round 5, below n = 2048 (a window there is 0.1 - 0.2 ms: the turnaround between two windows is what counts): 256 slots — n = 64
     * 1.44 -> 1.56 M evals/s, n = 128 1.32 -> 1.39 M, n = 256 1.03 -> 1.08 M, n = 512 unchanged (profiles/r05_staged_ab.txt) */
    if (S->forward && pb->max_spec <= 0 && S->Kmax > (n >= 2048 ? 128 : 256)) S->Kmax = n >= 2048 ? 128 : 256;
    /* a small population cannot feed a deep window: a value that is accepted lands among the window's K worst rows of N with probability
     * ~ K / N, and past CH_EXTRA = 32 such values per window (or one landing twice) the device's resolution stops being verifiable and
     * the rest of the window is dropped — K <= N / 16 keeps that to a few per window (N = 2000, n = 64: 27 % of the slots dropped at
     * K = 256 over the emulated device, 3 % at 125) */
    if (S->forward && pb->max_spec <= 0 && (int64_t) S->Kmax > N / 16) S->Kmax = N / 16 > 8 ? (int) (N / 16) : 8;
    S->runlen = 4.0;
    rs->ops = ops; rs->e = e; rs->pb = &S->pb; rs->x = x; rs->minf = minf;
    rs->need_x = (stop->xtol_rel > 0 || stop->xtol_abs != NULL);

    rs->F = (double *) malloc(sizeof(double) * (size_t) N);
    const int os_bad = os_alloc(&rs->os, N, 3 * S->Kmax + 64);       /* (top-k scratch for a whole reservoir: tl_alloc's cap) */
    rs->xtmp = (double *) malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));
    S->status = (nla_crs_slot_status *) malloc(sizeof(nla_crs_slot_status) * (size_t) S->Kmax);
    S->tprev = (int32_t *) calloc(TRING, sizeof(int32_t));
    S->W = (int64_t *) malloc(sizeof(int64_t) * (size_t) S->Kmax);
    const int tl_bad = tl_alloc(&S->tl, S->Kmax);
    S->cblock = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) S->Kmax);
    S->ckind = (int32_t *) malloc(sizeof(int32_t) * (size_t) S->Kmax);
    S->crow = (int64_t *) malloc(sizeof(int64_t) * (size_t) S->Kmax);
    S->Wf = (double *) malloc(sizeof(double) * (size_t) S->Kmax);
    if (S->forward) {
        S->fwcnt = (uint32_t *) calloc((size_t) S->Kmax, sizeof(uint32_t));
        S->fwrec = (uint32_t *) calloc((size_t) S->Kmax * FWCAP, sizeof(uint32_t));
        S->lastw = (uint64_t *) calloc((size_t) N, sizeof(uint64_t));
    }
    if (!rs->F || os_bad || !rs->xtmp || !S->status || !S->tprev || !S->W || tl_bad || !S->Wf || !S->cblock ||
        !S->ckind || !S->crow || (S->forward && (!S->fwcnt || !S->fwrec || !S->lastw))) {
        session_free(S);
        *ret_out = NLOPT_OUT_OF_MEMORY;
        return NULL;
    }
    rs->os.F = rs->F;
    S->tl.st = st;
    if (ops->set_idle) ops->set_idle(e, crs_idle, S);

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
        tl_need(&S->tl, &rs->os, nW, 3 * K + 16);
        for (a = 0; a < nW; ++a) { W[a] = S->tl.T[S->tl.head + a].row; S->Wf[a] = S->tl.T[S->tl.head + a].f; }
        if (S->forward) for (a = 0; a < nW; ++a) __builtin_prefetch(&S->lastw[W[a]], 1);   /* the walk writes a record per replaced row */
        const double t_eng0 = st ? nla_seconds() : 0.;
        const uint64_t gl0 = st ? st->gather_launches : 0;
        S->tl.inflight = nW;                    /* (what crs_idle may assume the walk consumes of the list) */
        if (S->forward) {
            /* every slot of the window is computed in this launch; what an earlier window left unconsumed is dropped */
            if (st && S->fresh_from > S->block) st->slots_invalid += S->fresh_from - S->block;
            S->fresh_from = S->block;
            if (ops->chain(e, S->block, K, rs->os.best, rs->F[rs->os.best], W, S->Wf, nW, status, S->fwcnt, S->fwrec, FWCAP)) { engine_failed(S); return S->ret; }
            if (pb->comm && ops->stop_flags_in && ops->stop_flags_out) {      /* a column-sharded window: the ranks' stop bits crossed inside the launch */
                int forced = 0, timed = 0;
                ops->stop_flags_out(e, &forced, &timed);
                nla_stop_view(stop, forced, timed, &S->view, &S->agreed_force);
                rs->sp = &S->view;
            }
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
        S->tl.inflight = 0;
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
            if (S->tl.head == S->tl.tail) tl_need(&S->tl, &rs->os, 1, 3 * K + 16);     /* (a walk never outruns its window's list; kept for safety) */
            worst = tl_worst(&S->tl);
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
                tl_accepted(&S->tl, &rs->os);
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
    if (S->rs.ops->set_idle) S->rs.ops->set_idle(S->rs.e, NULL, NULL);       /* the engine outlives the session */
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
