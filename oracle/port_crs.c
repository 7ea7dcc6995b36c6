/* port_crs.c — CPU ORACLE (test infrastructure): Controlled Random Search 2 with local mutation.
 *
 * Serial restatement of src/algs/crs/crs.c with 64-bit row indexing (the reference's `int`
 * products ps + i*(n+1) overflow at n=4096, N=1e6 — crs.c:101,212) and a
 * different container: the reference keeps rows [f, x...] in one array ordered by a red-black
 * tree of row pointers (crs.c:36-56); here f and x live in separate arrays and the ordered set
 * is a binary max-heap of row indices keyed (f, row) plus a tracked minimum — the *order* it
 * answers with (f ascending, ties by row address == row index, crs.c:51-56) is what parity
 * needs, not the tree.  RNG consumption order is exactly the reference's (Appendix A of SURVEY).
 */
#include "port_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n;
    int64_t N;
    const double *lb, *ub;
    orc_func f; void *f_data;
    orc_stop *stop;
    double *F;            /* N   objective values */
    double *X;            /* N*n candidate rows */
    double *p; double pf; /* scratch trial point (crs.c:45, the (N+1)-th row) */
    int64_t *heap;        /* max-heap of row indices by (F, row) */
    int64_t nheap;
    int64_t best;         /* argmin by (F, row) */
    orc_trace *trace;
} crs_state;

static int key_less(const crs_state *d, int64_t a, int64_t b)   /* crs_compare, crs.c:51-56 */
{
    if (d->F[a] < d->F[b]) return 1;
    if (d->F[a] > d->F[b]) return 0;
    return a < b;
}

static void sift_down(crs_state *d, int64_t pos)
{
    int64_t v = d->heap[pos];
    for (;;) {
        int64_t c = 2 * pos + 1;
        if (c >= d->nheap) break;
        if (c + 1 < d->nheap && key_less(d, d->heap[c], d->heap[c + 1])) ++c;
        if (!key_less(d, v, d->heap[c])) break;
        d->heap[pos] = d->heap[c];
        pos = c;
    }
    d->heap[pos] = v;
}

static void heap_push(crs_state *d, int64_t row)
{
    int64_t pos = d->nheap++;
    while (pos > 0) {
        int64_t par = (pos - 1) / 2;
        if (!key_less(d, d->heap[par], row)) break;
        d->heap[pos] = d->heap[par];
        pos = par;
    }
    d->heap[pos] = row;
    if (d->nheap == 1 || key_less(d, row, d->best)) d->best = row;
}

static void trace_add(crs_state *d, double f, int64_t row, int kind, int accepted)
{
    orc_trace *t = d->trace;
    if (!t || t->len >= t->cap) { if (t) ++t->len; return; }
    t->rec[t->len].f = f; t->rec[t->len].row = row; t->rec[t->len].kind = kind; t->rec[t->len].accepted = accepted;
    ++t->len;
}

/* x = 2G - x_n : reflection of one of n random rows through the centroid of best + the other
 * n-1 (crs.c:63-121).  Vitter method A over the N-1 non-best rows in ascending row order. */
static void reflection_trial(crs_state *d, double *x, int64_t i0)
{
    const int n = d->n;
    int jn, k;
    int Nleft = (int) (d->N - 1), nleft = n, Nfree = Nleft - nleft;
    int64_t i = 0;
    const double *xi;

    memcpy(x, d->X + i0 * n, sizeof(double) * (size_t) n);
    jn = orc_iurand(n);                                   /* crs.c:72 */
    i += (i == i0);
    while (nleft > 1) {                                   /* crs.c:93-108 */
        double q = ((double) Nfree) / Nleft;
        double v = orc_urand(0., 1.);
        while (q > v) {
            ++i; i += (i == i0);
            --Nfree; --Nleft;
            q = (q * Nfree) / Nleft;
        }
        xi = d->X + i * n;
        if (jn-- == 0) for (k = 0; k < n; ++k) x[k] -= xi[k] * (0.5 * n);
        else           for (k = 0; k < n; ++k) x[k] += xi[k];
        ++i; i += (i == i0);
        --Nleft; --nleft;
    }
    i += orc_iurand(Nleft); i += (i == i0);                 /* crs.c:109: bump only on equality */
    xi = d->X + i * n;
    if (jn-- == 0) for (k = 0; k < n; ++k) x[k] -= xi[k] * (0.5 * n);
    else           for (k = 0; k < n; ++k) x[k] += xi[k];
    for (k = 0; k < n; ++k) {                             /* crs.c:116-120 */
        x[k] *= 2.0 / n;
        if (x[k] > d->ub[k]) x[k] = d->ub[k];
        else if (x[k] < d->lb[k]) x[k] = d->lb[k];
    }
}

/* one accepted replacement (crs_trial, crs.c:125-156); NUM_MUTATION = 1 (crs.c:123) */
static int one_trial(crs_state *d)
{
    const int n = d->n;
    int64_t best = d->best, worst = d->heap[0];
    int mutation = 1, kind = 1, i;
    reflection_trial(d, d->p, best);
    for (;;) {
        d->pf = d->f((unsigned) n, d->p, NULL, d->f_data);
        ++d->stop->nevals;
        if (d->stop->force_stop) { trace_add(d, d->pf, -1, kind, 0); return ORC_FORCED_STOP; }
        if (d->pf < d->F[worst]) break;
        trace_add(d, d->pf, -1, kind, 0);
        if (orc_stop_evals(d->stop)) return ORC_MAXEVAL_REACHED;   /* only after a rejection: crs.c:137 */
        if (orc_stop_time(d->stop)) return ORC_MAXTIME_REACHED;
        if (mutation) {                                            /* crs.c:139-146 */
            const double *xb = d->X + best * n;
            for (i = 0; i < n; ++i) {
                double w = orc_urand(0., 1.);
                d->p[i] = xb[i] * (1 + w) - w * d->p[i];
                if (d->p[i] > d->ub[i]) d->p[i] = d->ub[i];
                else if (d->p[i] < d->lb[i]) d->p[i] = d->lb[i];
            }
            mutation--; kind = 2;
        } else {
            reflection_trial(d, d->p, best);
            mutation = 1; kind = 1;
        }
    }
    trace_add(d, d->pf, worst, kind, 1);
    memcpy(d->X + worst * n, d->p, sizeof(double) * (size_t) n);   /* crs.c:153 */
    d->F[worst] = d->pf;
    sift_down(d, 0);                                               /* rb resort, crs.c:154 */
    if (key_less(d, worst, d->best)) d->best = worst;
    return ORC_SUCCESS;
}

int orc_crs_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub,
                     double *x, double *minf, orc_stop *stop, long population, orc_trace *trace)
{
    crs_state d;
    int ret = ORC_SUCCESS;
    int64_t i;

    memset(&d, 0, sizeof d);
    d.N = population ? population : 10 * ((int64_t) n + 1);           /* crs.c:172-179 */
    if (d.N < n + 1) return ORC_INVALID_ARGS;                          /* crs.c:180-184 */
    d.n = n; d.lb = lb; d.ub = ub; d.f = f; d.f_data = f_data; d.stop = stop; d.trace = trace;
    d.F = (double *) malloc(sizeof(double) * (size_t) d.N);
    d.X = (double *) malloc(sizeof(double) * (size_t) d.N * (size_t) n);
    d.p = (double *) malloc(sizeof(double) * (size_t) n);
    d.heap = (int64_t *) malloc(sizeof(int64_t) * (size_t) d.N);
    if (!d.F || !d.X || !d.p || !d.heap) { ret = ORC_OUT_OF_MEMORY; goto done; }

    /* crs_init, crs.c:203-226: row 0 = starting guess, rows 1..N-1 = n urands each */
    memcpy(d.X, x, sizeof(double) * (size_t) n);
    for (i = 0; i < d.N && ret == ORC_SUCCESS; ++i) {
        double *xi = d.X + i * n;
        if (i > 0) for (int j = 0; j < n; ++j) xi[j] = orc_urand(lb[j], ub[j]);
        d.F[i] = f((unsigned) n, xi, NULL, f_data);
        ++stop->nevals;
        heap_push(&d, i);
        trace_add(&d, d.F[i], i, 0, 1);
        if (d.F[i] < stop->minf_max) ret = ORC_STOPVAL_REACHED;
        else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED;
        else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED;
    }

    *minf = d.F[d.best];                                               /* crs.c:246-248 */
    memcpy(x, d.X + d.best * n, sizeof(double) * (size_t) n);

    while (ret == ORC_SUCCESS) {                                       /* crs.c:250-270 */
        if (ORC_SUCCESS == (ret = one_trial(&d))) {
            int64_t b = d.best;
            if (d.F[b] < *minf) {
                if (d.F[b] < stop->minf_max) ret = ORC_STOPVAL_REACHED;
                else if (orc_stop_f(stop, d.F[b], *minf)) ret = ORC_FTOL_REACHED;   /* successive bests */
                else if (orc_stop_x(stop, d.X + b * n, x)) ret = ORC_XTOL_REACHED;
                *minf = d.F[b];
                memcpy(x, d.X + b * n, sizeof(double) * (size_t) n);
            }
            if (ret != ORC_SUCCESS) {                                   /* crs.c:263-268 (quirk kept) */
                if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED;
                else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED;
            }
        }
    }
done:
    free(d.F); free(d.X); free(d.p); free(d.heap);
    return ret;
}
