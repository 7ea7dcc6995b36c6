/* port_esch.c — CPU ORACLE (test infrastructure): NLOPT_GN_ESCH, the evolutionary strategy of
 * src/algs/esch/esch.c:28-262 (C. H. da Silva Santos' ES with Cauchy mutation), restated serially with 64-bit
 * indexing.  Same loop order and RNG consumption as the reference:
 *   randcauchy (:28-50)   urand(0,1) redrawn until t tan(pi (u - 1/2)) + mi lies in [mi - band/2, mi + band/2]
 *                          (mi = 0, t = 1, band = 10), folded to [0, band) and scaled into [lb, ub];
 *   initial parents and offspring, element by element (:133-164), parent 0 := x (:148);
 *   parent evaluation with the best-point / stop tests after every candidate (:168-183);
 *   generations (:187-251): crossover — three iurand per offspring (:192-203); (no n)/10 point mutations, each
 *   iurand(no), iurand(n), randcauchy, later ones overwriting earlier ones (:207-218); offspring evaluation
 *   (:222-238); selection — parents and offspring together, STABLE sort by fitness (nlopt_qsort_r = glibc's merge
 *   sort on the {pointer, fitness} structs, :243-251), best np become the parents, the rest the offspring.
 */
#include "port_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double randcauchy(double min, double max)
{
    const double mi = 0, t = 1, band = 10;
    const double limit_inf = mi - (band * 0.5), limit_sup = mi + (band * 0.5);
    double na_unif, cauchy_mit, valor;
    do {
        na_unif = orc_urand(0, 1);
        cauchy_mit = t * tan((na_unif - 0.5) * 3.14159265358979323846) + mi;
    } while ((cauchy_mit < limit_inf) || (cauchy_mit > limit_sup));
    if (cauchy_mit < 0) cauchy_mit = -cauchy_mit;
    else cauchy_mit = cauchy_mit + (band * 0.5);
    valor = cauchy_mit / band;
    valor = min + (max - min) * valor;
    return valor;
}

static const double *g_fit;
static int cmp_stable(const void *a_, const void *b_)
{
    const int64_t a = *(const int64_t *) a_, b = *(const int64_t *) b_;
    if (g_fit[a] < g_fit[b]) return -1;
    if (g_fit[a] > g_fit[b]) return +1;
    return a < b ? -1 : (a > b ? +1 : 0);
}

int orc_esch_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf, orc_stop *stop,
                      long np_, long no_, orc_trace *trace)
{
    int64_t np = np_ ? np_ : 40, no = no_ ? no_ : 60, id, item, i;            /* esch.c:96-97 */
    int ret = ORC_SUCCESS;
    double *rows, *fit, *fit2;
    int64_t *slot, *order, *slot2;          /* slot[i] = physical row of individual i (parents 0..np-1, offspring np..np+no-1) */
    if (np < 1 || no < 1) return ORC_INVALID_ARGS;
    *minf = HUGE_VAL;                       /* set by nlopt_optimize_ before the dispatch (optimize.c:541) */
    rows = (double *) malloc(sizeof(double) * (size_t) (np + no) * (size_t) n);
    fit = (double *) calloc((size_t) (np + no), sizeof(double));
    fit2 = (double *) calloc((size_t) (np + no), sizeof(double));
    slot = (int64_t *) malloc(sizeof(int64_t) * (size_t) (np + no));
    slot2 = (int64_t *) malloc(sizeof(int64_t) * (size_t) (np + no));
    order = (int64_t *) malloc(sizeof(int64_t) * (size_t) (np + no));
    for (i = 0; i < np + no; ++i) slot[i] = i;
    for (id = 0; id < np + no; ++id)        /* parents, then offspring (:133-164) */
        for (item = 0; item < n; ++item) rows[(size_t) id * n + item] = randcauchy(lb[item], ub[item]);
    memcpy(rows, x, sizeof(double) * (size_t) n);                                /* :148 (after all parents were drawn) */
#define ROW(i) (rows + (size_t) slot[i] * (size_t) n)
#define EVAL(i, knd) do { \
        fit[i] = f((unsigned) n, ROW(i), NULL, f_data); \
        ++stop->nevals; \
        if (trace) { if (trace->len < trace->cap) { trace->rec[trace->len].f = fit[i]; trace->rec[trace->len].row = slot[i]; \
                     trace->rec[trace->len].kind = (knd); trace->rec[trace->len].accepted = 0; } ++trace->len; } \
        if (*minf > fit[i]) { *minf = fit[i]; memcpy(x, ROW(i), sizeof(double) * (size_t) n); } \
        if (stop->force_stop) ret = ORC_FORCED_STOP; \
        else if (*minf < stop->minf_max) ret = ORC_STOPVAL_REACHED; \
        else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED; \
        else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED; \
    } while (0)
    /* note: the reference draws the parents, overwrites parent 0 with x, THEN draws the offspring — the stream order is
     * the same as drawing all np + no rows first, because the overwrite draws nothing */
    for (id = 0; id < np && ret == ORC_SUCCESS; ++id) EVAL(id, 0);
    while (ret == ORC_SUCCESS) {
        int64_t total, c;
        for (id = 0; id < no; ++id) {                                            /* crossover (:192-203) */
            const int64_t p1 = orc_iurand((int) np), p2 = orc_iurand((int) np);
            const int64_t cross = orc_iurand(n);
            double *o = ROW(np + id);
            for (item = 0; item < cross; ++item) o[item] = ROW(p1)[item];
            for (item = cross; item < n; ++item) o[item] = ROW(p2)[item];
        }
        total = (int64_t) (int) (((unsigned) no * (unsigned) n) / 10);           /* :207, unsigned product */
        if (total < 1) total = 1;
        for (c = 0; c < total; ++c) {                                            /* mutation (:209-218) */
            const int64_t io = orc_iurand((int) no), ip = orc_iurand(n);
            ROW(np + io)[ip] = randcauchy(lb[ip], ub[ip]);
        }
        for (id = 0; id < no && ret == ORC_SUCCESS; ++id) EVAL(np + id, 1);      /* :222-238 */
        if (ret != ORC_SUCCESS) break;
        for (i = 0; i < np + no; ++i) order[i] = i;                              /* selection (:243-251) */
        g_fit = fit;
        qsort(order, (size_t) (np + no), sizeof *order, cmp_stable);
        for (i = 0; i < np + no; ++i) { slot2[i] = slot[order[i]]; fit2[i] = fit[order[i]]; }
        memcpy(slot, slot2, sizeof(int64_t) * (size_t) (np + no));
        memcpy(fit, fit2, sizeof(double) * (size_t) (np + no));
    }
    free(rows); free(fit); free(fit2); free(slot); free(slot2); free(order);
    return ret;
}
