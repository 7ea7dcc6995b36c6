/* port_isres.c — CPU ORACLE (test infrastructure): Improved Stochastic Ranking Evolution Strategy.
 *
 * Serial restatement of src/algs/isres/isres.c:60-281 with 64-bit indexing.  Same loop order and
 * the same RNG consumption order as the reference: initial population
 * k-major (:122-127, row 0 drawn and then overwritten :128); per generation the evaluation loop
 * with the best-update predicate and stop tests after EVERY candidate (:134-199), the selection
 * (:202-229: all-feasible -> sort by fval with ties in index order — the reference calls glibc's
 * qsort_r, a stable merge sort, on the identity permutation (qsort_r.c:190); otherwise <= pop
 * sweeps of adjacent compare-exchange, each step drawing u = urand(0,1) unconditionally :210),
 * the standard mutation of the non-survivors (:234-252) and the differential variation of the
 * survivors (:253-280, including its reads of PHYSICAL rows 0 and k+1).
 * Constraints are scalar (m == 1 each), as in the configurations of SURVEY.md §8d.
 */
#include "port_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int isinf_(double x) { return fabs(x) >= HUGE_VAL * 0.99; }      /* nlopt_isinf, general.c */

typedef struct { const double *key; } sortctx;
static const double *g_keys;
static int key_cmp(const void *a_, const void *b_)                        /* isres.c:48-54 + stable tie-break */
{
    const int64_t a = *(const int64_t *) a_, b = *(const int64_t *) b_;
    if (g_keys[a] < g_keys[b]) return -1;
    if (g_keys[a] > g_keys[b]) return +1;
    return a < b ? -1 : (a > b ? +1 : 0);
}

int orc_isres_minimize(int n, orc_func f, void *f_data, int m, const orc_constraint *fc, int p, const orc_constraint *h,
                       const double *lb, const double *ub, double *x, double *minf, orc_stop *stop, long population,
                       orc_isres_trace *trace)
{
    const double ALPHA = 0.2, GAMMA = 0.85, PHI = 1.0, PF = 0.45, SURVIVOR = 1.0 / 7.0;   /* isres.c:69-73 */
    int64_t survivors, pop, k, i, j;
    int ret = ORC_SUCCESS, c;
    double *sigmas, *xs, *fval, *penalty, *x0;
    int64_t *irank;
    double minf_penalty = HUGE_VAL, minf_gpenalty = HUGE_VAL, taup, tau;

    *minf = HUGE_VAL;
    if (!population) population = 20 * ((long) n + 1);
    if (population < 1) return ORC_INVALID_ARGS;
    pop = population;
    survivors = (int64_t) ceil(pop * SURVIVOR);
    taup = PHI / sqrt(2 * n);
    tau = PHI / sqrt(2 * sqrt(n));
    for (j = 0; j < n; ++j) if (isinf_(lb[j]) || isinf_(ub[j])) return ORC_INVALID_ARGS;

    sigmas = (double *) malloc(sizeof(double) * ((size_t) pop * n * 2 + (size_t) pop * 2 + n));
    irank = (int64_t *) malloc(sizeof(int64_t) * (size_t) pop);
    if (!sigmas || !irank) { free(sigmas); free(irank); return ORC_OUT_OF_MEMORY; }
    xs = sigmas + (size_t) pop * n; fval = xs + (size_t) pop * n; penalty = fval + pop; x0 = penalty + pop;

    for (k = 0; k < pop; ++k)
        for (j = 0; j < n; ++j) {
            sigmas[k * n + j] = (ub[j] - lb[j]) / sqrt(n);
            xs[k * n + j] = orc_urand(lb[j], ub[j]);
        }
    memcpy(xs, x, sizeof(double) * n);

    for (;;) {
        int all_feasible = 1;
        for (k = 0; k < pop; ++k) {
            int feasible = 1;
            double gpenalty;
            ++stop->nevals;
            fval[k] = f((unsigned) n, xs + k * n, NULL, f_data);
            if (stop->force_stop) { ret = ORC_FORCED_STOP; goto done; }
            penalty[k] = 0;
            for (c = 0; c < m; ++c) {
                double gval = fc[c].f((unsigned) n, xs + k * n, NULL, fc[c].f_data);
                if (gval > fc[c].tol) feasible = 0;
                if (gval < 0) gval = 0;
                penalty[k] += gval * gval;
            }
            gpenalty = penalty[k];
            for (c = 0; c < p; ++c) {
                double hval = h[c].f((unsigned) n, xs + k * n, NULL, h[c].f_data);
                if (fabs(hval) > h[c].tol) feasible = 0;
                penalty[k] += hval * hval;
            }
            if (penalty[k] > 0) all_feasible = 0;
            if (trace && trace->len < trace->cap) {
                trace->f[trace->len] = fval[k];
                trace->pen[trace->len] = penalty[k];
            }
            if (trace) ++trace->len;

            if ((penalty[k] <= minf_penalty || feasible) && (fval[k] <= *minf || minf_gpenalty > 0)
                && ((feasible ? 0 : penalty[k]) != minf_penalty || fval[k] != *minf)) {          /* :174-177 */
                if (fval[k] < stop->minf_max && feasible) ret = ORC_STOPVAL_REACHED;
                else if (!isinf_(*minf)) {
                    if (orc_stop_f(stop, fval[k], *minf) && orc_stop_f(stop, feasible ? 0 : penalty[k], minf_penalty))
                        ret = ORC_FTOL_REACHED;
                    else if (orc_stop_x(stop, xs + k * n, x)) ret = ORC_XTOL_REACHED;
                }
                memcpy(x, xs + k * n, sizeof(double) * n);
                *minf = fval[k];
                minf_penalty = feasible ? 0 : penalty[k];
                minf_gpenalty = feasible ? 0 : gpenalty;
                if (ret != ORC_SUCCESS) goto done;
            }
            if (stop->force_stop) ret = ORC_FORCED_STOP;
            else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED;
            else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED;
            if (ret != ORC_SUCCESS) goto done;
        }

        for (k = 0; k < pop; ++k) irank[k] = k;
        if (all_feasible) { g_keys = fval; qsort(irank, (size_t) pop, sizeof(int64_t), key_cmp); }
        else {
            for (i = 0; i < pop; ++i) {
                int swapped = 0;
                for (j = 0; j < pop - 1; ++j) {
                    double u = orc_urand(0, 1);
                    if (u < PF || (penalty[irank[j]] == 0 && penalty[irank[j + 1]] == 0)) {
                        if (fval[irank[j]] > fval[irank[j + 1]]) { int64_t t = irank[j]; irank[j] = irank[j + 1]; irank[j + 1] = t; swapped = 1; }
                    } else if (penalty[irank[j]] > penalty[irank[j + 1]]) { int64_t t = irank[j]; irank[j] = irank[j + 1]; irank[j + 1] = t; swapped = 1; }
                }
                if (!swapped) break;
            }
        }
        if (trace) ++trace->generations;

        for (k = survivors; k < pop; ++k) {                                /* standard mutation :234-252 */
            double taup_rand = taup * orc_nrand(0, 1);
            int64_t rk = irank[k], ri;
            i = k % survivors;
            ri = irank[i];
            for (j = 0; j < n; ++j) {
                double sigmamax = (ub[j] - lb[j]) / sqrt(n);
                sigmas[rk * n + j] = sigmas[ri * n + j] * exp(taup_rand + tau * orc_nrand(0, 1));
                if (sigmas[rk * n + j] > sigmamax) sigmas[rk * n + j] = sigmamax;
                do {
                    xs[rk * n + j] = xs[ri * n + j] + sigmas[rk * n + j] * orc_nrand(0, 1);
                } while (xs[rk * n + j] < lb[j] || xs[rk * n + j] > ub[j]);
                sigmas[rk * n + j] = sigmas[ri * n + j] + ALPHA * (sigmas[rk * n + j] - sigmas[ri * n + j]);
            }
        }
        memcpy(x0, xs, n * sizeof(double));
        for (k = 0; k < survivors; ++k) {                                  /* differential variation :253-280 */
            double taup_rand = taup * orc_nrand(0, 1);
            int64_t rk = irank[k];
            for (j = 0; j < n; ++j) {
                double xi = xs[rk * n + j];
                if (k + 1 < survivors) xs[rk * n + j] += GAMMA * (x0[j] - xs[(k + 1) * n + j]);
                if (k + 1 == survivors || xs[rk * n + j] < lb[j] || xs[rk * n + j] > ub[j]) {
                    double sigmamax = (ub[j] - lb[j]) / sqrt(n);
                    double sigi = sigmas[rk * n + j];
                    sigmas[rk * n + j] *= exp(taup_rand + tau * orc_nrand(0, 1));
                    if (sigmas[rk * n + j] > sigmamax) sigmas[rk * n + j] = sigmamax;
                    do {
                        xs[rk * n + j] = xi + sigmas[rk * n + j] * orc_nrand(0, 1);
                    } while (xs[rk * n + j] < lb[j] || xs[rk * n + j] > ub[j]);
                    sigmas[rk * n + j] = sigi + ALPHA * (sigmas[rk * n + j] - sigi);
                }
            }
        }
    }
done:
    free(irank);
    free(sigmas);
    return ret;
}
