/* cobyla_host.c — NLOPT_LN_COBYLA behind the reference's entry point cobyla_minimize (src/algs/cobyla/cobyla.c:181-271):
 * Powell's COBYLA (linear models of f and of every constraint on a simplex of n+1 points, a trust-region LP step, a merit
 * function with an adaptive penalty) with the reference's modifications: bounds as linear constraints AND enforced on every
 * point (ENFORCE_BOUNDS), pseudo-random simplex-repair steps from an LCG, rho doubled after a well-predicted step,
 * NLopt's stopping criteria, coordinates rescaled by the initial steps.
 *
 * Why it is here: NLOPT_GN_MLSL / GN_MLSL_LDS run their local searches with LN_COBYLA by default
 * (optimize.c:763-768).  COBYLA is a serial algorithm of one objective call per iteration around O(n^2) host arithmetic — there
 * is little for a GPU in ONE search — so with a host callback it runs on the HOST, calling the objective on the caller's thread exactly as
 * the reference does; MLSL's sampling, distances and bookkeeping stay on the device (mlsl_driver.c).  The sequence of points is the
 * reference's evaluation by evaluation: every sum is formed in the reference's order (citations per block), no FMA.
 * The algorithm itself is cobyla_core.h — one source for this file and for the batched device kernel (hip/cobyla_kernels.hip: GN_MLSL's
 * local searches with a compiled-in device objective, one workgroup per start).
 *
 * Layout (0-based, column-major like the reference's Fortran heritage, but named):
 *   SIM(i, j)   j < n: displacement of vertex j from the pole, j == n: the pole (best vertex)     cobyla.c:493-497
 *   SIMI(j, i)  inverse of the displacement matrix
 *   DAT(k, j)   values at vertex j: k < m constraints, k == m the objective, k == m+1 the greatest violation
 *   A(i, k)     gradient of the linear model of constraint k; column m = MINUS the objective's gradient
 */
#include "nla_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef struct {
    nlopt_func f; void *f_data;
    unsigned m_nl; const nla_constraint *fc;      /* inequality constraints fc(x) <= 0 */
    unsigned p; const nla_constraint *h;          /* equality constraints h(x) = 0: two rows each */
    const double *lb, *ub;                        /* scaled bounds */
    const double *scale;
    double *xtmp;
    const double *con_tol;
    const nla_stopping *stop;
} cob_problem;

static void eval_constraint(double *res, const nla_constraint *c, unsigned n, const double *x)   /* nlopt_eval_constraint, general.c */
{
    if (c->f) res[0] = c->f(n, x, NULL, c->f_data);
    else c->mf(c->m, res, n, x, NULL, c->f_data);
}

/* f and the constraint rows at the scaled point x (cobyla.c:79-124): the objective sees the point clipped to the box and
 * unscaled; rows = -fc, (h, -h), then x - lb / ub - x of the UNclipped point for every finite bound.  Returns 1 on a forced stop. */
static int cob_eval(const cob_problem *P, unsigned n, const double *x, double *f, double *con)
{
    unsigned i = 0, j, k;
    for (j = 0; j < n; ++j) P->xtmp[j] = x[j] < P->lb[j] ? P->lb[j] : (x[j] > P->ub[j] ? P->ub[j] : x[j]);
    for (j = 0; j < n; ++j) P->xtmp[j] = P->xtmp[j] * P->scale[j];
    *f = P->f(n, P->xtmp, NULL, P->f_data);
    if (nla_stop_forced(P->stop)) return 1;
    for (j = 0; j < P->m_nl; ++j) {
        eval_constraint(con + i, P->fc + j, n, P->xtmp);
        if (nla_stop_forced(P->stop)) return 1;
        for (k = 0; k < P->fc[j].m; ++k) con[i + k] = -con[i + k];
        i += P->fc[j].m;
    }
    for (j = 0; j < P->p; ++j) {
        eval_constraint(con + i, P->h + j, n, P->xtmp);
        if (nla_stop_forced(P->stop)) return 1;
        for (k = 0; k < P->h[j].m; ++k) con[(i + P->h[j].m) + k] = -con[i + k];
        i += 2 * P->h[j].m;
    }
    for (j = 0; j < n; ++j) {
        if (!nla_isinf(P->lb[j])) con[i++] = x[j] - P->lb[j];
        if (!nla_isinf(P->ub[j])) con[i++] = P->ub[j] - x[j];
    }
    return 0;
}

#define COB_FN static
#define COB_FABS fabs
#define COB_SQRT sqrt
#define COB_ISINF nla_isinf
#define COB_HUGE HUGE_VAL
#include "cobyla_core.h"

/* the main iteration (Powell's COBYLB with the reference's changes, cobyla.c:452-1244) around the caller's callbacks */
static nlopt_result cob_iterate(const cob_problem *P, int n, int m, double *x, double *minf, double rhobeg, double rhoend,
                                const nla_stopping *stop, const double *lb, const double *ub)
{
    cob_state C;
    cob_stop st;
    double *buf = (double *) calloc(cob_core_doubles(n, m), sizeof(double));
    int *iact = (int *) malloc(sizeof(int) * cob_core_ints(m));
    *minf = HUGE_VAL;
    if (!buf || !iact) { free(buf); free(iact); return NLOPT_OUT_OF_MEMORY; }
    cob_core_init(&C, n, m, buf, iact, x, lb, ub, P->con_tol, rhobeg, rhoend);
    st.minf_max = stop->minf_max; st.ftol_rel = stop->ftol_rel; st.ftol_abs = stop->ftol_abs; st.maxeval = stop->maxeval;
    for (;;) {
        st.nevals = *stop->nevals_p; st.forced = nla_stop_forced(stop); st.timed = !st.forced && st.nevals > 0 && nla_stop_time(stop);
        if (!cob_core_advance(&C, &st)) break;
        *stop->nevals_p = st.nevals;
        if (cob_eval(P, (unsigned) n, x, &C.f, C.W.con)) { C.rc = NLOPT_FORCED_STOP; C.go = FINISH_POLE; }    /* a stop forced inside the callback (cobyla.c:587) */
    }
    *minf = C.minf;
    free(buf); free(iact);
    return (nlopt_result) C.rc;
}

/* reference-shaped entry: cobyla_minimize(n, f, f_data, m, fc, p, h, lb, ub, x, minf, stop, dx) (cobyla.c:181-271) */
nlopt_result nla_cobyla_minimize(unsigned n, nlopt_func f, void *f_data, unsigned m, const nla_constraint *fc, unsigned p, const nla_constraint *h,
                                 const double *lb, const double *ub, double *x, double *minf, nla_stopping *stop, const double *dx)
{
    cob_problem P;
    double *scale = NULL, *slb = NULL, *sub = NULL, *xtmp = NULL, *con_tol = NULL, rhobeg, rhoend;
    unsigned i, j, mtot;
    nlopt_result ret;

    memset(&P, 0, sizeof P);
    if (n == 0) { *stop->nevals_p = 0; return NLOPT_SUCCESS; }
    scale = (double *) malloc(sizeof(double) * n);                /* nlopt_compute_rescaling (rescale.c:30-48) */
    slb = (double *) malloc(sizeof(double) * n);
    sub = (double *) malloc(sizeof(double) * n);
    xtmp = (double *) malloc(sizeof(double) * n);
    if (!scale || !slb || !sub || !xtmp) { ret = NLOPT_OUT_OF_MEMORY; goto done; }
    for (i = 0; i < n; ++i) scale[i] = 1.0;
    if (n > 1) {
        for (i = 1; i < n && dx[i] == dx[i - 1]; ++i) { }
        if (i < n) for (i = 1; i < n; ++i) scale[i] = dx[i] / dx[0];
    }
    for (j = 0; j < n; ++j)
        if (scale[j] == 0 || !isfinite(scale[j])) {
            nla_stop_msg(stop, "invalid scaling %g of dimension %d: possible over/underflow?", scale[j], (int) j);
            ret = NLOPT_INVALID_ARGS; goto done;
        }
    for (j = 0; j < n; ++j) { slb[j] = lb[j] / scale[j]; sub[j] = ub[j] / scale[j]; }
    for (j = 0; j < n; ++j) if (slb[j] > sub[j]) { const double t = slb[j]; slb[j] = sub[j]; sub[j] = t; }    /* a negative scale flips them */
    rhobeg = fabs(dx[0] / scale[0]);
    rhoend = stop->xtol_rel * rhobeg;
    if (stop->xtol_abs)
        for (j = 0; j < n; ++j) if (rhoend < stop->xtol_abs[j] / fabs(scale[j])) rhoend = stop->xtol_abs[j] / fabs(scale[j]);
    mtot = 0;
    for (i = 0; i < m; ++i) mtot += fc[i].m;
    for (i = 0; i < p; ++i) mtot += 2 * h[i].m;
    for (j = 0; j < n; ++j) { if (!nla_isinf(lb[j])) ++mtot; if (!nla_isinf(ub[j])) ++mtot; }
    con_tol = (double *) calloc(mtot ? mtot : 1, sizeof(double));
    if (!con_tol) { ret = NLOPT_OUT_OF_MEMORY; goto done; }
    for (j = i = 0; i < m; ++i) { unsigned k; for (k = 0; k < fc[i].m; ++k) con_tol[j++] = fc[i].tol[k]; }
    for (i = 0; i < p; ++i) { unsigned k, r; for (r = 0; r < 2; ++r) for (k = 0; k < h[i].m; ++k) con_tol[j++] = h[i].tol[k]; }
    P.f = f; P.f_data = f_data; P.m_nl = m; P.fc = fc; P.p = p; P.h = h; P.lb = slb; P.ub = sub; P.scale = scale; P.xtmp = xtmp;
    P.con_tol = con_tol; P.stop = stop;
    for (j = 0; j < n; ++j) x[j] = x[j] / scale[j];
    *stop->nevals_p = 0;                                          /* cobyla.c:390 */
    ret = cob_iterate(&P, (int) n, (int) mtot, x, minf, rhobeg, rhoend, stop, slb, sub);
    for (j = 0; j < n; ++j) x[j] = x[j] * scale[j];
    for (j = 0; j < n; ++j) { if (x[j] < lb[j]) x[j] = lb[j]; if (x[j] > ub[j]) x[j] = ub[j]; }       /* cobyla.c:259-263 */
done:
    free(con_tol); free(xtmp); free(sub); free(slb); free(scale);
    return ret;
}
