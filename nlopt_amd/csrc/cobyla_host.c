/* cobyla_host.c — NLOPT_LN_COBYLA behind the reference's entry point cobyla_minimize (src/algs/cobyla/cobyla.c:181-271):
 * Powell's COBYLA (linear models of f and of every constraint on a simplex of n+1 points, a trust-region LP step, a merit
 * function with an adaptive penalty) with the reference's modifications: bounds as linear constraints AND enforced on every
 * point (ENFORCE_BOUNDS), pseudo-random simplex-repair steps from an LCG, rho doubled after a well-predicted step,
 * NLopt's stopping criteria, coordinates rescaled by the initial steps.
 *
 * Why it is here: NLOPT_GN_MLSL / GN_MLSL_LDS run their local searches with LN_COBYLA by default
 * (optimize.c:763-768).  COBYLA is a serial algorithm of one objective call per iteration around O(n^2) host arithmetic — there
 * is nothing for a GPU in it — so it runs on the HOST, calling the objective on the caller's thread exactly as the reference
 * does; MLSL's sampling, distances and bookkeeping stay on the device (mlsl_driver.c).  The sequence of points is the
 * reference's evaluation by evaluation: every sum below is formed in the reference's order (citations per block), no FMA.
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

/* the reference's deterministic LCG for the simplex-repair steps (cobyla.c:300-309) */
static double lcg_between(uint32_t *seed, double a, double b)
{
    *seed = *seed * 1103515245u + 12345u;
    return a + *seed * (b - a) / ((uint32_t) -1);
}

/* ============================================================================================================
 * The trust-region LP (Powell's TRSTLP, cobyla.c:1247-1872): stage one finds the shortest dx, |dx| <= rho, that minimises
 * the greatest violation of  a_k . dx >= b_k ; stage two uses what is left of the trust region to reduce the
 * objective (-a_m . dx) without increasing that violation.  Active set with an orthogonal basis Z kept by Givens rotations.
 * ============================================================================================================ */
typedef struct {
    int n, m;
    const double *a, *b;        /* A(i,k) = a[k*n+i], k <= m; b[k], k < m (b[m] is the objective's slot, zero) */
    double rho;
    double *dx;
    double *z, *zdota, *vmultc, *sdirn, *dxnew, *vmultd;
    int *iact;
    int nact, mcon;
} lp_state;

#define ZC(S, k) ((S)->z + (size_t) (k) * (size_t) (S)->n)          /* column k of Z */
#define AC(S, k) ((S)->a + (size_t) (k) * (size_t) (S)->n)          /* gradient of constraint k */

/* "is this scalar product more than its own rounding noise?" (the acca / accb device, e.g. cobyla.c:1422-1426) */
static int lp_significant(double sum, double sumabs, double c1, double c2)
{
    const double acca = sumabs + fabs(sum) * c1, accb = sumabs + fabs(sum) * c2;
    return sumabs < acca && acca < accb;
}
/* the same test as the reference writes it where a sum is to be ZEROED (e.g. cobyla.c:1424): not the negation of the above when
 * a NaN is involved — a NaN sum is noise for neither form, and runs that have gone NaN must still match the reference's */
static int lp_noise(double sum, double sumabs, double c1, double c2)
{
    const double acca = sumabs + fabs(sum) * c1, accb = sumabs + fabs(sum) * c2;
    return sumabs >= acca || acca >= accb;
}

/* rotate columns k, k+1 of Z so that active constraint k+1 takes position k (cobyla.c:1524-1551 and :1628-1655: the same
 * operation written twice in the reference): moves the constraint at position `from` to the end of the active set */
static void lp_move_to_end(lp_state *S, int from)
{
    const int n = S->n;
    const int isave = S->iact[from];
    const double vsave = S->vmultc[from];
    int k = from, i;
    while (k < S->nact - 1) {
        const int kp = k + 1, kw = S->iact[kp];
        double sp = 0., temp, alpha, beta;
        double *zk = ZC(S, k), *zkp = ZC(S, kp);
        const double *akw = AC(S, kw);
        for (i = 0; i < n; ++i) sp += zk[i] * akw[i];
        temp = sqrt(sp * sp + S->zdota[kp] * S->zdota[kp]);
        alpha = S->zdota[kp] / temp;
        beta = sp / temp;
        S->zdota[kp] = alpha * S->zdota[k];
        S->zdota[k] = temp;
        for (i = 0; i < n; ++i) {
            temp = alpha * zkp[i] + beta * zk[i];
            zkp[i] = alpha * zk[i] - beta * zkp[i];
            zk[i] = temp;
        }
        S->iact[k] = kw;
        S->vmultc[k] = S->vmultc[kp];
        k = kp;
    }
    S->iact[k] = isave;
    S->vmultc[k] = vsave;
}

/* returns NLOPT_SUCCESS or NLOPT_ROUNDOFF_LIMITED; *ifull = 0 if dx could not reach the length rho */
static nlopt_result cob_trust_lp(lp_state *S, int *ifull)
{
    const int n = S->n, m = S->m;
    const double tiny = (double) 1e-6f, c1f = (double) .1f, c2f = (double) .2f;   /* the reference writes these three as float literals */
    double resmax = 0., resold = 0., optold = 0., optnew, stpful, step, ratio, temp, tot;
    int icon = -1, icount = 0, nactx = 0, i, k, kk;
    enum { RESET_COUNT, ITERATE, STAGE_TWO, STUCK } phase;

    *ifull = 1;
    S->mcon = m;
    S->nact = 0;
    for (i = 0; i < n; ++i) {
        for (k = 0; k < n; ++k) S->z[(size_t) k * n + i] = 0.;
        S->z[(size_t) i * n + i] = 1.;
        S->dx[i] = 0.;
    }
    for (k = 0; k < m; ++k) if (S->b[k] > resmax) { resmax = S->b[k]; icon = k; }      /* cobyla.c:1341-1354 */
    for (k = 0; k < m; ++k) { S->iact[k] = k; S->vmultc[k] = resmax - S->b[k]; }
    if (resmax == 0.) phase = STAGE_TWO;
    else { for (i = 0; i < n; ++i) S->sdirn[i] = 0.; phase = RESET_COUNT; }

    for (;;) {
        if (phase == STUCK) {                                     /* L490 */
            if (S->mcon == m) phase = STAGE_TWO;
            else { *ifull = 0; return NLOPT_SUCCESS; }
        }
        if (phase == STAGE_TWO) {                                 /* L480 */
            S->mcon = m + 1;
            icon = m;
            S->iact[m] = m;
            S->vmultc[m] = 0.;
            phase = RESET_COUNT;
        }
        if (phase == RESET_COUNT) { optold = 0.; icount = 0; phase = ITERATE; }       /* L60 */

        /* ---- L70: cycling guard (cobyla.c:1363-1394) ---- */
        if (S->mcon == m) optnew = resmax;
        else { const double *am = AC(S, m); optnew = 0.; for (i = 0; i < n; ++i) optnew -= S->dx[i] * am[i]; }
        if (icount == 0 || optnew < optold) { optold = optnew; nactx = S->nact; icount = 3; }
        else if (S->nact > nactx) { nactx = S->nact; icount = 3; }
        else if (--icount == 0) { phase = STUCK; continue; }

        if (icon >= S->nact) {
            /* ---- add constraint iact[icon] to the active set (cobyla.c:1396-1457) ---- */
            kk = S->iact[icon];
            for (i = 0; i < n; ++i) S->dxnew[i] = AC(S, kk)[i];
            tot = 0.;
            for (k = n - 1; k >= S->nact; --k) {
                double sp = 0., spabs = 0.;
                double *zk = ZC(S, k);
                for (i = 0; i < n; ++i) { temp = zk[i] * S->dxnew[i]; sp += temp; spabs += fabs(temp); }
                if (lp_noise(sp, spabs, .1, .2)) sp = 0.;
                if (tot == 0.) tot = sp;
                else {
                    double *zkp = ZC(S, k + 1), alpha, beta;
                    temp = sqrt(sp * sp + tot * tot);
                    alpha = sp / temp;
                    beta = tot / temp;
                    tot = temp;
                    for (i = 0; i < n; ++i) {
                        temp = alpha * zk[i] + beta * zkp[i];
                        zkp[i] = alpha * zkp[i] - beta * zk[i];
                        zk[i] = temp;
                    }
                }
            }
            if (tot != 0.) {                                      /* room in the active set */
                S->zdota[S->nact] = tot;
                S->vmultc[icon] = S->vmultc[S->nact];
                S->vmultc[S->nact] = 0.;
                ++S->nact;
            } else {
                /* the new gradient is a combination of the active ones: one of them has to leave (cobyla.c:1459-1565) */
                ratio = -1.;
                for (k = S->nact - 1; k >= 0; --k) {
                    double zdotv = 0., zdvabs = 0.;
                    const double *zk = ZC(S, k);
                    for (i = 0; i < n; ++i) { temp = zk[i] * S->dxnew[i]; zdotv += temp; zdvabs += fabs(temp); }
                    if (lp_significant(zdotv, zdvabs, .1, .2)) {
                        temp = zdotv / S->zdota[k];
                        if (temp > 0. && S->iact[k] < m) {
                            const double tempa = S->vmultc[k] / temp;
                            if (ratio < 0. || tempa < ratio) ratio = tempa;
                        }
                        if (k >= 1) { const double *akw = AC(S, S->iact[k]); for (i = 0; i < n; ++i) S->dxnew[i] -= temp * akw[i]; }
                        S->vmultd[k] = temp;
                    } else S->vmultd[k] = 0.;
                }
                if (ratio < 0.) { phase = STUCK; continue; }
                for (k = 0; k < S->nact; ++k) { temp = S->vmultc[k] - ratio * S->vmultd[k]; S->vmultc[k] = 0. >= temp ? 0. : temp; }
                if (icon < S->nact - 1) lp_move_to_end(S, icon);
                temp = 0.;
                { const double *zl = ZC(S, S->nact - 1), *akk = AC(S, kk); for (i = 0; i < n; ++i) temp += zl[i] * akk[i]; }
                if (temp == 0.) { phase = STUCK; continue; }
                S->zdota[S->nact - 1] = temp;
                S->vmultc[icon] = 0.;
                S->vmultc[S->nact - 1] = ratio;
            }
            /* L210: bookkeeping; in stage two the objective stays the LAST active constraint (cobyla.c:1567-1599) */
            {
                const int last = S->nact - 1;
                S->iact[icon] = S->iact[last];
                S->iact[last] = kk;
                if (S->mcon > m && kk != m) {
                    double sp = 0., alpha, beta;
                    double *zk = ZC(S, last - 1), *zl = ZC(S, last);
                    const double *akk = AC(S, kk);
                    k = last - 1;
                    for (i = 0; i < n; ++i) sp += zk[i] * akk[i];
                    temp = sqrt(sp * sp + S->zdota[last] * S->zdota[last]);
                    alpha = S->zdota[last] / temp;
                    beta = sp / temp;
                    S->zdota[last] = alpha * S->zdota[k];
                    S->zdota[k] = temp;
                    for (i = 0; i < n; ++i) {
                        temp = alpha * zl[i] + beta * zk[i];
                        zl[i] = alpha * zk[i] - beta * zl[i];
                        zk[i] = temp;
                    }
                    S->iact[last] = S->iact[k];
                    S->iact[k] = kk;
                    temp = S->vmultc[k];
                    S->vmultc[k] = S->vmultc[last];
                    S->vmultc[last] = temp;
                }
                if (S->mcon == m) {                               /* stage one: next search direction (cobyla.c:1607-1618) */
                    const double *zl = ZC(S, last), *ak = AC(S, S->iact[last]);
                    temp = 0.;
                    for (i = 0; i < n; ++i) temp += S->sdirn[i] * ak[i];
                    temp += -1.;
                    temp /= S->zdota[last];
                    for (i = 0; i < n; ++i) S->sdirn[i] -= temp * zl[i];
                }
            }
        } else {
            /* ---- L260: delete constraint iact[icon] from the active set (cobyla.c:1621-1676) ---- */
            if (icon < S->nact - 1) lp_move_to_end(S, icon);
            --S->nact;
            if (S->mcon == m) {
                const double *zd = ZC(S, S->nact);
                temp = 0.;
                for (i = 0; i < n; ++i) temp += S->sdirn[i] * zd[i];
                for (i = 0; i < n; ++i) S->sdirn[i] -= temp * zd[i];
            }
        }
        if (S->mcon > m) {                                        /* L320: search direction of stage two */
            const double *zl = ZC(S, S->nact - 1);
            temp = 1. / S->zdota[S->nact - 1];
            for (i = 0; i < n; ++i) S->sdirn[i] = temp * zl[i];
        }

        /* ---- L340: step to the trust-region boundary, or the step that takes resmax to zero (cobyla.c:1687-1726) ---- */
        {
            double dd = S->rho * S->rho, sd = 0., ss = 0.;
            for (i = 0; i < n; ++i) {
                if (fabs(S->dx[i]) >= S->rho * tiny) dd -= S->dx[i] * S->dx[i];
                sd += S->dx[i] * S->sdirn[i];
                ss += S->sdirn[i] * S->sdirn[i];
            }
            if (dd <= 0.) { phase = STUCK; continue; }
            temp = sqrt(ss * dd);
            if (fabs(sd) >= temp * tiny) temp = sqrt(ss * dd + sd * sd);
            stpful = dd / (temp + sd);
            step = stpful;
            if (S->mcon == m) {
                const double acca = step + resmax * .1, accb = step + resmax * .2;
                if (step >= acca || acca >= accb) { phase = STAGE_TWO; continue; }
                step = step <= resmax ? step : resmax;
            }
            if (nla_isinf(step)) return NLOPT_ROUNDOFF_LIMITED;
        }
        for (i = 0; i < n; ++i) S->dxnew[i] = S->dx[i] + step * S->sdirn[i];
        if (S->mcon == m) {                                       /* cobyla.c:1737-1750 */
            resold = resmax;
            resmax = 0.;
            for (k = 0; k < S->nact; ++k) {
                const double *ak = AC(S, S->iact[k]);
                temp = S->b[S->iact[k]];
                for (i = 0; i < n; ++i) temp -= ak[i] * S->dxnew[i];
                resmax = resmax >= temp ? resmax : temp;
            }
        }
        /* multipliers the active constraints would have at dxnew (cobyla.c:1752-1785) */
        for (k = S->nact - 1; k >= 0; --k) {
            double zdotw = 0., zdwabs = 0.;
            const double *zk = ZC(S, k);
            for (i = 0; i < n; ++i) { temp = zk[i] * S->dxnew[i]; zdotw += temp; zdwabs += fabs(temp); }
            if (lp_noise(zdotw, zdwabs, .1, .2)) zdotw = 0.;
            S->vmultd[k] = zdotw / S->zdota[k];
            if (k >= 1) { const double *ak = AC(S, S->iact[k]); for (i = 0; i < n; ++i) S->dxnew[i] -= S->vmultd[k] * ak[i]; }
        }
        if (S->mcon > m && S->nact >= 1) { temp = S->vmultd[S->nact - 1]; S->vmultd[S->nact - 1] = 0. >= temp ? 0. : temp; }
        /* residuals of the inactive constraints at dxnew (cobyla.c:1787-1813) */
        for (i = 0; i < n; ++i) S->dxnew[i] = S->dx[i] + step * S->sdirn[i];
        for (k = S->nact; k < S->mcon; ++k) {
            const int id = S->iact[k];
            const double *ak = AC(S, id);
            double sum = resmax - S->b[id], sumabs = resmax + fabs(S->b[id]);
            for (i = 0; i < n; ++i) { temp = ak[i] * S->dxnew[i]; sum += temp; sumabs += fabs(temp); }
            if (lp_noise(sum, sumabs, c1f, c2f)) sum = 0.;
            S->vmultd[k] = sum;
        }
        /* how much of the step can be taken (cobyla.c:1815-1844) */
        ratio = 1.;
        icon = -1;
        for (k = 0; k < S->mcon; ++k)
            if (S->vmultd[k] < 0.) {
                temp = S->vmultc[k] / (S->vmultc[k] - S->vmultd[k]);
                if (temp < ratio) { ratio = temp; icon = k; }
            }
        temp = 1. - ratio;
        for (i = 0; i < n; ++i) S->dx[i] = temp * S->dx[i] + ratio * S->dxnew[i];
        for (k = 0; k < S->mcon; ++k) { const double v = temp * S->vmultc[k] + ratio * S->vmultd[k]; S->vmultc[k] = 0. >= v ? 0. : v; }
        if (S->mcon == m) resmax = resold + ratio * (resmax - resold);
        if (icon >= 0) { phase = ITERATE; continue; }
        if (step == stpful) return NLOPT_SUCCESS;                 /* L500 */
        phase = STAGE_TWO;
    }
}

/* ============================================================================================================
 * The main iteration (Powell's COBYLB with the reference's changes, cobyla.c:452-1244)
 * ============================================================================================================ */
typedef struct {
    int n, m, mp, mpp;
    double *sim, *simi, *dat, *a, *vsig, *veta, *sigbar, *dx, *con, *w;
} cob_work;
#define SIM(i, j)  W.sim[(size_t) (j) * n + (i)]
#define SIMI(j, i) W.simi[(size_t) (i) * n + (j)]
#define DAT(k, j)  W.dat[(size_t) (j) * W.mpp + (k)]
#define ACOL(i, k) W.a[(size_t) (k) * n + (i)]

/* replace vertex jdrop's displacement by dx and update the inverse (cobyla.c:869-897 / :1079-1103: written twice there) */
static void cob_replace_vertex(cob_work *Wp, int jdrop, int after_repair)
{
    cob_work W = *Wp;
    const int n = W.n;
    double temp = 0.;
    int i, j;
    if (!after_repair) for (i = 0; i < n; ++i) { SIM(i, jdrop) = W.dx[i]; temp += SIMI(jdrop, i) * W.dx[i]; }
    else for (i = 0; i < n; ++i) temp += SIMI(jdrop, i) * W.dx[i];          /* (the repair step stored SIM itself, inside its bound fix-up) */
    for (i = 0; i < n; ++i) SIMI(jdrop, i) /= temp;
    for (j = 0; j < n; ++j) {
        if (j == jdrop) continue;
        temp = 0.;
        for (i = 0; i < n; ++i) temp += SIMI(j, i) * W.dx[i];
        for (i = 0; i < n; ++i) SIMI(j, i) -= temp * SIMI(jdrop, i);
    }
}

static nlopt_result cob_iterate(const cob_problem *P, int n, int m, double *x, double *minf, double rhobeg, double rhoend,
                                const nla_stopping *stop, const double *lb, const double *ub)
{
    const double alpha = .25, beta = 2.1, gamma_ = .5, delta = 1.1;
    const int np = n, mp = m, mpp = m + 1;        /* 0-based: the pole's column, the objective's row, the violation's row */
    cob_work W;
    lp_state S;
    double *buf;
    int *iact;
    double rho = rhobeg, parmu = 0., parsig = 0., pareta, prerec = 0., prerem = 0., f = 0., resmax = 0., temp, tempa, sum = 0.;
    int i, j, k, jdrop = np, ibrnch = 0, iflag = 0, ifull = 0, nbest;
    nlopt_result rc = NLOPT_SUCCESS;
    uint32_t seed = (uint32_t) (n + m);
    enum { EVALUATE, POLE, TRUST_STEP, JUDGE, SHRINK, FINISH_POLE, FINISH_HERE } go = EVALUATE;
    size_t need = (size_t) n * (n + 1) + (size_t) n * n + (size_t) (m + 2) * (n + 1) + (size_t) n * (m + 1) + 4 * (size_t) n + (size_t) (m + 2)
                  + (size_t) n * n + (size_t) n + 2 * (size_t) (m + 2) + 2 * (size_t) n + (size_t) n;

    *minf = HUGE_VAL;
    buf = (double *) calloc(need, sizeof(double));
    iact = (int *) malloc(sizeof(int) * (size_t) (m + 2));
    if (!buf || !iact) { free(buf); free(iact); return NLOPT_OUT_OF_MEMORY; }
    memset(&W, 0, sizeof W);
    W.n = n; W.m = m; W.mp = m + 1; W.mpp = m + 2;
    W.sim = buf; W.simi = W.sim + (size_t) n * (n + 1); W.dat = W.simi + (size_t) n * n; W.a = W.dat + (size_t) (m + 2) * (n + 1);
    W.vsig = W.a + (size_t) n * (m + 1); W.veta = W.vsig + n; W.sigbar = W.veta + n; W.dx = W.sigbar + n; W.con = W.dx + n;
    S.n = n; S.m = m; S.a = W.a; S.b = W.con; S.dx = W.dx; S.iact = iact;
    S.z = W.con + (m + 2); S.zdota = S.z + (size_t) n * n; S.vmultc = S.zdota + n; S.sdirn = S.vmultc + (m + 2); S.dxnew = S.sdirn + n;
    S.vmultd = S.dxnew + n;
    W.w = S.vmultd + (m + 2);                                     /* n doubles of scratch for the model gradients */

    /* the initial simplex: the pole at x, vertex i one step along coordinate i, the step kept inside the box (cobyla.c:538-562) */
    for (i = 0; i < n; ++i) {
        double rhocur = rho;
        SIM(i, np) = x[i];
        for (j = 0; j < n; ++j) { SIM(i, j) = 0.; SIMI(i, j) = 0.; }
        if (x[i] + rhocur > ub[i]) {
            if (x[i] - rhocur >= lb[i]) rhocur = -rhocur;
            else if (ub[i] - x[i] > x[i] - lb[i]) rhocur = 0.5 * (ub[i] - x[i]);
            else rhocur = 0.5 * (x[i] - lb[i]);
        }
        SIM(i, i) = rhocur;
        SIMI(i, i) = 1.0 / rhocur;
    }

    for (;;) switch (go) {
    case EVALUATE: {                                              /* L40 (cobyla.c:573-631) */
        int feasible = 1;
        if (nla_stop_forced(stop)) rc = NLOPT_FORCED_STOP;
        else if (*stop->nevals_p > 0) {
            if (nla_stop_evals(stop)) rc = NLOPT_MAXEVAL_REACHED;
            else if (nla_stop_time(stop)) rc = NLOPT_MAXTIME_REACHED;
        }
        if (rc != NLOPT_SUCCESS) { go = FINISH_POLE; break; }
        ++*stop->nevals_p;
        if (cob_eval(P, (unsigned) n, x, &f, W.con)) { rc = NLOPT_FORCED_STOP; go = FINISH_POLE; break; }
        resmax = 0.;
        for (k = 0; k < m; ++k) {
            const double v = -W.con[k];
            resmax = resmax >= v ? resmax : v;
            if (v > P->con_tol[k]) feasible = 0;
        }
        if (f < stop->minf_max && feasible) { rc = NLOPT_MINF_MAX_REACHED; go = FINISH_HERE; break; }
        W.con[mp] = f;
        W.con[mpp] = resmax;
        if (ibrnch == 1) { go = JUDGE; break; }
        /* a vertex of the simplex: its values go into its column (cobyla.c:633-684) */
        for (k = 0; k <= mpp; ++k) DAT(k, jdrop) = W.con[k];
        if (*stop->nevals_p <= n + 1) {
            if (jdrop < n) {                                      /* a vertex of the initial simplex: the better of it and the pole becomes the pole */
                if (DAT(mp, np) <= f) x[jdrop] = SIM(jdrop, np);
                else {
                    const double rhocur = x[jdrop] - SIM(jdrop, np);
                    SIM(jdrop, np) = x[jdrop];
                    for (k = 0; k <= mpp; ++k) { DAT(k, jdrop) = DAT(k, np); DAT(k, np) = W.con[k]; }
                    for (k = 0; k <= jdrop; ++k) {
                        SIM(jdrop, k) = -rhocur;
                        temp = 0.;
                        for (i = k; i <= jdrop; ++i) temp -= SIMI(i, k);
                        SIMI(jdrop, k) = temp;
                    }
                }
            }
            if (*stop->nevals_p <= n) {                           /* next vertex of the initial simplex */
                jdrop = *stop->nevals_p - 1;
                x[jdrop] += SIM(jdrop, jdrop);
                break;                                            /* go == EVALUATE */
            }
        }
        ibrnch = 1;
        go = POLE;
        break;
    }
    case POLE: {                                                  /* L140 (cobyla.c:688-914) */
        double phimin = DAT(mp, np) + parmu * DAT(mpp, np), error = 0.;
        nbest = np;
        for (j = 0; j < n; ++j) {
            temp = DAT(mp, j) + parmu * DAT(mpp, j);
            if (temp < phimin) { nbest = j; phimin = temp; }
            else if (temp == phimin && parmu == 0.) { if (DAT(mpp, j) < DAT(mpp, nbest)) nbest = j; }
        }
        if (nbest < n) {                                          /* the best vertex becomes the pole */
            for (i = 0; i <= mpp; ++i) { temp = DAT(i, np); DAT(i, np) = DAT(i, nbest); DAT(i, nbest) = temp; }
            for (i = 0; i < n; ++i) {
                temp = SIM(i, nbest);
                SIM(i, nbest) = 0.;
                SIM(i, np) += temp;
                tempa = 0.;
                for (k = 0; k < n; ++k) { SIM(i, k) -= temp; tempa -= SIMI(k, i); }
                SIMI(nbest, i) = tempa;
            }
        }
        /* SIMI must still be the inverse (cobyla.c:735-761) */
        for (i = 0; i < n; ++i)
            for (j = 0; j < n; ++j) {
                temp = 0.;
                if (i == j) temp += -1.;
                for (k = 0; k < n; ++k) if (SIM(k, j) != 0) temp += SIMI(i, k) * SIM(k, j);
                error = error >= fabs(temp) ? error : fabs(temp);
            }
        if (error > .1) { rc = NLOPT_ROUNDOFF_LIMITED; go = FINISH_POLE; break; }
        /* gradients of the linear models (cobyla.c:763-787) */
        for (k = 0; k <= mp; ++k) {
            W.con[k] = -DAT(k, np);
            for (j = 0; j < n; ++j) W.w[j] = DAT(k, j) + W.con[k];
            for (i = 0; i < n; ++i) {
                temp = 0.;
                for (j = 0; j < n; ++j) temp += W.w[j] * SIMI(j, i);
                if (k == mp) temp = -temp;
                ACOL(i, k) = temp;
            }
        }
        /* is the simplex acceptable? (cobyla.c:789-811) */
        iflag = 1;
        parsig = alpha * rho;
        pareta = beta * rho;
        for (j = 0; j < n; ++j) {
            double wsig = 0., weta = 0.;
            for (i = 0; i < n; ++i) { wsig += SIMI(j, i) * SIMI(j, i); weta += SIM(i, j) * SIM(i, j); }
            W.vsig[j] = 1. / sqrt(wsig);
            W.veta[j] = sqrt(weta);
            if (W.vsig[j] < parsig || W.veta[j] > pareta) iflag = 0;
        }
        if (ibrnch == 1 || iflag == 1) { go = TRUST_STEP; break; }
        /* a repair step: drop the worst-placed vertex, step orthogonally to the opposite face (cobyla.c:813-914) */
        {
            double cvmaxp = 0., cvmaxm = 0., dxsign = 1.;
            jdrop = -1;
            temp = pareta;
            for (j = 0; j < n; ++j) if (W.veta[j] > temp) { jdrop = j; temp = W.veta[j]; }
            if (jdrop < 0) for (j = 0; j < n; ++j) if (W.vsig[j] < temp) { jdrop = j; temp = W.vsig[j]; }
            temp = gamma_ * rho * W.vsig[jdrop];
            for (i = 0; i < n; ++i) W.dx[i] = temp * SIMI(jdrop, i);
            for (k = 0; k <= mp; ++k) {
                sum = 0.;
                for (i = 0; i < n; ++i) sum += ACOL(i, k) * W.dx[i];
                if (k < mp) {
                    temp = DAT(k, np);
                    cvmaxp = cvmaxp >= -sum - temp ? cvmaxp : -sum - temp;
                    cvmaxm = cvmaxm >= sum - temp ? cvmaxm : sum - temp;
                }
            }
            if (parmu * (cvmaxp - cvmaxm) > sum + sum) dxsign = -1.;
            temp = 0.;
            for (i = 0; i < n; ++i) {
                const double xi = SIM(i, np);
                W.dx[i] = dxsign * W.dx[i] * lcg_between(&seed, 0.01, 1);
                for (;;) {                                        /* keep the new vertex inside the box (cobyla.c:876-889) */
                    if (xi + W.dx[i] > ub[i]) W.dx[i] = -W.dx[i];
                    if (xi + W.dx[i] < lb[i]) {
                        if (xi - W.dx[i] <= ub[i]) W.dx[i] = -W.dx[i];
                        else { W.dx[i] *= 0.5; continue; }
                    }
                    break;
                }
                SIM(i, jdrop) = W.dx[i];
            }
            cob_replace_vertex(&W, jdrop, 1);
            for (j = 0; j < n; ++j) x[j] = SIM(j, np) + W.dx[j];
        }
        go = EVALUATE;
        break;
    }
    case TRUST_STEP: {                                            /* L370 (cobyla.c:918-1011) */
        double resnew = 0., barmu = 0.;
        int again = 0;
        S.rho = rho;
        rc = cob_trust_lp(&S, &ifull);
        if (rc != NLOPT_SUCCESS) { go = FINISH_POLE; break; }
        for (i = 0; i < n; ++i) {                                 /* (paranoia of the reference: the bound rows are linear) */
            const double xi = SIM(i, np);
            if (xi + W.dx[i] > ub[i]) W.dx[i] = ub[i] - xi;
            if (xi + W.dx[i] < lb[i]) W.dx[i] = xi - lb[i];
        }
        if (ifull == 0) {
            temp = 0.;
            for (i = 0; i < n; ++i) temp += W.dx[i] * W.dx[i];
            if (temp < rho * .25 * rho) { ibrnch = 1; go = SHRINK; break; }
        }
        /* predicted change of f and of the greatest violation (cobyla.c:952-967) */
        W.con[mp] = 0.;
        for (k = 0; k <= mp; ++k) {
            sum = W.con[k];
            for (i = 0; i < n; ++i) sum -= ACOL(i, k) * W.dx[i];
            if (k < mp) resnew = resnew >= sum ? resnew : sum;
        }
        /* raise the penalty parameter if necessary; if that changes the pole, start over from there (cobyla.c:969-1001) */
        prerec = DAT(mpp, np) - resnew;
        if (prerec > 0.) barmu = sum / prerec;
        if (parmu < barmu * 1.5) {
            double phi;
            parmu = barmu * 2.;
            phi = DAT(mp, np) + parmu * DAT(mpp, np);
            for (j = 0; j < n && !again; ++j) {
                temp = DAT(mp, j) + parmu * DAT(mpp, j);
                if (temp < phi) again = 1;
                else if (temp == phi && parmu == 0.) { if (DAT(mpp, j) < DAT(mpp, np)) again = 1; }
            }
            if (again) { go = POLE; break; }
        }
        prerem = parmu * prerec - sum;
        for (i = 0; i < n; ++i) x[i] = SIM(i, np) + W.dx[i];
        ibrnch = 1;
        go = EVALUATE;
        break;
    }
    case JUDGE: {                                                 /* L440 (cobyla.c:1012-1124) */
        const double vmold = DAT(mp, np) + parmu * DAT(mpp, np), vmnew = f + parmu * resmax;
        double trured = vmold - vmnew, ratio = 0., edgmax;
        int l = -1;
        if (parmu == 0. && f == DAT(mp, np)) { prerem = prerec; trured = DAT(mpp, np) - resmax; }
        if (trured <= 0.) ratio = 1.;
        jdrop = -1;
        for (j = 0; j < n; ++j) {
            temp = 0.;
            for (i = 0; i < n; ++i) temp += SIMI(j, i) * W.dx[i];
            temp = fabs(temp);
            if (temp > ratio) { jdrop = j; ratio = temp; }
            W.sigbar[j] = temp * W.vsig[j];
        }
        edgmax = delta * rho;
        for (j = 0; j < n; ++j)
            if (W.sigbar[j] >= parsig || W.sigbar[j] >= W.vsig[j]) {
                temp = W.veta[j];
                if (trured > 0.) {
                    temp = 0.;
                    for (i = 0; i < n; ++i) { const double d = W.dx[i] - SIM(i, j); temp += d * d; }
                    temp = sqrt(temp);
                }
                if (temp > edgmax) { l = j; edgmax = temp; }
            }
        if (l >= 0) jdrop = l;
        if (jdrop < 0) { go = SHRINK; break; }
        cob_replace_vertex(&W, jdrop, 0);
        for (k = 0; k <= mpp; ++k) DAT(k, jdrop) = W.con[k];
        if (trured > 0. && trured >= prerem * .1) {
            if (trured >= prerem * 0.9 && trured <= prerem * 1.1 && iflag) rho *= 2.0;      /* the reference's addition (cobyla.c:1112-1122) */
            go = POLE;
            break;
        }
        go = SHRINK;
        break;
    }
    case SHRINK: {                                                /* L550 (cobyla.c:1125-1206) */
        double fbest;
        if (iflag == 0) { ibrnch = 0; go = POLE; break; }
        fbest = ifull == 1 ? f : DAT(mp, np);
        if (fbest < *minf && nla_stop_ftol(stop, fbest, *minf)) { rc = NLOPT_FTOL_REACHED; go = FINISH_POLE; break; }
        *minf = fbest;
        if (rho > rhoend) {
            rho *= .5;
            if (rho <= rhoend * 1.5) rho = rhoend;
            if (parmu > 0.) {
                double denom = 0., cmin = 0., cmax = 0.;
                for (k = 0; k <= mp; ++k) {
                    cmin = DAT(k, np);
                    cmax = cmin;
                    for (i = 0; i < n; ++i) {
                        cmin = cmin <= DAT(k, i) ? cmin : DAT(k, i);
                        cmax = cmax >= DAT(k, i) ? cmax : DAT(k, i);
                    }
                    if (k < m && cmin < cmax * .5) {
                        temp = (cmax >= 0. ? cmax : 0.) - cmin;
                        if (denom <= 0.) denom = temp;
                        else denom = denom <= temp ? denom : temp;
                    }
                }
                if (denom == 0.) parmu = 0.;
                else if (cmax - cmin < parmu * denom) parmu = (cmax - cmin) / denom;
            }
            go = POLE;
            break;
        }
        rc = rhoend > 0 ? NLOPT_XTOL_REACHED : NLOPT_ROUNDOFF_LIMITED;
        go = ifull == 1 ? FINISH_HERE : FINISH_POLE;
        break;
    }
    case FINISH_POLE:                                             /* L600 */
        for (i = 0; i < n; ++i) x[i] = SIM(i, np);
        f = DAT(mp, np);
        /* fall through */
    case FINISH_HERE:                                             /* L620 */
        *minf = f;
        free(buf); free(iact);
        return rc;
    }
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
