/* cobyla_core.h — Powell's COBYLA with the reference's modifications (src/algs/cobyla/cobyla.c:452-1872) as ONE source for the host
 * (cobyla_host.c: LN_COBYLA behind cobyla_minimize for any callback) and the device (hip/cobyla_kernels.hip: GN_MLSL's local searches
 * batched, one workgroup per start).  The algorithm is a state machine around ONE evaluation point
 * (cobyla.c:573, label 40): cob_core_advance() runs it from the values of the last evaluation to the next point that wants one and
 * returns; whoever includes this file evaluates — a host callback, or a workgroup's reduction — and calls it again.
 *
 * Every sum is formed in the reference's order, no FMA (both compilers are run with -ffp-contract=off): given the same f values the
 * sequence of points is the reference's evaluation by evaluation (tests/test_cobyla_differential.py for the host; the device differs
 * by the objective's own rounding only).
 *
 * Layout (0-based, column-major like the reference's Fortran heritage, but named):
 *   SIM(i, j)   j < n: displacement of vertex j from the pole, j == n: the pole (best vertex)     cobyla.c:493-497
 *   SIMI(j, i)  inverse of the displacement matrix
 *   DAT(k, j)   values at vertex j: k < m constraints, k == m the objective, k == m+1 the greatest violation
 *   A(i, k)     gradient of the linear model of constraint k; column m = MINUS the objective's gradient
 *
 * The includer defines COB_FN (function qualifier), COB_FABS / COB_SQRT / COB_ISINF before including. */
#ifndef NLA_COBYLA_CORE_H
#define NLA_COBYLA_CORE_H
#include <stddef.h>
#include <stdint.h>

/* result codes = nlopt_result values (nlopt.h:167-181) */
#define COB_SUCCESS 1
#define COB_MINF_MAX_REACHED 2
#define COB_FTOL_REACHED 3
#define COB_XTOL_REACHED 4
#define COB_MAXEVAL_REACHED 5
#define COB_MAXTIME_REACHED 6
#define COB_FORCED_STOP (-5)
#define COB_ROUNDOFF_LIMITED (-4)

/* what the algorithm reads of the caller's stopping criteria (nlopt-util.h:79-91); forced / timed are the caller's latest readings of
 * nlopt_stop_forced / nlopt_stop_time, refreshed before every call of cob_core_advance */
typedef struct { double minf_max, ftol_rel, ftol_abs; int maxeval, nevals, forced, timed; } cob_stop;

COB_FN int cob_tol_reached(double vold, double vnew, double reltol, double abstol)       /* relstop, stop.c:81-86 */
{
    double d;
    if (COB_ISINF(vold)) return 0;
    d = COB_FABS(vnew - vold);
    return d < abstol || d < reltol * (COB_FABS(vnew) + COB_FABS(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

/* the reference's deterministic LCG for the simplex-repair steps (cobyla.c:300-309) */
COB_FN double lcg_between(uint32_t *seed, double a, double b)
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
COB_FN int lp_significant(double sum, double sumabs, double c1, double c2)
{
    const double acca = sumabs + COB_FABS(sum) * c1, accb = sumabs + COB_FABS(sum) * c2;
    return sumabs < acca && acca < accb;
}
/* the same test as the reference writes it where a sum is to be ZEROED (e.g. cobyla.c:1424): not the negation of the above when
 * a NaN is involved — a NaN sum is noise for neither form, and runs that have gone NaN must still match the reference's */
COB_FN int lp_noise(double sum, double sumabs, double c1, double c2)
{
    const double acca = sumabs + COB_FABS(sum) * c1, accb = sumabs + COB_FABS(sum) * c2;
    return sumabs >= acca || acca >= accb;
}

/* rotate columns k, k+1 of Z so that active constraint k+1 takes position k (cobyla.c:1524-1551 and :1628-1655: the same
 * operation written twice in the reference): moves the constraint at position `from` to the end of the active set */
COB_FN void lp_move_to_end(lp_state *S, int from)
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
        temp = COB_SQRT(sp * sp + S->zdota[kp] * S->zdota[kp]);
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

/* returns COB_SUCCESS or COB_ROUNDOFF_LIMITED; *ifull = 0 if dx could not reach the length rho */
COB_FN int cob_trust_lp(lp_state *S, int *ifull)
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
            else { *ifull = 0; return COB_SUCCESS; }
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
                for (i = 0; i < n; ++i) { temp = zk[i] * S->dxnew[i]; sp += temp; spabs += COB_FABS(temp); }
                if (lp_noise(sp, spabs, .1, .2)) sp = 0.;
                if (tot == 0.) tot = sp;
                else {
                    double *zkp = ZC(S, k + 1), alpha, beta;
                    temp = COB_SQRT(sp * sp + tot * tot);
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
                    for (i = 0; i < n; ++i) { temp = zk[i] * S->dxnew[i]; zdotv += temp; zdvabs += COB_FABS(temp); }
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
                    temp = COB_SQRT(sp * sp + S->zdota[last] * S->zdota[last]);
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
                if (COB_FABS(S->dx[i]) >= S->rho * tiny) dd -= S->dx[i] * S->dx[i];
                sd += S->dx[i] * S->sdirn[i];
                ss += S->sdirn[i] * S->sdirn[i];
            }
            if (dd <= 0.) { phase = STUCK; continue; }
            temp = COB_SQRT(ss * dd);
            if (COB_FABS(sd) >= temp * tiny) temp = COB_SQRT(ss * dd + sd * sd);
            stpful = dd / (temp + sd);
            step = stpful;
            if (S->mcon == m) {
                const double acca = step + resmax * .1, accb = step + resmax * .2;
                if (step >= acca || acca >= accb) { phase = STAGE_TWO; continue; }
                step = step <= resmax ? step : resmax;
            }
            if (COB_ISINF(step)) return COB_ROUNDOFF_LIMITED;
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
            for (i = 0; i < n; ++i) { temp = zk[i] * S->dxnew[i]; zdotw += temp; zdwabs += COB_FABS(temp); }
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
            double sum = resmax - S->b[id], sumabs = resmax + COB_FABS(S->b[id]);
            for (i = 0; i < n; ++i) { temp = ak[i] * S->dxnew[i]; sum += temp; sumabs += COB_FABS(temp); }
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
        if (step == stpful) return COB_SUCCESS;                 /* L500 */
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
COB_FN void cob_replace_vertex(cob_work *Wp, int jdrop, int after_repair)
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

/* the algorithm's state between two evaluations */
typedef struct {
    int n, m;
    cob_work W;
    lp_state S;
    double rho, rhoend, parmu, parsig, prerec, prerem, f, resmax, minf;
    int jdrop, ibrnch, iflag, ifull, rc, go;
    uint32_t seed;
    double *x;                          /* the point to evaluate / the result (scaled coordinates) */
    const double *lb, *ub;              /* scaled bounds */
    const double *con_tol;              /* m feasibility tolerances */
} cob_state;

/* doubles / ints of workspace for n variables and m constraint rows */
COB_FN size_t cob_core_doubles(int n, int m)
{
    return (size_t) n * (n + 1) + (size_t) n * n + (size_t) (m + 2) * (n + 1) + (size_t) n * (m + 1) + 4 * (size_t) n + (size_t) (m + 2)
           + (size_t) n * n + (size_t) n + 2 * (size_t) (m + 2) + 2 * (size_t) n + (size_t) n;
}
COB_FN size_t cob_core_ints(int m) { return (size_t) (m + 2); }

enum { EVAL_PRE, EVAL_POST, POLE, TRUST_STEP, JUDGE, SHRINK, FINISH_POLE, FINISH_HERE, FINISHED };

/* buf: cob_core_doubles(n, m) doubles, ZERO; iact: cob_core_ints(m) ints; x: the start point (scaled), kept by reference */
COB_FN void cob_core_init(cob_state *C, int n, int m, double *buf, int *iact, double *x, const double *lb, const double *ub,
                          const double *con_tol, double rhobeg, double rhoend)
{
    cob_work W;
    lp_state S;
    int i, j;
    const int np = n;
    W.n = n; W.m = m; W.mp = m + 1; W.mpp = m + 2;
    W.sim = buf; W.simi = W.sim + (size_t) n * (n + 1); W.dat = W.simi + (size_t) n * n; W.a = W.dat + (size_t) (m + 2) * (n + 1);
    W.vsig = W.a + (size_t) n * (m + 1); W.veta = W.vsig + n; W.sigbar = W.veta + n; W.dx = W.sigbar + n; W.con = W.dx + n;
    S.n = n; S.m = m; S.a = W.a; S.b = W.con; S.dx = W.dx; S.iact = iact; S.rho = 0.; S.nact = 0; S.mcon = 0;
    S.z = W.con + (m + 2); S.zdota = S.z + (size_t) n * n; S.vmultc = S.zdota + n; S.sdirn = S.vmultc + (m + 2); S.dxnew = S.sdirn + n;
    S.vmultd = S.dxnew + n;
    W.w = S.vmultd + (m + 2);                                     /* n doubles of scratch for the model gradients */
    /* the initial simplex: the pole at x, vertex i one step along coordinate i, the step kept inside the box (cobyla.c:538-562) */
    for (i = 0; i < n; ++i) {
        double rhocur = rhobeg;
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
    C->n = n; C->m = m; C->W = W; C->S = S;
    C->rho = rhobeg; C->rhoend = rhoend; C->parmu = 0.; C->parsig = 0.; C->prerec = 0.; C->prerem = 0.; C->f = 0.; C->resmax = 0.;
    C->minf = COB_HUGE;
    C->jdrop = np; C->ibrnch = 0; C->iflag = 0; C->ifull = 0; C->rc = COB_SUCCESS; C->go = EVAL_PRE;
    C->seed = (uint32_t) (n + m);
    C->x = x; C->lb = lb; C->ub = ub; C->con_tol = con_tol;
}

/* 1: evaluate at C->x (objective -> C->f, the m constraint rows -> C->W.con[0 .. m)) and call again; 0: finished — C->rc, the result in
 * C->x, its value in C->minf */
COB_FN int cob_core_advance(cob_state *C, cob_stop *st)
{
    const double alpha = .25, beta = 2.1, gamma_ = .5, delta = 1.1;
    const int n = C->n, m = C->m, np = n, mp = m, mpp = m + 1;        /* 0-based: the pole's column, the objective's row, the violation's row */
    cob_work W = C->W;
    lp_state S = C->S;
    double *x = C->x;
    const double *lb = C->lb, *ub = C->ub, *con_tol = C->con_tol;
    const double rhoend = C->rhoend;
    double rho = C->rho, parmu = C->parmu, parsig = C->parsig, pareta, prerec = C->prerec, prerem = C->prerem, f = C->f, resmax = C->resmax, temp, tempa,
           sum = 0., minf = C->minf;
    int i, j, k, jdrop = C->jdrop, ibrnch = C->ibrnch, iflag = C->iflag, ifull = C->ifull, nbest, go = C->go;
    int rc = C->rc;
    uint32_t seed = C->seed;
#define COB_STORE() do { C->S = S; C->rho = rho; C->parmu = parmu; C->parsig = parsig; C->prerec = prerec; C->prerem = prerem; C->f = f; C->resmax = resmax; \
        C->minf = minf; C->jdrop = jdrop; C->ibrnch = ibrnch; C->iflag = iflag; C->ifull = ifull; C->rc = rc; C->go = go; C->seed = seed; } while (0)

    for (;;) switch (go) {
    case EVAL_PRE: {                                              /* L40 (cobyla.c:573-631): the stop tests in front of an evaluation */
        if (st->forced) rc = COB_FORCED_STOP;
        else if (st->nevals > 0) {
            if (st->maxeval > 0 && st->nevals >= st->maxeval) rc = COB_MAXEVAL_REACHED;
            else if (st->timed) rc = COB_MAXTIME_REACHED;
        }
        if (rc != COB_SUCCESS) { go = FINISH_POLE; break; }
        ++st->nevals;
        go = EVAL_POST;
        COB_STORE();
        return 1;                                                 /* the caller evaluates at C->x: f -> C->f, rows -> C->W.con[0 .. m) */
    }
    case EVAL_POST: {
        int feasible = 1;
        resmax = 0.;
        for (k = 0; k < m; ++k) {
            const double v = -W.con[k];
            resmax = resmax >= v ? resmax : v;
            if (v > con_tol[k]) feasible = 0;
        }
        if (f < st->minf_max && feasible) { rc = COB_MINF_MAX_REACHED; go = FINISH_HERE; break; }
        W.con[mp] = f;
        W.con[mpp] = resmax;
        if (ibrnch == 1) { go = JUDGE; break; }
        /* a vertex of the simplex: its values go into its column (cobyla.c:633-684) */
        for (k = 0; k <= mpp; ++k) DAT(k, jdrop) = W.con[k];
        if (st->nevals <= n + 1) {
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
            if (st->nevals <= n) {                           /* next vertex of the initial simplex */
                jdrop = st->nevals - 1;
                x[jdrop] += SIM(jdrop, jdrop);
                go = EVAL_PRE; break;
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
                error = error >= COB_FABS(temp) ? error : COB_FABS(temp);
            }
        if (error > .1) { rc = COB_ROUNDOFF_LIMITED; go = FINISH_POLE; break; }
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
            W.vsig[j] = 1. / COB_SQRT(wsig);
            W.veta[j] = COB_SQRT(weta);
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
        go = EVAL_PRE;
        break;
    }
    case TRUST_STEP: {                                            /* L370 (cobyla.c:918-1011) */
        double resnew = 0., barmu = 0.;
        int again = 0;
        S.rho = rho;
        rc = cob_trust_lp(&S, &ifull);
        if (rc != COB_SUCCESS) { go = FINISH_POLE; break; }
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
        go = EVAL_PRE;
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
            temp = COB_FABS(temp);
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
                    temp = COB_SQRT(temp);
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
        if (fbest < minf && cob_tol_reached(minf, fbest, st->ftol_rel, st->ftol_abs)) { rc = COB_FTOL_REACHED; go = FINISH_POLE; break; }
        minf = fbest;
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
        rc = rhoend > 0 ? COB_XTOL_REACHED : COB_ROUNDOFF_LIMITED;
        go = ifull == 1 ? FINISH_HERE : FINISH_POLE;
        break;
    }
    case FINISH_POLE:                                             /* L600 */
        for (i = 0; i < n; ++i) x[i] = SIM(i, np);
        f = DAT(mp, np);
        /* fall through */
    case FINISH_HERE:                                             /* L620 */
        minf = f;
        go = FINISHED;
        COB_STORE();
        return 0;
    case FINISHED:
        return 0;
    }
}
#undef COB_STORE

#endif
