/* port_lbfgs.c — CPU ORACLE (test infrastructure): NLOPT_LD_LBFGS, i.e. Luksan's PLIS, the
 * bound-constrained limited-memory BFGS method based on the Strang recurrences
 * (src/algs/luksan/plis.c:106-510 with pssubs.c / mssubs.c).  Sequential vector loops in the
 * reference's summation order; the scalar line search / termination logic comes from the shared
 * source header nlopt_amd/csrc/lbfgs_scalar.h (like objfuncs.h, one source for oracle, host and
 * device).  Pinned evaluation-by-evaluation against the real reference by tests/test_oracle_pins.py.
 *
 * Storage differs from the reference on purpose: the reference physically shifts all history
 * columns by one every iteration (mxdrsu, mssubs.c:503-524); here the history is a ring — column
 * "i-th newest" is found by index arithmetic — which produces the same numbers.
 */
#include "port_oracle.h"
#include "lbfgs_scalar.h"
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define MEMAVAIL 1310720                 /* luksan.h:149 */

typedef struct {
    int n, mf, head;                     /* head: ring slot of column 1 (the newest pair) */
    double *xo, *go, *uo, *vo;
} hist;
static double *colx(hist *h, int i) { return h->xo + (size_t) ((h->head + i - 1) % h->mf) * h->n; }   /* i = 1..k */
static double *colg(hist *h, int i) { return h->go + (size_t) ((h->head + i - 1) % h->mf) * h->n; }
static double *colu(hist *h, int i) { return h->uo + (h->head + i - 1) % h->mf; }

static double mdot(int n, const double *x, const double *y, const int *ix)        /* mxudot, job > 0 (mssubs.c:692-733) */
{
    double t = 0;
    for (int i = 0; i < n; ++i) if (ix[i] >= 0) t += x[i] * y[i];
    return t;
}
static void maxpy(int n, double a, const double *x, double *z, const int *ix)      /* mxudir z = z + a x (mssubs.c:601-641) */
{
    for (int i = 0; i < n; ++i) if (ix[i] >= 0) z[i] = z[i] + a * x[i];
}
static void mneg(int n, const double *x, double *y, const int *ix)                 /* mxuneg (mssubs.c:749-790) */
{
    for (int i = 0; i < n; ++i) y[i] = ix[i] >= 0 ? -x[i] : 0.;
}
static void project(int n, double *x, const int *ix, const double *xl, const double *xu, double eps9)   /* pcbs04, pssubs.c:26-65 */
{
    for (int i = 0; i < n; ++i) {
        const int t = ix[i] < 0 ? -ix[i] : ix[i];
        if ((t == 1 || t == 3 || t == 4) && x[i] <= xl[i] + eps9 * LB_MAX(fabs(xl[i]), 1.)) x[i] = xl[i];
        if ((t == 2 || t == 3 || t == 4) && x[i] >= xu[i] - eps9 * LB_MAX(fabs(xu[i]), 1.)) x[i] = xu[i];
    }
}
static void add_active(int n, double *x, int *ix, const double *xl, const double *xu)   /* pyadc0, pssubs.c:791-841 */
{
    for (int i = 0; i < n; ++i) {
        const int ii = ix[i], t = ii < 0 ? -ii : ii;
        if (t >= 5) ix[i] = -t;
        else if ((t == 1 || t == 3 || t == 4) && x[i] <= xl[i]) { x[i] = xl[i]; ix[i] = (t == 4) ? -3 : -t; }
        else if ((t == 2 || t == 3 || t == 4) && x[i] >= xu[i]) { x[i] = xu[i]; ix[i] = (t == 3) ? -4 : -t; }
    }
}

int orc_lbfgs_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                       orc_stop *stop, int mf, double tolg)
{
    int *ix, i, k, iold, xstop = 0;
    double *work, *xl, *xu, *gf, *s;
    hist H;
    lb_ls_state lss;
    lb_ls_io q;
    lb_counters c;
    lb_stop ls;
    double gmax = 0, umax = 0, fval, fo, p = 0, po = 0, a, b, gnorm, snorm = 0, rmax, rmin = 0, maxf = 1e20, minf_est = -HUGE_VAL;
    const double eta9 = 1e120, eps8 = 1., eps9 = 1e-8, alf1 = 1e-10, alf2 = 1e10, told = 1e-4, xmax = 1e16;
    int kd = 1, ld = -1, nred = 0, maxst = 0, ret;

    if (mf <= 0) {                                                           /* plis.c:441-445 */
        mf = LB_MAX(MEMAVAIL / n, 10);
        if (stop->maxeval > 0 && stop->maxeval <= mf) mf = LB_MAX((int) stop->maxeval, 1);
    }
    ix = (int *) malloc(sizeof(int) * (size_t) n);
    work = (double *) calloc((size_t) n * 4 + (size_t) n * mf * 2 + (size_t) mf * 2, sizeof(double));
    if (!ix || !work) { free(ix); free(work); return ORC_OUT_OF_MEMORY; }
    xl = work; xu = xl + n; gf = xu + n; s = gf + n;
    H.n = n; H.mf = mf; H.head = 0;
    H.xo = s + n; H.go = H.xo + (size_t) n * mf; H.uo = H.go + (size_t) n * mf; H.vo = H.uo + mf;
    for (i = 0; i < n; ++i) {                                                /* plis.c:463-469 */
        int lbu = lb[i] <= -0.99 * HUGE_VAL, ubu = ub[i] >= 0.99 * HUGE_VAL;
        ix[i] = lbu ? (ubu ? 0 : 2) : (ubu ? 1 : (lb[i] == ub[i] ? 5 : 3));
        xl[i] = lb[i]; xu[i] = ub[i];
    }
    /* tolerances the reference patches into the caller's stop struct (plis.c:196-214,479-482) */
    if (stop->xtol_rel <= 0.) stop->xtol_rel = 1e-16;
    if (stop->ftol_rel <= 0.) stop->ftol_rel = 1e-14;
    if (tolg <= 0.) tolg = 1e-8;
    ls.minf_max = stop->minf_max; ls.ftol_rel = stop->ftol_rel; ls.ftol_abs = stop->ftol_abs; ls.maxeval = (int) stop->maxeval;

    memset(&c, 0, sizeof c);
    memset(&lss, 0, sizeof lss);
    memset(&q, 0, sizeof q);
    c.ites = 1; c.mtesx = 2; c.mtesf = 2; c.iters = 2; c.ires1 = 999; c.ires2 = 0; c.kd = 1;
    c.mit = INT_MAX; c.mfg = stop->maxeval > 0 ? (int) stop->maxeval : INT_MAX;
    c.kit = -(c.ires1 * n + c.ires2);
    rmax = eta9;
    fo = minf_est;

    /* initial operations with simple bounds (plis.c:232-249) */
    for (i = 0; i < n; ++i) {
        if ((ix[i] == 3 || ix[i] == 4) && xu[i] <= xl[i]) { xu[i] = xl[i]; ix[i] = 5; }
        else if (ix[i] == 5 || ix[i] == 6) { xl[i] = x[i]; xu[i] = x[i]; ix[i] = 5; }
    }
    project(n, x, ix, xl, xu, eps9);
    add_active(n, x, ix, xl, xu);
    fval = f((unsigned) n, x, gf, f_data);
    ++stop->nevals;
    ++c.nfg;

    for (;;) {
        /* L11120: multiplier test values (pytrcg, pssubs.c:1139-1184) and termination */
        gmax = 0.; umax = 0.; iold = 0;
        for (i = 0; i < n; ++i) {
            const double t = gf[i];
            if (ix[i] >= 0) gmax = LB_MAX(gmax, fabs(t));
            else if (ix[i] <= -5) { }
            else if ((ix[i] == -1 || ix[i] == -3) && umax + t >= 0.) { }
            else if ((ix[i] == -2 || ix[i] == -4) && umax - t >= 0.) { }
            else { iold = i + 1; umax = fabs(t); }
        }
        c.kd = kd;
        lb_pyfut1(n, fval, &fo, umax, gmax, xstop, &ls, stop->force_stop, (int) stop->nevals, tolg, &c);
        if (c.iterm != 0) break;
        if (rmax > 0.) {                                                     /* pyrmc0, pssubs.c:989-1032 (n == nf here) */
            if (umax > eps8 * gmax) {
                int released = 0;
                for (i = 0; i < n; ++i) {
                    const int t = ix[i];
                    if (t >= 0 || t <= -5) continue;
                    if ((t == -1 || t == -3) && -gf[i] <= 0.) continue;
                    if ((t == -2 || t == -4) && gf[i] <= 0.) continue;
                    ++released;
                    ix[i] = LB_MIN(-t, 3);
                    if (rmax == 0.) break;
                }
                if (released > 1) c.irest = LB_MAX(c.irest, 1);
            }
        }
        (void) iold;
    direction:                                                               /* L11130 */
        gnorm = sqrt(mdot(n, gf, gf, ix));
        if (c.irest == 0) {
            k = LB_MIN(c.nit - c.kit, mf);
            if (k <= 0) c.irest = LB_MAX(c.irest, 1);
            else {
                b = mdot(n, colx(&H, 1), colg(&H, 1), ix);
                if (b <= 0.) c.irest = LB_MAX(c.irest, 1);
                else {
                    *colu(&H, 1) = 1. / b;
                    mneg(n, gf, s, ix);
                    for (i = 1; i <= k; ++i) {                               /* mxdrcb, mssubs.c:353-383 */
                        H.vo[i - 1] = *colu(&H, i) * mdot(n, s, colx(&H, i), ix);
                        maxpy(n, -H.vo[i - 1], colg(&H, i), s, ix);
                    }
                    a = mdot(n, colg(&H, 1), colg(&H, 1), ix);
                    if (a > 0.) { const double sc = b / a; for (i = 0; i < n; ++i) s[i] = s[i] * sc; }   /* mxvscl */
                    for (i = k; i >= 1; --i) {                               /* mxdrcf, mssubs.c:412-441 */
                        const double t = *colu(&H, i) * mdot(n, s, colg(&H, i), ix);
                        maxpy(n, H.vo[i - 1] - t, colx(&H, i), s, ix);
                    }
                    snorm = sqrt(mdot(n, s, s, ix));
                    /* mxdrsu: every column becomes one older; the slot that falls off (or a fresh one) is the new column 1 */
                    H.head = (H.head + mf - 1) % mf;
                    if (LB_MIN(k + 1, mf) > 1) {
                        memcpy(colx(&H, 1), colx(&H, 2), sizeof(double) * (size_t) n);      /* the reference leaves the old column 1 in place */
                        memcpy(colg(&H, 1), colg(&H, 2), sizeof(double) * (size_t) n);
                        *colu(&H, 1) = *colu(&H, 2);
                    }
                }
            }
        }
        if (c.irest != 0) {                                                  /* steepest descent direction (L12620) */
            mneg(n, gf, s, ix);
            snorm = gnorm;
            if (c.kit < c.nit) { c.kit = c.nit; }
            else { c.iterm = -10; if (c.iters < 0) c.iterm = c.iters - 5; }
        }
        if (kd > 0) p = mdot(n, gf, s, ix);
        if (snorm <= 0.) c.irest = LB_MAX(c.irest, 1);
        else if (p + told * gnorm * snorm <= 0.) c.irest = 0;
        else c.irest = LB_MAX(c.irest, 1);
        if (c.irest == 0) {
            nred = 0;
            rmin = alf1 * gnorm / snorm;
            rmax = LB_MIN(alf2 * gnorm / snorm, xmax / snorm);
        }
        if (c.iterm != 0) break;
        if (c.irest != 0) goto direction;
        /* pytrcs (pssubs.c:1216-1271): save x, g in column 1; limit the step by the bounds */
        q.fp = fo; fo = fval; po = p;
        memcpy(colx(&H, 1), x, sizeof(double) * (size_t) n);
        memcpy(colg(&H, 1), gf, sizeof(double) * (size_t) n);
        for (i = 0; i < n; ++i) {
            if (ix[i] < 0) s[i] = 0.;
            else {
                if ((ix[i] == 1 || ix[i] >= 3) && s[i] < -1. / eta9) rmax = LB_MIN(rmax, (xl[i] - x[i]) / s[i]);
                if ((ix[i] == 2 || ix[i] >= 3) && s[i] > 1. / eta9) rmax = LB_MIN(rmax, (xu[i] - x[i]) / s[i]);
            }
        }
        if (rmax != 0.) {
            /* line search (L11170) */
            q.f = fval; q.fo = fo; q.p = p; q.po = po; q.minf_est = minf_est; q.maxf = maxf; q.rmin = rmin; q.rmax = rmax;
            q.tols = 1e-4; q.tolp = .8; q.kd = kd; q.ld = ld; q.nit = c.nit; q.kit = c.kit; q.nred = nred; q.mred = 10;
            q.maxst = maxst; q.iest = 0; q.inits = 2; q.iters = c.iters; q.kters = 3; q.mes = 4; q.isys = 0;
            for (;;) {
                lb_ps1l01(&q, &lss);
                if (q.isys == 0) break;
                {
                    const double *xs = colx(&H, 1);
                    for (i = 0; i < n; ++i) if (ix[i] >= 0) x[i] = xs[i] + q.r * s[i];          /* mxudir */
                }
                project(n, x, ix, xl, xu, eps9);
                q.f = f((unsigned) n, x, gf, f_data);
                ++stop->nevals;
                ++c.nfg;
                q.p = mdot(n, gf, s, ix);
            }
            fval = q.f; p = q.p; kd = q.kd; ld = q.ld; nred = q.nred; maxst = q.maxst; c.iters = q.iters;
            if (c.iters <= 0) {                                              /* L11174: zero step -> restore and restart */
                fval = fo; p = po;
                memcpy(x, colx(&H, 1), sizeof(double) * (size_t) n);
                memcpy(gf, colg(&H, 1), sizeof(double) * (size_t) n);
                c.irest = LB_MAX(c.irest, 1);
                ld = kd;
                goto direction;
            }
            /* pytrcd (pssubs.c:1065-1117): column 1 := differences, zeroed on active coordinates */
            {
                double *dx = colx(&H, 1), *dg = colg(&H, 1);
                for (i = 0; i < n; ++i) { dx[i] = x[i] - dx[i]; dg[i] = gf[i] - dg[i]; }
                po = q.r * po; p = q.r * p;
                for (i = 0; i < n; ++i) if (ix[i] < 0) { dx[i] = 0.; dg[i] = 0.; }
                xstop = orc_stop_dx(stop, x, dx);
            }
        }
        for (i = 0; i < n; ++i) if (ix[i] < 0) ix[i] = -ix[i];              /* mxvine */
        add_active(n, x, ix, xl, xu);
    }
    *minf = fval;
    ret = lb_result_of_iterm(c.iterm);
    free(work);
    free(ix);
    return ret;
}
