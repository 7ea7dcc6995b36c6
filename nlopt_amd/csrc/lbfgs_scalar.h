/* lbfgs_scalar.h — the scalar half of the bound-constrained limited-memory BFGS local optimiser
 * (NLOPT_LD_LBFGS = Luksan's PLIS, src/algs/luksan/plis.c), written once and compiled three ways,
 * like objfuncs.h: by gcc into the product's host code, by gcc into the CPU oracle
 * (oracle/port_lbfgs.c, pinned iterate-by-iterate against the real reference), and by hipcc into
 * the batched device kernel (hip/lbfgs_kernels.hip), where every thread of a workgroup runs it
 * redundantly on block-uniform values.
 *
 * What is here: the line search (PS1L01 with the PNINT1 inter/extrapolation, pssubs.c:92-204,
 * 283-482), the termination / restart test (PYFUT1, pssubs.c:886-969) and the iteration state of
 * plis_ (plis.c:106-417) as an explicit state machine.  What is NOT here: anything that touches a
 * vector (dot products, axpys, the Strang recurrences, bound handling) — those are the callers'
 * loops (sequential in the oracle, workgroup-parallel on the device).
 */
#ifndef NLA_LBFGS_SCALAR_H
#define NLA_LBFGS_SCALAR_H

#include <math.h>

#if defined(__HIPCC__)
#define LB_HD __host__ __device__ static inline
#else
#define LB_HD static inline
#endif

#define LB_MIN(a, b) ((a) < (b) ? (a) : (b))
#define LB_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ---- PNINT1 (pssubs.c:92-204): new trial step by extra/interpolation; returns merr --------------- */
LB_HD int lb_pnint1(double rl, double ru, double fl, double fu, double pl, double pu, double *r, int mode, int mtyp)
{
    double a = 0, b = 0, c, d, den = 0, dis, t;
    int ntyp;
    if (mode <= 0) return 0;
    if (pl >= 0.) return 2;
    if (ru <= rl) return 3;
    for (ntyp = mtyp; ntyp >= 1; --ntyp) {
        if (ntyp == 1) {                                   /* bisection */
            *r = (mode == 1) ? ru * 4. : (rl + ru) * .5;
            return 0;
        } else if (ntyp == mtyp) {
            a = (fu - fl) / (pl * (ru - rl));
            b = pu / pl;
        }
        if (ntyp == 2) den = (1. - a) * 2.;                /* quadratic, one derivative */
        else if (ntyp == 3) den = 1. - b;                  /* quadratic, two derivatives */
        else if (ntyp == 4) {                              /* cubic */
            c = b - a * 2. + 1.;
            d = b - a * 3. + 2.;
            dis = d * d - c * 3.;
            if (dis < 0.) continue;
            den = d + sqrt(dis);
        } else if (ntyp == 5) {                            /* conic */
            dis = a * a - b;
            if (dis < 0.) continue;
            den = a + sqrt(dis);
            if (den <= 0.) continue;
            t = 1. / den;
            den = 1. - b * (t * (t * t));
        }
        if (mode == 1 && den > 0. && den < 1.) {           /* extrapolation accepted */
            *r = rl + (ru - rl) / den;
            *r = LB_MAX(*r, ru * 1.1);
            *r = LB_MIN(*r, ru * 1e3);
            return 0;
        } else if (mode == 2 && den > 1.) {                /* interpolation accepted */
            *r = rl + (ru - rl) / den;
            if (rl == 0.) *r = LB_MAX(*r, rl + (ru - rl) * .01);
            else          *r = LB_MAX(*r, rl + (ru - rl) * .1);
            *r = LB_MIN(*r, rl + (ru - rl) * .9);
            return 0;
        }
    }
    return 0;
}

/* ---- PS1L01 (pssubs.c:283-482): reverse-communication line search ------------------------------- */
typedef struct {
    double fl, fu, pl, rl, pu, ru;
    int mes1, mes2, mes3, mode, mtyp;
} lb_ls_state;

typedef struct {
    double r, rp, f, fo, fp, p, po, pp, minf_est, maxf, rmin, rmax, tols, tolp, par1, par2;
    int kd, ld, nit, kit, nred, mred, maxst, iest, inits, iters, kters, mes, isys;
} lb_ls_io;

/* one entry of the line search: isys == 0 on entry starts a search, isys == 1 continues it after
 * the caller evaluated f, p at x + r*s.  On return isys == 1 asks for such an evaluation, isys == 0
 * means finished (io->iters tells how). */
LB_HD void lb_ps1l01(lb_ls_io *q, lb_ls_state *st)
{
    double fl = st->fl, fu = st->fu, pl = st->pl, rl = st->rl, pu = st->pu, ru = st->ru;
    int mes1 = st->mes1, mes2 = st->mes2, mes3 = st->mes3, mode = st->mode, mtyp = st->mtyp;
    int merr, init1, l1, l2, l3, l5, l7, m1, m2, m3;
    double rtemp;
    if (q->isys != 1) {
        mes1 = 2; mes2 = 2; mes3 = 2;
        q->iters = 0;
        if (q->po >= 0.) { q->r = 0.; q->iters = -2; goto finish; }
        if (q->rmax <= 0.) { q->iters = 0; goto finish; }
        /* initial stepsize */
        if (q->inits > 0) rtemp = q->minf_est - q->f;
        else if (q->iest == 0) rtemp = q->f - q->fp;
        else rtemp = LB_MAX(q->f - q->fp, q->minf_est - q->f);
        init1 = q->inits < 0 ? -q->inits : q->inits;
        q->rp = 0.;
        q->fp = q->fo;
        q->pp = q->po;
        if (init1 == 0) { }
        else if (init1 == 1 || (q->inits >= 1 && q->iest == 0)) q->r = 1.;
        else if (init1 == 2) q->r = LB_MIN(1., rtemp * 4. / q->po);
        else if (init1 == 3) q->r = LB_MIN(1., rtemp * 2. / q->po);
        else if (init1 == 4) q->r = rtemp * 2. / q->po;
        q->r = LB_MAX(q->r, q->rmin);
        q->r = LB_MIN(q->r, q->rmax);
        mode = 0;
        ru = 0.;
        fu = q->fo;
        pu = q->po;
        goto newstep;
    }
    /* continuation: f, p at the trial step are in q */
    if (mode == 0) { q->par1 = q->p / q->po; q->par2 = q->f - q->fo; }
    if (q->iters != 0) goto finish;
    if (q->f <= q->minf_est) { q->iters = 7; goto finish; }
    l1 = q->r <= q->rmin && q->nit != q->kit;
    l2 = q->r >= q->rmax;
    l3 = q->f - q->fo <= q->tols * q->r * q->po;
    l5 = q->p >= q->tolp * q->po || (mes2 == 2 && mode == 2);
    l7 = mes2 <= 2 || mode != 0;
    m1 = 0; m2 = 0;
    m3 = l3;
    if (mes3 >= 1) {
        m1 = fabs(q->p) <= fabs(q->po) * .01 && q->fo - q->f >= fabs(q->fo) * 9.9999999999999994e-12;
        l3 = l3 || m1;
    }
    if (mes3 >= 2) {
        m2 = fabs(q->p) <= fabs(q->po) * .5 && fabs(q->fo - q->f) <= fabs(q->fo) * 2.0000000000000001e-13;
        l3 = l3 || m2;
    }
    q->maxst = l2 ? 1 : 0;
    /* termination tests */
    if (l1 && !l3) { q->iters = 0; goto finish; }
    else if (l2 && l3 && !l5) { q->iters = 7; goto finish; }
    else if (m3 && mes1 == 3) { q->iters = 5; goto finish; }
    else if (l3 && l5 && l7) { q->iters = 4; goto finish; }
    else if (q->kters < 0 || (q->kters == 6 && l7)) { q->iters = 6; goto finish; }
    else if ((q->nred < 0 ? -q->nred : q->nred) >= q->mred) { q->iters = -1; goto finish; }
    else {
        q->rp = q->r; q->fp = q->f; q->pp = q->p;
        mode = LB_MAX(mode, 1);
        mtyp = q->mes < 0 ? -q->mes : q->mes;
        if (q->f >= q->maxf) mtyp = 1;
    }
    if (mode == 1) {                                       /* interval change after extrapolation */
        rl = ru; fl = fu; pl = pu;
        ru = q->r; fu = q->f; pu = q->p;
        if (!l3) { q->nred = 0; mode = 2; }
        else if (mes1 == 1) mtyp = 1;
    } else {                                               /* ... after interpolation */
        if (!l3) { ru = q->r; fu = q->f; pu = q->p; }
        else     { rl = q->r; fl = q->f; pl = q->p; }
    }
newstep:
    merr = lb_pnint1(rl, ru, fl, fu, pl, pu, &q->r, mode, mtyp);
    if (merr > 0) { q->iters = -merr; goto finish; }
    else if (mode == 1) { --q->nred; q->r = LB_MIN(q->r, q->rmax); }
    else if (mode == 2) ++q->nred;
    q->kd = 1; q->ld = -1; q->isys = 1;
    goto save;
finish:
    q->isys = 0;
save:
    st->fl = fl; st->fu = fu; st->pl = pl; st->rl = rl; st->pu = pu; st->ru = ru;
    st->mes1 = mes1; st->mes2 = mes2; st->mes3 = mes3; st->mode = mode; st->mtyp = mtyp;
}

/* ---- stopping tests used inside (stop.c:81-96,136-139) ------------------------------------------ */
typedef struct {
    double minf_max, ftol_rel, ftol_abs;
    int maxeval;               /* <= 0: none */
} lb_stop;

LB_HD int lb_isinf(double x) { return fabs(x) >= HUGE_VAL * 0.99; }
LB_HD int lb_stop_ftol(const lb_stop *s, double f, double oldf)
{
    double d;
    if (lb_isinf(oldf)) return 0;
    d = fabs(f - oldf);
    return d < s->ftol_abs || d < s->ftol_rel * (fabs(f) + fabs(oldf)) * 0.5 || (s->ftol_rel > 0 && f == oldf);
}

/* ---- PYFUT1 (pssubs.c:886-969): termination and restart test ------------------------------------- */
typedef struct {
    int nit, kit, mit, nfg, mfg, ntesx, mtesx, ntesf, mtesf, ites, ires1, ires2, irest, iters, iterm, kd;
} lb_counters;

LB_HD void lb_pyfut1(int n, double f, double *fo, double umax, double gmax, int xstop, const lb_stop *stop, int forced,
                     int nevals, double tolg, lb_counters *c)
{
    if (c->iterm < 0) return;
    if (c->ites > 0 && c->iters != 0) {
        if (c->nit <= 0) *fo = f + LB_MIN(sqrt(fabs(f)), fabs(f) / 10.);
        if (forced) { c->iterm = -999; return; }
        if (f <= stop->minf_max) { c->iterm = 3; return; }
        if (c->kd > 0 && gmax <= tolg && umax <= tolg) { c->iterm = 4; return; }
        if (c->nit <= 0) { c->ntesx = 0; c->ntesf = 0; }
        if (xstop) {
            c->iterm = 1;
            if (++c->ntesx >= c->mtesx) return;
        } else c->ntesx = 0;
        if (lb_stop_ftol(stop, f, *fo)) {
            c->iterm = 2;
            if (++c->ntesf >= c->mtesf) return;
        } else c->ntesf = 0;
    }
    if (c->nit >= c->mit) { c->iterm = 11; return; }
    if (stop->maxeval > 0 && nevals >= stop->maxeval) { c->iterm = 12; return; }
    if (c->nfg >= c->mfg) { c->iterm = 13; return; }
    c->iterm = 0;
    if (n > 0 && c->nit - c->kit >= c->ires1 * n + c->ires2) c->irest = LB_MAX(c->irest, 1);
    ++c->nit;
}

/* nlopt_result of a finished run (plis.c:499-509) */
LB_HD int lb_result_of_iterm(int iterm)
{
    switch (iterm) {
    case 1: return 4;      /* NLOPT_XTOL_REACHED */
    case 2: return 3;      /* NLOPT_FTOL_REACHED */
    case 3: return 2;      /* NLOPT_MINF_MAX_REACHED */
    case 4: return 1;      /* NLOPT_SUCCESS: gradient tolerance */
    case 6: return 1;
    case 12: case 13: return 5;   /* NLOPT_MAXEVAL_REACHED */
    case 100: return 6;    /* NLOPT_MAXTIME_REACHED */
    case -999: return -5;  /* NLOPT_FORCED_STOP */
    default: return -1;    /* NLOPT_FAILURE */
    }
}

#endif /* NLA_LBFGS_SCALAR_H */
