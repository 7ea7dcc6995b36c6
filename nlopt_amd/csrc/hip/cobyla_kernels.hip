/* cobyla_kernels.hip — batched NLOPT_LN_COBYLA on gfx950: the local searches of NLOPT_GN_MLSL / GN_MLSL_LDS (whose default local
 * optimiser it is, src/api/optimize.c:763-768) with a compiled-in device objective, ONE WAVEFRONT per start point, the whole search
 * on the device and its whole state in LDS (rounds 2-5 ran these searches one after another on the host).
 *
 * What one search is (src/algs/cobyla/cobyla.c:181-271 around :452-1872, as nlopt_optimize reaches it through optimize.c:836-851):
 *   set-up     the default initial step from the start point and the box (options.c:921-946) unless the caller gave one; coordinates
 *              rescaled by the steps (rescale.c:30-48); the box as 2n linear constraint rows AND enforced on every point (cobyla.c:79-124);
 *              rhobeg = |dx_0 / scale_0|, rhoend from xtol_rel / xtol_abs
 *   iteration  Powell's COBYLA: linear models of f and of every row on a simplex of n+1 points (cobyla.c:688-811), a trust-region LP
 *              with an active set kept by Givens rotations (TRSTLP, cobyla.c:1247-1872), a merit function with an adaptive penalty
 *   result     the best point any evaluation saw (the dispatcher's memoize wrapper for unconstrained COBYLA, optimize.c:450-508,1026-1071)
 *
 * How the wavefront runs it.  The algorithm is a chain of small dense operations (n <= 51 here) whose every sum the reference forms
 * in one accumulator over ascending indices.  To stay the reference's run evaluation by evaluation that ORDER is kept; what is spread
 * over the 64 lanes is the set of independent sums and the element-wise updates:
 *   "map"      v[i] = ... for all i: lane-strided                                  (vertex / direction / multiplier updates)
 *   "many"     a SET of independent sums (the n^2 entries of SIMI x SIM - 1, the (m+1) n model gradients, the residuals of the inactive
 *              rows, the n rows of a rank-one update ...): one sum per lane, each lane serial over its own
 *   "one"      a single sum that decides the next step (a Givens angle, a multiplier): every lane forms it redundantly out of LDS
 *              (broadcast reads), so the value is in every lane's registers without an exchange
 *   "rows"     a CHAIN of Givens rotations over neighbouring columns of Z (adding a row to the active set, cobyla.c:1402-1448): the
 *              angles depend only on sums over columns no earlier rotation of the chain touches, so all angles come first ("many" +
 *              a scalar recurrence) and then every lane carries ONE ROW of Z through the whole chain
 * Scalars (rho, parmu, the LP's counters ...) live in every lane's registers, identical by construction; control flow is uniform.
 * Matrices are column-major with an ODD leading dimension: both the walk down a column and the walk along a row are free of LDS bank
 * conflicts.  Between two phases that touch the same array from different lanes stands a workgroup barrier (one wavefront: cheap).
 *
 * The same algorithm as ONE thread's state machine is ../cobyla_core.h (the host's LN_COBYLA, cobyla_host.c); the two are compared
 * evaluation by evaluation on the CPU (tools/cobyla_emu_check.py: this file compiled by g++ over tools/simt_emu, 64 lockstep threads)
 * and on the device against the real reference (tests/test_gpu_cobyla.py).
 *
 * Bound by: LDS latency x dependent fp64 adds of the serial sums; HBM and MFMA play no part (per evaluation: n doubles of the start /
 * result row).  Occupancy: LDS per search ~ (4n + m + 2)(n|1) + (n+1)((m+2)|1) + 14n + 8m doubles (n = 16: 22 KB, 7 searches per
 * compute unit; n = 40: 102 KB, one). */
#define LB_T 64                      /* one wavefront per search: local_common.h's workgroup helpers become wavefront helpers */
#define LB_XCH 64
#include "local_common.h"
#include "../../../include/nlopt_amd.h"
#include <float.h>

#define CW_SYNC() __syncthreads()
#define CW_LANES 64

/* result codes = nlopt_result values (nlopt.h:167-181) */
#define CW_SUCCESS 1
#define CW_MINF_MAX_REACHED 2
#define CW_FTOL_REACHED 3
#define CW_XTOL_REACHED 4
#define CW_MAXEVAL_REACHED 5
#define CW_MAXTIME_REACHED 6
#define CW_FORCED_STOP (-5)
#define CW_ROUNDOFF_LIMITED (-4)
#define CW_INVALID_ARGS (-2)

__device__ static inline int cw_isinf(double x) { return fabs(x) >= __builtin_huge_val() * 0.99 || isinf(x); }      /* nlopt_isinf, stop.c:219-227 */
__device__ static inline int cw_istiny(double x) { return x == 0.0 || fabs(x) < DBL_MIN; }                          /* nlopt_istiny, stop.c:240-254 */

__device__ static inline int cw_tol_reached(double vold, double vnew, double reltol, double abstol)       /* relstop, stop.c:81-86 */
{
    if (cw_isinf(vold)) return 0;
    const double d = fabs(vnew - vold);
    return d < abstol || d < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}
/* the reference's deterministic LCG for the simplex-repair steps (cobyla.c:300-309) */
__device__ static inline double cw_lcg_between(uint32_t *seed, double a, double b)
{
    *seed = *seed * 1103515245u + 12345u;
    return a + *seed * (b - a) / ((uint32_t) -1);
}
/* "is this scalar product more than its own rounding noise?" (the acca / accb device, e.g. cobyla.c:1422-1426), and the form the
 * reference uses where a sum is to be ZEROED (not the negation of the first when a NaN is involved) */
__device__ static inline int cw_significant(double sum, double sumabs, double c1, double c2)
{
    const double acca = sumabs + fabs(sum) * c1, accb = sumabs + fabs(sum) * c2;
    return sumabs < acca && acca < accb;
}
__device__ static inline int cw_noise(double sum, double sumabs, double c1, double c2)
{
    const double acca = sumabs + fabs(sum) * c1, accb = sumabs + fabs(sum) * c2;
    return sumabs >= acca || acca >= accb;
}

/* the search's arrays in LDS (pointers and sizes are the same in every lane) */
struct cw_ws {
    int n, m, ldn, ldd;
    double *sim, *simi, *dat, *a, *z;                                         /* matrices */
    double *vsig, *veta, *sigbar, *dx, *zdota, *sdirn, *dxnew, *x;            /* n each */
    double *scale, *slb, *sub, *xev, *bestx, *step0;                          /* n each */
    double *con, *vmultc, *vmultd, *s1, *s2, *s3;                             /* max(m + 2, n + 1) each */
    int *iact, *rot;                                                          /* m + 2; n + 1: which steps of a rotation chain rotate */
};
__host__ __device__ static inline int cw_odd(int v) { return v | 1; }
__host__ __device__ static inline int cw_vlen(int n, int m) { return (m + 2 > n + 1 ? m + 2 : n + 1); }
/* doubles of LDS for n variables and m constraint rows (ints counted as halves) */
__host__ __device__ static inline size_t cw_lds_doubles(int n, int m)
{
    const size_t ldn = (size_t) cw_odd(n), ldd = (size_t) cw_odd(m + 2), v = (size_t) cw_vlen(n, m);
    return ldn * (size_t) (n + 1) + ldn * (size_t) n + ldd * (size_t) (n + 1) + ldn * (size_t) (m + 1) + ldn * (size_t) n
           + 14 * (size_t) n + 6 * v + ((size_t) (m + 2) + (size_t) n + 1) / 2 + 1;
}

#define SIM(i, j)  W.sim[(j) * W.ldn + (i)]            /* j < n: displacement of vertex j from the pole; j == n: the pole                cobyla.c:493-497 */
#define SIMI(j, i) W.simi[(i) * W.ldn + (j)]           /* inverse of the displacement matrix */
#define DAT(k, j)  W.dat[(j) * W.ldd + (k)]            /* values at vertex j: k < m rows, k == m the objective, k == m+1 the greatest violation */
#define ACOL(i, k) W.a[(k) * W.ldn + (i)]              /* gradient of the linear model of row k; column m = MINUS the objective's gradient */
#define ZC(k) (W.z + (k) * W.ldn)
#define AC(k) (W.a + (k) * W.ldn)

/* ---- the trust-region LP (Powell's TRSTLP, cobyla.c:1247-1872): stage one finds the shortest dx, |dx| <= rho, that minimises the greatest
 * violation of a_k . dx >= b_k (b = W.con); stage two uses what is left of the trust region to reduce the objective (-a_m . dx) without
 * increasing that violation.  Active set with an orthogonal basis Z kept by Givens rotations. ---- */

/* rotate columns k, k+1 of Z so that active row k+1 takes position k (cobyla.c:1524-1551 and :1628-1655): moves the row at position
 * `from` to the end of the active set.  Each step's angle needs a column the step before has just rotated: a serial chain. */
__device__ static void cw_move_to_end(const cw_ws &W, int from, int nact, int lane)
{
    const int n = W.n;
    const int isave = W.iact[from];
    const double vsave = W.vmultc[from];
    int k = from;
    while (k < nact - 1) {
        const int kp = k + 1, kw = W.iact[kp];
        double *zk = ZC(k), *zkp = ZC(kp);
        const double *akw = AC(kw);
        double sp = 0.;
        for (int i = 0; i < n; ++i) sp += zk[i] * akw[i];                                   /* "one" */
        const double zdkp = W.zdota[kp], zdk = W.zdota[k], vkp = W.vmultc[kp];
        const double temp = sqrt(sp * sp + zdkp * zdkp), alpha = zdkp / temp, beta = sp / temp;
        CW_SYNC();
        if (lane == 0) { W.zdota[kp] = alpha * zdk; W.zdota[k] = temp; W.iact[k] = kw; W.vmultc[k] = vkp; }
        for (int i = lane; i < n; i += CW_LANES) {
            const double t = alpha * zkp[i] + beta * zk[i];
            zkp[i] = alpha * zk[i] - beta * zkp[i];
            zk[i] = t;
        }
        CW_SYNC();
        k = kp;
    }
    CW_SYNC();
    if (lane == 0) { W.iact[k] = isave; W.vmultc[k] = vsave; }
    CW_SYNC();
}

/* returns CW_SUCCESS or CW_ROUNDOFF_LIMITED; *ifull_out = 0 if dx could not reach the length rho */
__device__ static int cw_trust_lp(const cw_ws &W, double rho, int *ifull_out, int lane)
{
    const int n = W.n, m = W.m;
    const double tiny = (double) 1e-6f, c1f = (double) .1f, c2f = (double) .2f;   /* the reference writes these three as float literals */
    double resmax = 0., resold = 0., optold = 0., optnew, stpful, step, ratio, temp, tot;
    int icon = -1, icount = 0, nactx = 0, nact = 0, mcon = m, i, k, kk;
    enum { RESET_COUNT, ITERATE, STAGE_TWO, STUCK } phase;
    const double *b = W.con;

    *ifull_out = 1;
    for (int e = lane; e < n * n; e += CW_LANES) { const int kc = e / n, ir = e - kc * n; W.z[kc * W.ldn + ir] = kc == ir ? 1. : 0.; }
    for (i = lane; i < n; i += CW_LANES) W.dx[i] = 0.;
    for (k = 0; k < m; ++k) if (b[k] > resmax) { resmax = b[k]; icon = k; }      /* cobyla.c:1341-1354 */
    for (k = lane; k < m; k += CW_LANES) { W.iact[k] = k; W.vmultc[k] = resmax - b[k]; }
    if (resmax == 0.) phase = STAGE_TWO;
    else { for (i = lane; i < n; i += CW_LANES) W.sdirn[i] = 0.; phase = RESET_COUNT; }
    CW_SYNC();

    for (;;) {
        if (phase == STUCK) {                                     /* L490 */
            if (mcon == m) phase = STAGE_TWO;
            else { *ifull_out = 0; return CW_SUCCESS; }
        }
        if (phase == STAGE_TWO) {                                 /* L480 */
            mcon = m + 1;
            icon = m;
            CW_SYNC();
            if (lane == 0) { W.iact[m] = m; W.vmultc[m] = 0.; }
            CW_SYNC();
            phase = RESET_COUNT;
        }
        if (phase == RESET_COUNT) { optold = 0.; icount = 0; phase = ITERATE; }       /* L60 */

        /* ---- L70: cycling guard (cobyla.c:1363-1394) ---- */
        if (mcon == m) optnew = resmax;
        else { const double *am = AC(m); optnew = 0.; for (i = 0; i < n; ++i) optnew -= W.dx[i] * am[i]; }
        if (icount == 0 || optnew < optold) { optold = optnew; nactx = nact; icount = 3; }
        else if (nact > nactx) { nactx = nact; icount = 3; }
        else if (--icount == 0) { phase = STUCK; continue; }

        if (icon >= nact) {
            /* ---- add row iact[icon] to the active set (cobyla.c:1396-1457) ---- */
            kk = W.iact[icon];
            CW_SYNC();
            for (i = lane; i < n; i += CW_LANES) W.dxnew[i] = AC(kk)[i];
            CW_SYNC();
            /* the projections of the new gradient on the free columns of Z — no rotation of this chain touches column k before step k
             * reads it: "many" */
            for (k = nact + lane; k < n; k += CW_LANES) {
                const double *zk = ZC(k);
                double sp = 0., spabs = 0.;
                for (i = 0; i < n; ++i) { temp = zk[i] * W.dxnew[i]; sp += temp; spabs += fabs(temp); }
                if (cw_noise(sp, spabs, .1, .2)) sp = 0.;
                W.s1[k] = sp;
            }
            CW_SYNC();
            /* the angles: a scalar recurrence (s2 = alpha, s3 = beta; beta stays NaN-free "no rotation" where the reference only sets tot) */
            tot = 0.;
            for (k = n - 1; k >= nact; --k) {
                const double sp = W.s1[k];
                double al = 0., be = 0.;
                int rot = 0;
                if (tot == 0.) tot = sp;
                else { temp = sqrt(sp * sp + tot * tot); al = sp / temp; be = tot / temp; tot = temp; rot = 1; }
                if (lane == (k & (CW_LANES - 1))) { W.s2[k] = al; W.s3[k] = be; W.rot[k] = rot; }
            }
            CW_SYNC();
            /* every lane carries one ROW of Z through the chain: "rows" */
            for (i = lane; i < n; i += CW_LANES)
                for (k = n - 1; k >= nact; --k)
                    if (W.rot[k]) {
                        const double al = W.s2[k], be = W.s3[k], zk = W.z[k * W.ldn + i], zkp = W.z[(k + 1) * W.ldn + i];
                        W.z[(k + 1) * W.ldn + i] = al * zkp - be * zk;
                        W.z[k * W.ldn + i] = al * zk + be * zkp;
                    }
            CW_SYNC();
            if (tot != 0.) {                                      /* room in the active set */
                const double vn = W.vmultc[nact];
                CW_SYNC();
                if (lane == 0) { W.zdota[nact] = tot; W.vmultc[icon] = vn; W.vmultc[nact] = 0.; }
                ++nact;
                CW_SYNC();
            } else {
                /* the new gradient is a combination of the active ones: one of them has to leave (cobyla.c:1459-1565) */
                ratio = -1.;
                for (k = nact - 1; k >= 0; --k) {
                    double zdotv = 0., zdvabs = 0.;
                    const double *zk = ZC(k);
                    for (i = 0; i < n; ++i) { temp = zk[i] * W.dxnew[i]; zdotv += temp; zdvabs += fabs(temp); }
                    if (cw_significant(zdotv, zdvabs, .1, .2)) {
                        temp = zdotv / W.zdota[k];
                        if (temp > 0. && W.iact[k] < m) {
                            const double tempa = W.vmultc[k] / temp;
                            if (ratio < 0. || tempa < ratio) ratio = tempa;
                        }
                        if (k >= 1) {
                            const double *akw = AC(W.iact[k]);
                            CW_SYNC();
                            for (i = lane; i < n; i += CW_LANES) W.dxnew[i] -= temp * akw[i];
                            CW_SYNC();
                        }
                        if (lane == 0) W.vmultd[k] = temp;
                    } else if (lane == 0) W.vmultd[k] = 0.;
                }
                CW_SYNC();
                if (ratio < 0.) { phase = STUCK; continue; }
                for (k = lane; k < nact; k += CW_LANES) { temp = W.vmultc[k] - ratio * W.vmultd[k]; W.vmultc[k] = 0. >= temp ? 0. : temp; }
                CW_SYNC();
                if (icon < nact - 1) cw_move_to_end(W, icon, nact, lane);
                temp = 0.;
                { const double *zl = ZC(nact - 1), *akk = AC(kk); for (i = 0; i < n; ++i) temp += zl[i] * akk[i]; }
                if (temp == 0.) { phase = STUCK; continue; }
                CW_SYNC();
                if (lane == 0) { W.zdota[nact - 1] = temp; W.vmultc[icon] = 0.; W.vmultc[nact - 1] = ratio; }
                CW_SYNC();
            }
            /* L210: bookkeeping; in stage two the objective stays the LAST active row (cobyla.c:1567-1599) */
            {
                const int last = nact - 1;
                const int il = W.iact[last];
                CW_SYNC();
                if (lane == 0) { W.iact[icon] = il; W.iact[last] = kk; }
                CW_SYNC();
                if (mcon > m && kk != m) {
                    double sp = 0.;
                    double *zk = ZC(last - 1), *zl = ZC(last);
                    const double *akk = AC(kk);
                    k = last - 1;
                    for (i = 0; i < n; ++i) sp += zk[i] * akk[i];
                    const double zdl = W.zdota[last], zdk = W.zdota[k], vk = W.vmultc[k], vl = W.vmultc[last];
                    const int ik = W.iact[k];
                    temp = sqrt(sp * sp + zdl * zdl);
                    const double alpha = zdl / temp, beta = sp / temp;
                    CW_SYNC();
                    if (lane == 0) { W.zdota[last] = alpha * zdk; W.zdota[k] = temp; W.iact[last] = ik; W.iact[k] = kk; W.vmultc[k] = vl; W.vmultc[last] = vk; }
                    for (i = lane; i < n; i += CW_LANES) {
                        const double t = alpha * zl[i] + beta * zk[i];
                        zl[i] = alpha * zk[i] - beta * zl[i];
                        zk[i] = t;
                    }
                    CW_SYNC();
                }
                if (mcon == m) {                                  /* stage one: next search direction (cobyla.c:1607-1618) */
                    const double *zl = ZC(last), *ak = AC(W.iact[last]);
                    temp = 0.;
                    for (i = 0; i < n; ++i) temp += W.sdirn[i] * ak[i];
                    temp += -1.;
                    temp /= W.zdota[last];
                    CW_SYNC();
                    for (i = lane; i < n; i += CW_LANES) W.sdirn[i] -= temp * zl[i];
                    CW_SYNC();
                }
            }
        } else {
            /* ---- L260: delete row iact[icon] from the active set (cobyla.c:1621-1676) ---- */
            if (icon < nact - 1) cw_move_to_end(W, icon, nact, lane);
            --nact;
            if (mcon == m) {
                const double *zd = ZC(nact);
                temp = 0.;
                for (i = 0; i < n; ++i) temp += W.sdirn[i] * zd[i];
                CW_SYNC();
                for (i = lane; i < n; i += CW_LANES) W.sdirn[i] -= temp * zd[i];
                CW_SYNC();
            }
        }
        if (mcon > m) {                                           /* L320: search direction of stage two */
            const double *zl = ZC(nact - 1);
            temp = 1. / W.zdota[nact - 1];
            CW_SYNC();
            for (i = lane; i < n; i += CW_LANES) W.sdirn[i] = temp * zl[i];
            CW_SYNC();
        }

        /* ---- L340: step to the trust-region boundary, or the step that takes resmax to zero (cobyla.c:1687-1726) ---- */
        {
            double dd = rho * rho, sd = 0., ss = 0.;
            for (i = 0; i < n; ++i) {
                const double dxi = W.dx[i], si = W.sdirn[i];
                if (fabs(dxi) >= rho * tiny) dd -= dxi * dxi;
                sd += dxi * si;
                ss += si * si;
            }
            if (dd <= 0.) { phase = STUCK; continue; }
            temp = sqrt(ss * dd);
            if (fabs(sd) >= temp * tiny) temp = sqrt(ss * dd + sd * sd);
            stpful = dd / (temp + sd);
            step = stpful;
            if (mcon == m) {
                const double acca = step + resmax * .1, accb = step + resmax * .2;
                if (step >= acca || acca >= accb) { phase = STAGE_TWO; continue; }
                step = step <= resmax ? step : resmax;
            }
            if (cw_isinf(step)) return CW_ROUNDOFF_LIMITED;
        }
        CW_SYNC();
        for (i = lane; i < n; i += CW_LANES) W.dxnew[i] = W.dx[i] + step * W.sdirn[i];
        CW_SYNC();
        if (mcon == m) {                                          /* cobyla.c:1737-1750 */
            resold = resmax;
            resmax = 0.;
            for (k = lane; k < nact; k += CW_LANES) {             /* "many" */
                const int id = W.iact[k];
                const double *ak = AC(id);
                double t = b[id];
                for (i = 0; i < n; ++i) t -= ak[i] * W.dxnew[i];
                W.s1[k] = t;
            }
            CW_SYNC();
            for (k = 0; k < nact; ++k) { temp = W.s1[k]; resmax = resmax >= temp ? resmax : temp; }
        }
        /* multipliers the active rows would have at dxnew (cobyla.c:1752-1785): dxnew changes between two of them, a serial chain */
        for (k = nact - 1; k >= 0; --k) {
            double zdotw = 0., zdwabs = 0.;
            const double *zk = ZC(k);
            for (i = 0; i < n; ++i) { temp = zk[i] * W.dxnew[i]; zdotw += temp; zdwabs += fabs(temp); }
            if (cw_noise(zdotw, zdwabs, .1, .2)) zdotw = 0.;
            const double vm = zdotw / W.zdota[k];
            if (lane == 0) W.vmultd[k] = vm;
            if (k >= 1) {
                const double *ak = AC(W.iact[k]);
                CW_SYNC();
                for (i = lane; i < n; i += CW_LANES) W.dxnew[i] -= vm * ak[i];
                CW_SYNC();
            }
        }
        CW_SYNC();
        if (mcon > m && nact >= 1) { if (lane == 0) { temp = W.vmultd[nact - 1]; W.vmultd[nact - 1] = 0. >= temp ? 0. : temp; } }
        /* residuals of the inactive rows at dxnew (cobyla.c:1787-1813): "many" */
        for (i = lane; i < n; i += CW_LANES) W.dxnew[i] = W.dx[i] + step * W.sdirn[i];
        CW_SYNC();
        for (k = nact + lane; k < mcon; k += CW_LANES) {
            const int id = W.iact[k];
            const double *ak = AC(id);
            double sum = resmax - b[id], sumabs = resmax + fabs(b[id]);
            for (i = 0; i < n; ++i) { temp = ak[i] * W.dxnew[i]; sum += temp; sumabs += fabs(temp); }
            if (cw_noise(sum, sumabs, c1f, c2f)) sum = 0.;
            W.vmultd[k] = sum;
        }
        CW_SYNC();
        /* how much of the step can be taken (cobyla.c:1815-1844) */
        ratio = 1.;
        icon = -1;
        for (k = 0; k < mcon; ++k) {
            const double vd = W.vmultd[k];
            if (vd < 0.) {
                const double vc = W.vmultc[k];
                temp = vc / (vc - vd);
                if (temp < ratio) { ratio = temp; icon = k; }
            }
        }
        temp = 1. - ratio;
        CW_SYNC();
        for (i = lane; i < n; i += CW_LANES) W.dx[i] = temp * W.dx[i] + ratio * W.dxnew[i];
        for (k = lane; k < mcon; k += CW_LANES) { const double v = temp * W.vmultc[k] + ratio * W.vmultd[k]; W.vmultc[k] = 0. >= v ? 0. : v; }
        CW_SYNC();
        if (mcon == m) resmax = resold + ratio * (resmax - resold);
        if (icon >= 0) { phase = ITERATE; continue; }
        if (step == stpful) return CW_SUCCESS;                  /* L500 */
        phase = STAGE_TWO;
    }
}

/* replace vertex jdrop's displacement by dx and update the inverse (cobyla.c:869-897 / :1079-1103) */
__device__ static void cw_replace_vertex(const cw_ws &W, int jdrop, int after_repair, int lane)
{
    const int n = W.n;
    double temp = 0.;
    int i, j;
    CW_SYNC();
    if (!after_repair) for (i = lane; i < n; i += CW_LANES) SIM(i, jdrop) = W.dx[i];          /* (the repair step stored SIM itself, inside its bound fix-up) */
    for (i = 0; i < n; ++i) temp += SIMI(jdrop, i) * W.dx[i];
    CW_SYNC();
    for (i = lane; i < n; i += CW_LANES) SIMI(jdrop, i) /= temp;
    CW_SYNC();
    for (j = lane; j < n; j += CW_LANES) {                                                    /* one row of the inverse per lane */
        if (j == jdrop) continue;
        double t = 0.;
        for (i = 0; i < n; ++i) t += SIMI(j, i) * W.dx[i];
        for (i = 0; i < n; ++i) SIMI(j, i) -= t * SIMI(jdrop, i);
    }
    CW_SYNC();
}

#ifdef NLA_SIMT_EMU
static double cw_lds[24576];                                                                  /* (the CPU emulation runs one workgroup at a time) */
#endif

template <int OBJ>
__global__ __launch_bounds__(CW_LANES) void cobyla_batch_kernel(int n, int ld, int count, const double *__restrict__ lb, const double *__restrict__ ub,
                                                                const double *__restrict__ dx_given, double *__restrict__ X,
                                                                nla_cobyla_params P, nla_lbfgs_result *__restrict__ out)
{
#ifndef NLA_SIMT_EMU
    extern __shared__ double cw_lds[];
#endif
    __shared__ lb_shared S;
    __shared__ double oscratch[8];
    __shared__ lb_exact_buf XB;
    const int inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= count) return;
    double *x0 = X + (size_t) inst * ld;
    const double alpha = .25, beta = 2.1, gamma_ = .5, delta = 1.1;
    enum { EVAL_PRE, EVAL_POST, POLE, TRUST_STEP, JUDGE, SHRINK, FINISH_POLE, FINISH_HERE, FINISHED };
    int i, j, k, m = 0;

    /* the number of rows first: the finite bounds (cobyla.c:233-241) */
    for (j = 0; j < n; ++j) { if (!cw_isinf(lb[j])) ++m; if (!cw_isinf(ub[j])) ++m; }
    cw_ws W;
    {
        const int v = cw_vlen(n, m);
        double *p = cw_lds;
        W.n = n; W.m = m; W.ldn = cw_odd(n); W.ldd = cw_odd(m + 2);
        W.sim = p; p += W.ldn * (n + 1);
        W.simi = p; p += W.ldn * n;
        W.dat = p; p += W.ldd * (n + 1);
        W.a = p; p += W.ldn * (m + 1);
        W.z = p; p += W.ldn * n;
        W.vsig = p; p += n; W.veta = p; p += n; W.sigbar = p; p += n; W.dx = p; p += n; W.zdota = p; p += n; W.sdirn = p; p += n; W.dxnew = p; p += n;
        W.x = p; p += n; W.scale = p; p += n; W.slb = p; p += n; W.sub = p; p += n; W.xev = p; p += n; W.bestx = p; p += n; W.step0 = p; p += n;
        W.con = p; p += v; W.vmultc = p; p += v; W.vmultd = p; p += v; W.s1 = p; p += v; W.s2 = p; p += v; W.s3 = p; p += v;
        W.iact = (int *) p; W.rot = W.iact + (m + 2);
    }
    const int np = n, mp = m, mpp = m + 1;        /* the pole's column, the objective's row, the violation's row */
    const size_t total = cw_lds_doubles(n, m);
    for (size_t e = lane; e < total; e += CW_LANES) cw_lds[e] = 0.;
    CW_SYNC();

    /* ---- set-up ---- */
    int bad = 0;
    double rhobeg, rhoend;
    /* the initial step: the caller's, or nlopt_set_default_initial_step(opt, x) (options.c:921-946) — a quarter of the box, or 3/4 of the
     * gap to a bound that is nearer than that */
    for (j = lane; j < n; j += CW_LANES) {
        double step;
        if (dx_given) step = dx_given[j];
        else {
            const double lo = lb[j], hi = ub[j], xj = x0[j];
            step = __builtin_huge_val();
            if (!cw_isinf(hi) && !cw_isinf(lo) && (hi - lo) * 0.25 < step && hi > lo) step = (hi - lo) * 0.25;
            if (!cw_isinf(hi) && hi - xj < step && hi > xj) step = (hi - xj) * 0.75;
            if (!cw_isinf(lo) && xj - lo < step && xj > lo) step = (xj - lo) * 0.75;
            if (cw_isinf(step)) {
                if (!cw_isinf(hi) && fabs(hi - xj) < fabs(step)) step = (hi - xj) * 1.1;
                if (!cw_isinf(lo) && fabs(xj - lo) < fabs(step)) step = (xj - lo) * 1.1;
            }
            if (cw_isinf(step) || cw_istiny(step)) step = xj;
            if (cw_isinf(step) || step == 0.0) step = 1;
        }
        W.step0[j] = step;
    }
    CW_SYNC();
    /* nlopt_compute_rescaling (rescale.c:30-48), the scaled box and point (cobyla.c:200-232) */
    {
        int uniform = 1;
        for (j = 1; j < n; ++j) if (W.step0[j] != W.step0[j - 1]) { uniform = 0; break; }
        const double d0 = W.step0[0];
        rhobeg = fabs(d0 / 1.0);
        rhoend = P.xtol_rel * rhobeg;
        for (j = 0; j < n; ++j) {
            const double sc = (uniform || j == 0) ? 1.0 : W.step0[j] / d0;
            if (sc == 0 || !isfinite(sc)) bad = 1;
            if (P.xtol_abs && rhoend < P.xtol_abs[j] / fabs(sc)) rhoend = P.xtol_abs[j] / fabs(sc);
            if (lane == (j & (CW_LANES - 1))) {
                double l = lb[j] / sc, u = ub[j] / sc;
                if (l > u) { const double t = l; l = u; u = t; }
                W.scale[j] = sc; W.slb[j] = l; W.sub[j] = u; W.x[j] = x0[j] / sc;
            }
        }
    }
    CW_SYNC();

    /* the initial simplex: the pole at x, vertex i one step along coordinate i, the step kept inside the box (cobyla.c:538-562) */
    for (i = lane; i < n; i += CW_LANES) {
        double rhocur = rhobeg;
        const double xi = W.x[i];
        SIM(i, np) = xi;
        if (xi + rhocur > W.sub[i]) {
            if (xi - rhocur >= W.slb[i]) rhocur = -rhocur;
            else if (W.sub[i] - xi > xi - W.slb[i]) rhocur = 0.5 * (W.sub[i] - xi);
            else rhocur = 0.5 * (xi - W.slb[i]);
        }
        SIM(i, i) = rhocur;
        SIMI(i, i) = 1.0 / rhocur;
    }
    CW_SYNC();

    double rho = rhobeg, parmu = 0., parsig = 0., pareta, prerec = 0., prerem = 0., f = 0., resmax = 0., temp, tempa, sum = 0., minf = __builtin_huge_val();
    double bestf = DBL_MAX;
    int jdrop = np, ibrnch = 0, iflag = 0, ifull = 0, nbest, go = EVAL_PRE, rc = CW_SUCCESS, nevals = 0, forced = 0, timed = 0;
    uint32_t seed = (uint32_t) (n + m);
    if (bad) { rc = CW_INVALID_ARGS /* invalid scaling (cobyla.c:207-212) */; go = FINISHED; }

    while (go != FINISHED) switch (go) {
    case EVAL_PRE: {                                              /* L40 (cobyla.c:573-631): the stop tests in front of an evaluation */
        if (P.abort) { const int ab = lb_poll_abort(P.abort); forced = ab == -999; timed = ab == 100; }
        if (forced) rc = CW_FORCED_STOP;
        else if (nevals > 0) {
            if (P.maxeval > 0 && nevals >= P.maxeval) rc = CW_MAXEVAL_REACHED;
            else if (timed) rc = CW_MAXTIME_REACHED;
        }
        if (rc != CW_SUCCESS) { go = FINISH_POLE; break; }
        ++nevals;
        /* ---- the evaluation: the point clipped to the (scaled) box and unscaled (cobyla.c:90-99), f by the wavefront ---- */
        CW_SYNC();
        for (j = lane; j < n; j += CW_LANES) {
            const double xj = W.x[j], v = xj < W.slb[j] ? W.slb[j] : (xj > W.sub[j] ? W.sub[j] : xj);
            W.xev[j] = v * W.scale[j];
        }
        CW_SYNC();
        if (P.exact) { nla_obj_part t; f = lb_obj_exact<OBJ>(n, W.xev, &t, XB); }
        else f = nla_block_objective_as<OBJ, 1, 4>(n, [&](int q) { return W.xev[q]; }, oscratch);      /* the bits of the other local optimisers' 256-thread reduction */
        f *= P.sign;
        {   /* memoize_func (optimize.c:450-483): the best value seen at a point inside the caller's box */
            int outside = 0;
            for (j = lane; j < n; j += CW_LANES) if (W.xev[j] < lb[j] || W.xev[j] > ub[j]) outside = 1;
            outside = lb_block_isum(outside, S);
            if (!outside && f < bestf) { bestf = f; for (j = lane; j < n; j += CW_LANES) W.bestx[j] = W.xev[j]; }
        }
        /* the box as rows at the UNclipped scaled point (cobyla.c:112-121) */
        {
            int off = 0;
            for (j = 0; j < n; ++j) {
                const int fl = !cw_isinf(W.slb[j]), fu = !cw_isinf(W.sub[j]);
                if (lane == (j & (CW_LANES - 1))) {
                    if (fl) W.con[off] = W.x[j] - W.slb[j];
                    if (fu) W.con[off + fl] = W.sub[j] - W.x[j];
                }
                off += fl + fu;
            }
        }
        CW_SYNC();
        go = EVAL_POST;
        break;
    }
    case EVAL_POST: {
        int feasible = 1;
        resmax = 0.;
        for (k = 0; k < m; ++k) {
            const double v = -W.con[k];
            resmax = resmax >= v ? resmax : v;
            if (v > 0.) feasible = 0;                             /* (bound rows: tolerance zero) */
        }
        if (f < P.minf_max && feasible) { rc = CW_MINF_MAX_REACHED; go = FINISH_HERE; break; }
        CW_SYNC();
        if (lane == 0) { W.con[mp] = f; W.con[mpp] = resmax; }
        CW_SYNC();
        if (ibrnch == 1) { go = JUDGE; break; }
        /* a vertex of the simplex: its values go into its column (cobyla.c:633-684) */
        for (k = lane; k <= mpp; k += CW_LANES) DAT(k, jdrop) = W.con[k];
        CW_SYNC();
        if (nevals <= n + 1) {
            if (jdrop < n) {                                      /* a vertex of the initial simplex: the better of it and the pole becomes the pole */
                if (DAT(mp, np) <= f) { CW_SYNC(); if (lane == 0) W.x[jdrop] = SIM(jdrop, np); CW_SYNC(); }
                else {
                    const double rhocur = W.x[jdrop] - SIM(jdrop, np), xj = W.x[jdrop];
                    CW_SYNC();
                    if (lane == 0) SIM(jdrop, np) = xj;
                    for (k = lane; k <= mpp; k += CW_LANES) { DAT(k, jdrop) = DAT(k, np); DAT(k, np) = W.con[k]; }
                    for (k = lane; k <= jdrop; k += CW_LANES) {
                        double t = 0.;
                        SIM(jdrop, k) = -rhocur;
                        for (i = k; i <= jdrop; ++i) t -= SIMI(i, k);
                        SIMI(jdrop, k) = t;
                    }
                    CW_SYNC();
                }
            }
            if (nevals <= n) {                                    /* next vertex of the initial simplex */
                jdrop = nevals - 1;
                CW_SYNC();
                if (lane == 0) W.x[jdrop] += SIM(jdrop, jdrop);
                CW_SYNC();
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
            CW_SYNC();
            for (i = lane; i <= mpp; i += CW_LANES) { const double t = DAT(i, np); DAT(i, np) = DAT(i, nbest); DAT(i, nbest) = t; }
            for (i = lane; i < n; i += CW_LANES) {                /* row i of SIM, column i of SIMI: one lane each */
                const double t = SIM(i, nbest);
                double ta = 0.;
                SIM(i, nbest) = 0.;
                SIM(i, np) += t;
                for (k = 0; k < n; ++k) { SIM(i, k) -= t; ta -= SIMI(k, i); }
                SIMI(nbest, i) = ta;
            }
            CW_SYNC();
        }
        /* SIMI must still be the inverse (cobyla.c:735-761): the n^2 entries of SIMI x SIM - 1, one per lane ("many"); their greatest
         * magnitude is order-free unless a NaN is among them — then the reference's running comparison is replayed in its order */
        {
            double lerr = 0.;
            int lnan = 0;
            for (int e = lane; e < n * n; e += CW_LANES) {
                const int ii = e / n, jj = e - ii * n;
                double t = 0.;
                if (ii == jj) t += -1.;
                for (k = 0; k < n; ++k) { const double s = SIM(k, jj); if (s != 0) t += SIMI(ii, k) * s; }
                if (t != t) lnan = 1;
                lerr = lerr >= fabs(t) ? lerr : fabs(t);
            }
            if (lb_block_isum(lnan, S)) {
                for (i = 0; i < n; ++i)
                    for (j = 0; j < n; ++j) {
                        temp = 0.;
                        if (i == j) temp += -1.;
                        for (k = 0; k < n; ++k) if (SIM(k, j) != 0) temp += SIMI(i, k) * SIM(k, j);
                        error = error >= fabs(temp) ? error : fabs(temp);
                    }
            } else error = lb_block_max(lerr, S);
        }
        if (error > .1) { rc = CW_ROUNDOFF_LIMITED; go = FINISH_POLE; break; }
        /* gradients of the linear models (cobyla.c:763-787): (m+1) n sums of length n ("many") */
        CW_SYNC();
        for (k = lane; k <= mp; k += CW_LANES) W.con[k] = -DAT(k, np);
        CW_SYNC();
        for (int e = lane; e < (mp + 1) * n; e += CW_LANES) {
            const int kk = e / n, ii = e - kk * n;
            const double ck = W.con[kk];
            double t = 0.;
            for (j = 0; j < n; ++j) t += (DAT(kk, j) + ck) * SIMI(j, ii);
            if (kk == mp) t = -t;
            ACOL(ii, kk) = t;
        }
        /* is the simplex acceptable? (cobyla.c:789-811) */
        parsig = alpha * rho;
        pareta = beta * rho;
        {
            int lbad = 0;
            for (j = lane; j < n; j += CW_LANES) {
                double wsig = 0., weta = 0.;
                for (i = 0; i < n; ++i) { wsig += SIMI(j, i) * SIMI(j, i); weta += SIM(i, j) * SIM(i, j); }
                const double vs = 1. / sqrt(wsig), ve = sqrt(weta);
                W.vsig[j] = vs;
                W.veta[j] = ve;
                if (vs < parsig || ve > pareta) lbad = 1;
            }
            iflag = lb_block_isum(lbad, S) ? 0 : 1;
        }
        CW_SYNC();
        if (ibrnch == 1 || iflag == 1) { go = TRUST_STEP; break; }
        /* a repair step: drop the worst-placed vertex, step orthogonally to the opposite face (cobyla.c:813-914) */
        {
            double cvmaxp = 0., cvmaxm = 0., dxsign = 1.;
            jdrop = -1;
            temp = pareta;
            for (j = 0; j < n; ++j) if (W.veta[j] > temp) { jdrop = j; temp = W.veta[j]; }
            if (jdrop < 0) for (j = 0; j < n; ++j) if (W.vsig[j] < temp) { jdrop = j; temp = W.vsig[j]; }
            temp = gamma_ * rho * W.vsig[jdrop];
            CW_SYNC();
            for (i = lane; i < n; i += CW_LANES) W.dx[i] = temp * SIMI(jdrop, i);
            CW_SYNC();
            for (k = lane; k <= mp; k += CW_LANES) {
                double s = 0.;
                for (i = 0; i < n; ++i) s += ACOL(i, k) * W.dx[i];
                W.s1[k] = s;
            }
            CW_SYNC();
            for (k = 0; k <= mp; ++k) {
                sum = W.s1[k];
                if (k < mp) {
                    temp = DAT(k, np);
                    cvmaxp = cvmaxp >= -sum - temp ? cvmaxp : -sum - temp;
                    cvmaxm = cvmaxm >= sum - temp ? cvmaxm : sum - temp;
                }
            }
            if (parmu * (cvmaxp - cvmaxm) > sum + sum) dxsign = -1.;
            CW_SYNC();
            for (i = 0; i < n; ++i) {                             /* (the generator's state runs through the coordinates in order) */
                const double xi = SIM(i, np);
                double d = dxsign * W.dx[i] * cw_lcg_between(&seed, 0.01, 1);
                for (;;) {                                        /* keep the new vertex inside the box (cobyla.c:876-889) */
                    if (xi + d > W.sub[i]) d = -d;
                    if (xi + d < W.slb[i]) {
                        if (xi - d <= W.sub[i]) d = -d;
                        else { d *= 0.5; continue; }
                    }
                    break;
                }
                W.s2[i] = d;                                      /* (every lane stores the same value) */
            }
            CW_SYNC();
            for (i = lane; i < n; i += CW_LANES) { W.dx[i] = W.s2[i]; SIM(i, jdrop) = W.s2[i]; }
            CW_SYNC();
            cw_replace_vertex(W, jdrop, 1, lane);
            for (j = lane; j < n; j += CW_LANES) W.x[j] = SIM(j, np) + W.dx[j];
            CW_SYNC();
        }
        go = EVAL_PRE;
        break;
    }
    case TRUST_STEP: {                                            /* L370 (cobyla.c:918-1011) */
        double resnew = 0., barmu = 0.;
        int again = 0;
        rc = cw_trust_lp(W, rho, &ifull, lane);
        if (rc != CW_SUCCESS) { go = FINISH_POLE; break; }
        CW_SYNC();
        for (i = lane; i < n; i += CW_LANES) {                    /* (paranoia of the reference: the bound rows are linear) */
            const double xi = SIM(i, np);
            if (xi + W.dx[i] > W.sub[i]) W.dx[i] = W.sub[i] - xi;
            if (xi + W.dx[i] < W.slb[i]) W.dx[i] = xi - W.slb[i];
        }
        CW_SYNC();
        if (ifull == 0) {
            temp = 0.;
            for (i = 0; i < n; ++i) temp += W.dx[i] * W.dx[i];
            if (temp < rho * .25 * rho) { ibrnch = 1; go = SHRINK; break; }
        }
        /* predicted change of f and of the greatest violation (cobyla.c:952-967) */
        CW_SYNC();
        if (lane == 0) W.con[mp] = 0.;
        CW_SYNC();
        for (k = lane; k <= mp; k += CW_LANES) {
            double s = W.con[k];
            for (i = 0; i < n; ++i) s -= ACOL(i, k) * W.dx[i];
            W.s1[k] = s;
        }
        CW_SYNC();
        for (k = 0; k <= mp; ++k) { sum = W.s1[k]; if (k < mp) resnew = resnew >= sum ? resnew : sum; }
        /* raise the penalty parameter if necessary; if that changes the pole, start over from there (cobyla.c:969-1001) */
        prerec = DAT(mpp, np) - resnew;
        if (prerec > 0.) barmu = sum / prerec;
        if (parmu < barmu * 1.5) {
            parmu = barmu * 2.;
            const double phi = DAT(mp, np) + parmu * DAT(mpp, np);
            for (j = 0; j < n && !again; ++j) {
                temp = DAT(mp, j) + parmu * DAT(mpp, j);
                if (temp < phi) again = 1;
                else if (temp == phi && parmu == 0.) { if (DAT(mpp, j) < DAT(mpp, np)) again = 1; }
            }
            if (again) { go = POLE; break; }
        }
        prerem = parmu * prerec - sum;
        CW_SYNC();
        for (i = lane; i < n; i += CW_LANES) W.x[i] = SIM(i, np) + W.dx[i];
        CW_SYNC();
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
        CW_SYNC();
        for (j = lane; j < n; j += CW_LANES) {                    /* |row j of SIMI . dx|, and the edge the new vertex would make with vertex j */
            double t = 0., e2 = 0.;
            for (i = 0; i < n; ++i) t += SIMI(j, i) * W.dx[i];
            for (i = 0; i < n; ++i) { const double d = W.dx[i] - SIM(i, j); e2 += d * d; }
            W.s1[j] = fabs(t);
            W.s2[j] = sqrt(e2);
        }
        CW_SYNC();
        for (j = 0; j < n; ++j) {
            temp = W.s1[j];
            if (temp > ratio) { jdrop = j; ratio = temp; }
        }
        for (j = lane; j < n; j += CW_LANES) W.sigbar[j] = W.s1[j] * W.vsig[j];
        CW_SYNC();
        edgmax = delta * rho;
        for (j = 0; j < n; ++j)
            if (W.sigbar[j] >= parsig || W.sigbar[j] >= W.vsig[j]) {
                temp = trured > 0. ? W.s2[j] : W.veta[j];
                if (temp > edgmax) { l = j; edgmax = temp; }
            }
        if (l >= 0) jdrop = l;
        if (jdrop < 0) { go = SHRINK; break; }
        cw_replace_vertex(W, jdrop, 0, lane);
        for (k = lane; k <= mpp; k += CW_LANES) DAT(k, jdrop) = W.con[k];
        CW_SYNC();
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
        if (fbest < minf && cw_tol_reached(minf, fbest, P.ftol_rel, P.ftol_abs)) { rc = CW_FTOL_REACHED; go = FINISH_POLE; break; }
        minf = fbest;
        if (rho > rhoend) {
            rho *= .5;
            if (rho <= rhoend * 1.5) rho = rhoend;
            if (parmu > 0.) {
                double denom = 0., cmin = 0., cmax = 0.;
                CW_SYNC();
                for (k = lane; k <= mp; k += CW_LANES) {          /* the spread of every row over the vertices ("many") */
                    double lo = DAT(k, np), hi = lo;
                    for (i = 0; i < n; ++i) {
                        const double d = DAT(k, i);
                        lo = lo <= d ? lo : d;
                        hi = hi >= d ? hi : d;
                    }
                    W.s1[k] = lo; W.s2[k] = hi;
                }
                CW_SYNC();
                for (k = 0; k <= mp; ++k) {
                    cmin = W.s1[k];
                    cmax = W.s2[k];
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
        rc = rhoend > 0 ? CW_XTOL_REACHED : CW_ROUNDOFF_LIMITED;
        go = ifull == 1 ? FINISH_HERE : FINISH_POLE;
        break;
    }
    case FINISH_POLE:                                             /* L600 */
        CW_SYNC();
        for (i = lane; i < n; i += CW_LANES) W.x[i] = SIM(i, np);
        f = DAT(mp, np);
        CW_SYNC();
        /* fall through */
    case FINISH_HERE:                                             /* L620 */
        minf = f;
        go = FINISHED;
        break;
    }
    /* cobyla.c:255-263: unscale, clip; then the dispatcher's memoized best point (optimize.c:1064-1071) */
    CW_SYNC();
    for (j = lane; j < n; j += CW_LANES) {
        double v = W.x[j] * W.scale[j];
        if (v < lb[j]) v = lb[j];
        if (v > ub[j]) v = ub[j];
        x0[j] = bestf < DBL_MAX ? W.bestx[j] : v;
    }
    if (lane == 0) {
        out[inst].f = bestf < DBL_MAX ? bestf : minf;
        out[inst].ret = rc; out[inst].nevals = nevals; out[inst].iterm = nevals; out[inst].cols = 0;
        if (P.done) __hip_atomic_fetch_add(P.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    (void) tempa; (void) pareta;
}

/* the searches keep their state in LDS: no workspace in global memory (the sizes stay in the interface for the allocation's sake) */
extern "C" size_t nla_cobyla_work_doubles(int n, int ld, int count) { (void) n; (void) ld; return (size_t) (count > 0 ? count : 1); }
extern "C" size_t nla_cobyla_work_ints(int n, int count) { (void) n; return (size_t) (count > 0 ? count : 1); }
/* bytes of LDS one search of n variables inside a fully finite box needs (m = 2n rows); the kernel serves n while this fits the
 * compute unit's 160 KB beside the fixed buffers */
extern "C" size_t nla_cobyla_lds_bytes(int n) { return sizeof(double) * cw_lds_doubles(n, 2 * n) + 2048; }
extern "C" int nla_cobyla_fits(int n) { return n >= 1 && nla_cobyla_lds_bytes(n) <= 160 * 1024; }

extern "C" int nla_k_cobyla_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *dx, double *X,
                                  double *work, int *iwork, const nla_cobyla_params *params, nla_lbfgs_result *out, void *stream)
{
    (void) work; (void) iwork;
    if (count <= 0) return 0;
    if (obj < 0 || n < 1 || ld < n || !nla_cobyla_fits(n)) return (int) hipErrorInvalidValue;
    hipStream_t st = (hipStream_t) stream;
    nla_cobyla_params P = *params;
    if (P.sign == 0.) P.sign = 1.;
    const size_t lds = sizeof(double) * cw_lds_doubles(n, 2 * n);
#ifdef NLA_SIMT_EMU
#define CALL(O) hipLaunchKernelGGL((cobyla_batch_kernel<O>), dim3(count), dim3(CW_LANES), lds, st, n, ld, count, lb, ub, dx, X, P, out)
#else
#define CALL(O) do { if (lds > 48 * 1024) { hipError_t e_ = hipFuncSetAttribute((const void *) cobyla_batch_kernel<O>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds); \
                                            if (e_ != hipSuccess) return (int) e_; } \
                     hipLaunchKernelGGL((cobyla_batch_kernel<O>), dim3(count), dim3(CW_LANES), lds, st, n, ld, count, lb, ub, dx, X, P, out); } while (0)
#endif
    NLA_OBJ_DISPATCH(obj, CALL)
#undef CALL
    NLA_LAUNCH_CHECK();
    return 0;
}
