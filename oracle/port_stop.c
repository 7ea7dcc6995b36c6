/* port_stop.c — CPU ORACLE (test infrastructure): stopping criteria.
 * Restates src/util/stop.c:81-159 (relstop, nlopt_stop_ftol/f/x/dx/evals/time) on orc_stop. */
#include "port_oracle.h"
#include <math.h>
#include <sys/time.h>

double orc_seconds(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double) tv.tv_sec + 1e-6 * (double) tv.tv_usec;
}

void orc_stop_default(orc_stop *s, unsigned n)
{
    s->n = n; s->minf_max = -HUGE_VAL; s->ftol_rel = s->ftol_abs = s->xtol_rel = 0;
    s->xtol_abs = s->x_weights = NULL; s->nevals = 0; s->maxeval = 0; s->maxtime = 0;
    s->start = orc_seconds(); s->force_stop = 0;
}

static int is_inf(double v) { return fabs(v) >= HUGE_VAL * 0.99 || isinf(v); }   /* stop.c:219-227 */

static int rel_or_abs(double vold, double vnew, double reltol, double abstol)      /* stop.c:81-86 */
{
    double d;
    if (is_inf(vold)) return 0;
    d = fabs(vnew - vold);
    return d < abstol || d < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

int orc_stop_ftol(const orc_stop *s, double f, double oldf) { return rel_or_abs(oldf, f, s->ftol_rel, s->ftol_abs); }
int orc_stop_f(const orc_stop *s, double f, double oldf) { return f <= s->minf_max || orc_stop_ftol(s, f, oldf); }

static double wnorm(unsigned n, const double *v, const double *w)                 /* stop.c:37-57 */
{
    double r = 0;
    for (unsigned i = 0; i < n; ++i) r += w ? w[i] * fabs(v[i]) : fabs(v[i]);
    return r;
}

int orc_stop_x(const orc_stop *s, const double *x, const double *oldx)            /* stop.c:98-108 */
{
    double d = 0;
    for (unsigned i = 0; i < s->n; ++i)
        d += s->x_weights ? s->x_weights[i] * fabs(x[i] - oldx[i]) : fabs(x[i] - oldx[i]);
    if (d < s->xtol_rel * wnorm(s->n, x, s->x_weights)) return 1;
    if (!s->xtol_abs) return 0;
    for (unsigned i = 0; i < s->n; ++i)
        if (fabs(x[i] - oldx[i]) >= s->xtol_abs[i]) return 0;
    return 1;
}

int orc_stop_dx(const orc_stop *s, const double *x, const double *dx)             /* stop.c:110-120 */
{
    if (wnorm(s->n, dx, s->x_weights) < s->xtol_rel * wnorm(s->n, x, s->x_weights)) return 1;
    if (!s->xtol_abs) return 0;
    for (unsigned i = 0; i < s->n; ++i)
        if (fabs(dx[i]) >= s->xtol_abs[i]) return 0;
    return 1;
}

int orc_stop_evals(const orc_stop *s) { return s->maxeval > 0 && s->nevals >= s->maxeval; }   /* :136-139 */
int orc_stop_time(const orc_stop *s) { return s->maxtime > 0 && orc_seconds() - s->start >= s->maxtime; } /* :141-149 */
