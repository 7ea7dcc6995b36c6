/* stopping.c — stopping criteria, timers and small numeric predicates of the host side.
 * Semantics of the reference's src/util/stop.c:81-159 (relstop, nlopt_stop_ftol/f/x/dx/evals/
 * time/forced), :207-215 (stop_msg), :219-266 (isinf/istiny) and src/util/timer.c (seconds since
 * the first call per thread, time-based seed). */
#include "nla_internal.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

int nla_isinf(double x) { return fabs(x) >= HUGE_VAL * 0.99 || isinf(x); }          /* stop.c:219-227 */
int nla_istiny(double x) { return x == 0.0 || fpclassify(x) == FP_SUBNORMAL; }      /* stop.c:240-254 */

static int tol_reached(double vold, double vnew, double reltol, double abstol)       /* stop.c:81-86 */
{
    double d;
    if (nla_isinf(vold)) return 0;
    d = fabs(vnew - vold);
    return d < abstol || d < reltol * (fabs(vnew) + fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

int nla_stop_ftol(const nla_stopping *s, double f, double oldf) { return tol_reached(oldf, f, s->ftol_rel, s->ftol_abs); }
int nla_stop_f(const nla_stopping *s, double f, double oldf) { return f <= s->minf_max || nla_stop_ftol(s, f, oldf); }

static double weighted_l1(unsigned n, const double *v, const double *w)               /* stop.c:37-57 */
{
    double r = 0;
    unsigned i;
    if (w) for (i = 0; i < n; ++i) r += w[i] * fabs(v[i]);
    else   for (i = 0; i < n; ++i) r += fabs(v[i]);
    return r;
}

int nla_stop_x(const nla_stopping *s, const double *x, const double *oldx)            /* stop.c:98-108 */
{
    double d = 0;
    unsigned i;
    if (s->x_weights) for (i = 0; i < s->n; ++i) d += s->x_weights[i] * fabs(x[i] - oldx[i]);
    else              for (i = 0; i < s->n; ++i) d += fabs(x[i] - oldx[i]);
    if (d < s->xtol_rel * weighted_l1(s->n, x, s->x_weights)) return 1;
    if (!s->xtol_abs) return 0;
    for (i = 0; i < s->n; ++i)
        if (fabs(x[i] - oldx[i]) >= s->xtol_abs[i]) return 0;
    return 1;
}

int nla_stop_dx(const nla_stopping *s, const double *x, const double *dx)             /* stop.c:110-120 */
{
    unsigned i;
    if (weighted_l1(s->n, dx, s->x_weights) < s->xtol_rel * weighted_l1(s->n, x, s->x_weights)) return 1;
    if (!s->xtol_abs) return 0;
    for (i = 0; i < s->n; ++i)
        if (fabs(dx[i]) >= s->xtol_abs[i]) return 0;
    return 1;
}

int nla_stop_evals(const nla_stopping *s) { return s->maxeval > 0 && *(s->nevals_p) >= s->maxeval; }   /* :136-139 */
int nla_stop_time(const nla_stopping *s) { return s->maxtime > 0 && nla_seconds() - s->start >= s->maxtime; } /* :141-149 */
int nla_stop_forced(const nla_stopping *s) { return s->force_stop && *(s->force_stop); }               /* :156-159 */

char *nla_vsprintf(char *p, const char *fmt, va_list ap)                              /* stop.c:186-205 */
{
    size_t len = strlen(fmt) + 128;
    for (;;) {
        va_list aq;
        int need;
        p = (char *) realloc(p, len);
        if (!p) abort();
        va_copy(aq, ap);
        need = vsnprintf(p, len, fmt, aq);
        va_end(aq);
        if (need >= 0 && (size_t) need < len) return p;
        len = need >= 0 ? (size_t) need + 1 : (len * 3) >> 1;
    }
}

void nla_stop_msg(const nla_stopping *s, const char *fmt, ...)                        /* stop.c:207-215 */
{
    va_list ap;
    if (!s->stop_msg) return;
    va_start(ap, fmt);
    *(s->stop_msg) = nla_vsprintf(*(s->stop_msg), fmt, ap);
    va_end(ap);
}

/* seconds since the first call on this thread (src/util/timer.c:38-63) */
double nla_seconds(void)
{
    static __thread int started = 0;
    static __thread struct timeval t0;
    struct timeval tv;
    if (!started) { started = 1; gettimeofday(&t0, NULL); }
    gettimeofday(&tv, NULL);
    return (double) (tv.tv_sec - t0.tv_sec) + 1.e-6 * (double) (tv.tv_usec - t0.tv_usec);
}

unsigned long nla_time_seed(void)                                                     /* timer.c:70-92 */
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (unsigned long) (tv.tv_sec ^ tv.tv_usec);
}

long nla_thread_id(void) { return (long) syscall(SYS_gettid); }
