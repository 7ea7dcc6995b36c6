/* port_mlsl.c — CPU ORACLE (test infrastructure): Multi-Level Single-Linkage global optimisation
 * (src/algs/mlsl/mlsl.c:251-438) with the pseudo-random sampler (lds = 0; the reference's LDS mode
 * silently is pseudo-random too for n > 1111) and NLOPT_LD_LBFGS (port_lbfgs.c) or
 * NLOPT_LD_MMA (port_mma.c, the GD_MLSL default) as the local optimiser, called the way nlopt_optimize_limited would call it
 * (src/api/optimize.c:1087-1113, :514-566, :716-718).
 *
 * Containers: the reference keeps points and local minima in red-black trees ordered by f
 * (mlsl.c:102-115; equal keys go to the LEFT of existing ones, redblack.c:120); here they are
 * arrays kept sorted with the same tie rule — only the order matters.
 */
#include "port_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double f; int minimized; double closest_pt_d, closest_lm_d; double *x; int id; } pt;
typedef struct { double f; double *x; } lm_t;

typedef struct {
    int n;
    orc_func f; void *f_data;
    orc_stop *stop;
} counted;
static double fcount(unsigned n, const double *x, double *grad, void *p_)     /* mlsl.c:246-251 */
{
    counted *c = (counted *) p_;
    ++c->stop->nevals;
    return c->f(n, x, grad, c->f_data);
}

static double distance2(int n, const double *a, const double *b)              /* mlsl.c:118-127 */
{
    double d = 0.;
    for (int i = 0; i < n; ++i) { double dx = a[i] - b[i]; d += dx * dx; }
    return d;
}

#define K2PI (6.2831853071795864769252867665590057683943388)
static double gam(int n) { double z = n / 2; return sqrt(pow(K2PI * z, 1.0 / n) * z) * exp(-0.5); }   /* mlsl.c:227-237, integer n/2 */

/* lds != 0: NLOPT_G*_MLSL_LDS — Sobol points instead of urand when a generator exists for this n (mlsl.c:306,332,355-359) */
static int mlsl_lds = 0;
void orc_mlsl_set_lds(int lds) { mlsl_lds = lds; }

int orc_mlsl_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                      orc_stop *stop, int Nsamples, const orc_local_params *loc, orc_mlsl_trace *trace)
{
    const double MLSL_SIGMA = 2., MLSL_GAMMA = 0.3, dlm = 1.0, dbound = 1e-6;
    int ret = ORC_SUCCESS, N = Nsamples ? Nsamples : 4, i, j;
    pt **pts = NULL; size_t npts = 0, cappts = 0;
    lm_t *lms = NULL; size_t nlms = 0, caplms = 0;
    double R_prefactor;
    counted cnt;
    int next_id = 0;
    orc_sobol *sob = mlsl_lds ? orc_sobol_create((unsigned) n) : NULL;
    if (N < 1) { orc_sobol_destroy(sob); return ORC_INVALID_ARGS; }
    cnt.n = n; cnt.f = f; cnt.f_data = f_data; cnt.stop = stop;
    R_prefactor = sqrt(2. / K2PI) * pow(gam(n) * MLSL_SIGMA, 1.0 / n);
    for (i = 0; i < n; ++i) R_prefactor *= pow(ub[i] - lb[i], 1.0 / n);

#define INSERT_PT(P) do { size_t pos_ = 0; if (npts == cappts) { cappts = cappts ? 2 * cappts : 1024; pts = (pt **) realloc(pts, cappts * sizeof *pts); } \
        while (pos_ < npts && pts[pos_]->f < (P)->f) ++pos_;   /* before the first element that is not smaller */ \
        memmove(pts + pos_ + 1, pts + pos_, (npts - pos_) * sizeof *pts); pts[pos_] = (P); ++npts; } while (0)
#define NEWPT(P) do { (P) = (pt *) malloc(sizeof(pt)); (P)->x = (double *) malloc(sizeof(double) * (size_t) n); (P)->minimized = 0; (P)->id = next_id++; \
        (P)->closest_pt_d = HUGE_VAL; (P)->closest_lm_d = HUGE_VAL; } while (0)
#define STOPS(fv) do { if (stop->force_stop) ret = ORC_FORCED_STOP; else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED; \
        else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED; else if ((fv) < stop->minf_max) ret = ORC_STOPVAL_REACHED; } while (0)
#define GET_MINF() do { if (npts) { *minf = pts[0]->f; memcpy(x, pts[0]->x, sizeof(double) * (size_t) n); } \
        if (nlms && lms[0].f < *minf) { *minf = lms[0].f; memcpy(x, lms[0].x, sizeof(double) * (size_t) n); } } while (0)

    {
        pt *p;
        NEWPT(p);
        orc_sobol_skip(sob, (unsigned) (10 * n + N), p->x);                  /* mlsl.c:332 */
        memcpy(p->x, x, sizeof(double) * (size_t) n);
        p->f = f((unsigned) n, x, NULL, f_data);
        ++stop->nevals;
        INSERT_PT(p);
        STOPS(p->f);
    }
    while (ret == ORC_SUCCESS) {
        double R;
        size_t idx;
        GET_MINF();
        for (i = 0; i < N && ret == ORC_SUCCESS; ++i) {                      /* sampling phase, mlsl.c:349-374 */
            pt *p;
            size_t k;
            NEWPT(p);
            if (sob) orc_sobol_next(sob, p->x, lb, ub);
            else for (j = 0; j < n; ++j) p->x[j] = orc_urand(lb[j], ub[j]);
            p->f = f((unsigned) n, p->x, NULL, f_data);
            ++stop->nevals;
            if (trace && trace->nsamp < trace->cap) trace->fsamp[trace->nsamp] = p->f;
            if (trace) ++trace->nsamp;
            INSERT_PT(p);
            STOPS(p->f);
            if (ret != ORC_SUCCESS) break;
            for (k = 0; k < npts && pts[k]->f < p->f; ++k) {                 /* find_closest_pt: strictly smaller f */
                double d = distance2(n, p->x, pts[k]->x);
                if (d < p->closest_pt_d) p->closest_pt_d = d;
            }
            for (k = 0; k < nlms && lms[k].f < p->f; ++k) {                  /* find_closest_lm */
                double d = distance2(n, p->x, lms[k].x);
                if (d < p->closest_lm_d) p->closest_lm_d = d;
            }
            for (k = npts; k-- > 0 && pts[k]->f > p->f;)                     /* pts_update_newpt: strictly larger f */
                if (!pts[k]->minimized) {
                    double d = distance2(n, p->x, pts[k]->x);
                    if (d < pts[k]->closest_pt_d) pts[k]->closest_pt_d = d;
                }
        }
        R = R_prefactor * pow(log((double) npts) / npts, 1.0 / n);          /* mlsl.c:377-378 */
        idx = 0;
        for (i = (int) (ceil(MLSL_GAMMA * npts) + 0.5); idx < npts && i > 0 && ret == ORC_SUCCESS; --i, ++idx) {
            pt *p = pts[idx];
            int pot = !p->minimized && !(p->closest_pt_d <= R * R) && !(p->closest_lm_d <= (dlm * R) * (dlm * R));
            if (pot) for (j = 0; j < n; ++j)
                if ((p->x[j] - lb[j] <= dbound * R || ub[j] - p->x[j] <= dbound * R) && ub[j] - lb[j] > dbound * R) { pot = 0; break; }
            if (pot) {
                orc_stop ls;
                double *lx, lf;
                int lret;
                size_t pos, k;
                long limited;
                if (stop->force_stop) { ret = ORC_FORCED_STOP; break; }
                if (orc_stop_evals(stop)) { ret = ORC_MAXEVAL_REACHED; break; }
                lx = (double *) malloc(sizeof(double) * (size_t) n);
                memcpy(lx, p->x, sizeof(double) * (size_t) n);
                /* nlopt_optimize_limited -> nlopt_optimize -> luksan_plis with the local optimiser's own stop struct */
                orc_stop_default(&ls, (unsigned) n);
                ls.minf_max = stop->minf_max;                                /* nlopt_set_stopval(local_opt, stop->minf_max), mlsl.c:306 */
                ls.ftol_rel = loc->ftol_rel; ls.ftol_abs = loc->ftol_abs; ls.xtol_rel = loc->xtol_rel;
                limited = stop->maxeval - stop->nevals;
                ls.maxeval = loc->maxeval;
                if (loc->maxeval <= 0 || (limited > 0 && limited < loc->maxeval)) ls.maxeval = limited;
                ls.nevals = 0;
                lf = HUGE_VAL;
                lret = loc->alg == 1 ? orc_mma_minimize(n, fcount, &cnt, lb, ub, lx, &lf, &ls, &loc->mma)
                                     : orc_lbfgs_minimize(n, fcount, &cnt, lb, ub, lx, &lf, &ls, loc->mf, loc->tolg);
                p->minimized = 1;
                if (trace && trace->nloc < trace->cap) { trace->floc[trace->nloc] = lf; trace->eloc[trace->nloc] = (int) ls.nevals; if (trace->sloc) trace->sloc[trace->nloc] = p->id; }
                if (trace) ++trace->nloc;
                if (lret < 0) { free(lx); ret = lret; goto done; }
                if (nlms == caplms) { caplms = caplms ? 2 * caplms : 256; lms = (lm_t *) realloc(lms, caplms * sizeof *lms); }
                for (pos = 0; pos < nlms && lms[pos].f < lf; ++pos) { }
                memmove(lms + pos + 1, lms + pos, (nlms - pos) * sizeof *lms);
                lms[pos].f = lf; lms[pos].x = lx; ++nlms;
                if (stop->force_stop) ret = ORC_FORCED_STOP;
                else if (lf < stop->minf_max) ret = ORC_STOPVAL_REACHED;
                else if (orc_stop_evals(stop)) ret = ORC_MAXEVAL_REACHED;
                else if (orc_stop_time(stop)) ret = ORC_MAXTIME_REACHED;
                else
                    for (k = npts; k-- > 0 && pts[k]->f > lf;)               /* pts_update_newlm */
                        if (!pts[k]->minimized) {
                            double d = distance2(n, lx, pts[k]->x);
                            if (d < pts[k]->closest_lm_d) pts[k]->closest_lm_d = d;
                        }
            }
        }
        if (trace) {
            if (trace->it_nloc && (size_t) trace->iterations < trace->it_cap) {
                trace->it_nloc[trace->iterations] = (long) trace->nloc; trace->it_nevals[trace->iterations] = stop->nevals;
                trace->it_words[trace->iterations] = orc_mt_words_drawn();
            }
            ++trace->iterations;
        }
    }
    GET_MINF();
done:
    for (size_t k = 0; k < npts; ++k) { free(pts[k]->x); free(pts[k]); }
    for (size_t k = 0; k < nlms; ++k) free(lms[k].x);
    free(pts); free(lms); orc_sobol_destroy(sob);
    return ret;
}
