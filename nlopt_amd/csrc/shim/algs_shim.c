/* algs_shim.c — the SECONDARY drop-in boundary: the three algorithm entry points the reference's
 * dispatcher calls for this path, with the reference's own names and signatures,
 *
 *     crs_minimize    src/algs/crs/crs.h:34-40       (called at src/api/optimize.c:744-747)
 *     isres_minimize  src/algs/isres/isres.h:34-41   (optimize.c:941-944)
 *     mlsl_minimize   src/algs/mlsl/mlsl.h:34-41     (optimize.c:749-793)
 *
 * so that the REAL libnlopt keeps its whole API shell (object, setters, dispatcher, bindings) and only these algorithms run on the
 * MI355X:  LD_PRELOAD=libnlopt_algs_amd.so  in front of an unmodified libnlopt.so (its calls to the three functions go through the PLT),
 * or the three objects replaced in a rebuild.  nlopt_stopping (util/nlopt-util.h:79-91) and nlopt_constraint (:118-125) are laid out as
 * the library's own nla_stopping / nla_constraint, field for field.
 *
 * The algorithms and the API shell share ONE Mersenne Twister (nlopt_srand seeds what crs_minimize draws from): the shim therefore also
 * provides the generator's entry points of util/mt19937ar.c + api/general.c — nlopt_srand, nlopt_srand_time, nlopt_srand_time_default,
 * nlopt_init_genrand, nlopt_urand, nlopt_iurand, nlopt_nrand — over the library's generator (same algorithm, same sequences), so the
 * whole process draws from it.
 *
 * Built by nlopt_amd/_build.py into lib/libnlopt_algs_amd.so: the product's objects + this file, -Bsymbolic (its own nlopt_* calls bind
 * inside), only the names above exported (shim.map).  The objective it is handed is the dispatcher's wrapper — a host callback: the
 * exact host-callback path (the device builds every candidate, f is called on the caller's thread in the reference's order). */
#include "../nla_internal.h"
#include <stdlib.h>
#include <string.h>

/* the reference's optimiser object as far as an algorithm reads it (api/nlopt-internal.h:40-88); the library's own struct nlopt_opt_s has
 * the same fields in the same order up to vector_storage */
typedef struct {
    nlopt_algorithm algorithm; unsigned n;
    nlopt_func f; void *f_data; nlopt_precond pre; int maximize;
    struct { char *name; double val; } *params; unsigned nparams;
    double *lb, *ub;
    unsigned m, m_alloc; void *fc;
    unsigned p, p_alloc; void *h;
    nlopt_munge munge_on_destroy, munge_on_copy;
    double stopval, ftol_rel, ftol_abs, xtol_rel, *xtol_abs, *x_weights;
    int maxeval, numevals;
    double maxtime;
    int force_stop;
    void *force_stop_child;
    void *local_opt;
    unsigned stochastic_population;
    double *dx;
    unsigned vector_storage;
} ref_opt_view;

/* the caller's stopping criteria on THIS library's clock: stop->start is a reading of the caller's nlopt_seconds() (util/timer.c: seconds
 * since ITS first call), taken by the dispatcher right in front of the algorithm (optimize.c:1000-1002) — the algorithms here compare it
 * with nla_seconds().  Everything else of the struct is pointers into the caller's object (evaluation counter, force_stop flag, message). */
static nla_stopping on_this_clock(const nla_stopping *stop)
{
    nla_stopping s = *stop;
    s.start = nla_seconds();
    return s;
}

nlopt_result crs_minimize(int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                          nla_stopping *stop, int population, int lds)
{
    if (lds) { nla_stop_msg(stop, "libnlopt_algs_amd: crs_minimize with a low-discrepancy initial population is not provided"); return NLOPT_INVALID_ARGS; }
    nla_stopping s = on_this_clock(stop);
    return nla_crs_minimize(NULL, n, f, f_data, lb, ub, x, minf, &s, population);
}

nlopt_result isres_minimize(int n, nlopt_func f, void *f_data, int m, nla_constraint *fc, int p, nla_constraint *h, const double *lb,
                            const double *ub, double *x, double *minf, nla_stopping *stop, int population)
{
    nla_stopping s = on_this_clock(stop);
    return nla_isres_minimize(NULL, n, f, f_data, m, fc, p, h, lb, ub, x, minf, &s, population);
}

nlopt_result mlsl_minimize(int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                           nla_stopping *stop, void *ref_local_opt, int Nsamples, int lds)
{
    const ref_opt_view *r = (const ref_opt_view *) ref_local_opt;
    nlopt_opt loc;
    nlopt_result ret;
    unsigned i;
    if (!r) { nla_stop_msg(stop, "libnlopt_algs_amd: mlsl_minimize needs a local optimiser"); return NLOPT_INVALID_ARGS; }
    /* the local optimiser as an object of THIS library (the dispatcher's is the reference's: same fields, its own allocator and hooks) */
    loc = nlopt_create(r->algorithm, (unsigned) n);
    if (!loc) return NLOPT_OUT_OF_MEMORY;
    nlopt_set_ftol_rel(loc, r->ftol_rel); nlopt_set_ftol_abs(loc, r->ftol_abs); nlopt_set_xtol_rel(loc, r->xtol_rel);
    if (r->xtol_abs) nlopt_set_xtol_abs(loc, r->xtol_abs);
    if (r->x_weights) nlopt_set_x_weights(loc, r->x_weights);
    nlopt_set_maxeval(loc, r->maxeval); nlopt_set_maxtime(loc, r->maxtime);
    if (r->dx) nlopt_set_initial_step(loc, r->dx);
    nlopt_set_vector_storage(loc, r->vector_storage);
    for (i = 0; i < r->nparams; ++i) nlopt_set_param(loc, r->params[i].name, r->params[i].val);
    {
        nla_stopping s = on_this_clock(stop);
        ret = nla_mlsl_minimize(NULL, n, f, f_data, lb, ub, x, minf, &s, loc, Nsamples, lds);
    }
    nlopt_destroy(loc);
    return ret;
}

/* the generator the API shell and the algorithms share (util/mt19937ar.c:80-95,203-232; api/general.c:231-246): nlopt_srand, nlopt_srand_time,
 * nlopt_urand, nlopt_iurand, nlopt_nrand are mt_host.c's, exported as they are */
void nlopt_init_genrand(unsigned long s) { nla_init_genrand(s); }
void nlopt_srand_time_default(void) { nla_srand_time_default(); }
