/* lbfgs_driver.c — NLOPT_LD_LBFGS behind the reference's entry point luksan_plis (plis.c:420-510):
 * argument handling on the host, the optimisation itself in the batched device kernel
 * (hip/lbfgs_kernels.hip) — used directly by nlopt_optimize(LD_LBFGS) with count = 1 and by
 * MLSL (mlsl_driver.c) with one workgroup per start point.  The batch context (nla_local_ctx) also
 * carries the second local optimiser, LD_MMA (mma_driver.c, hip/mma_kernels.hip).
 *
 * Three kinds of objective (nla_evaluator):
 *   device    a compiled-in device objective (nlopt_amd_objective): evaluated inside the kernel, a whole search is
 *             one launch; the host only watches the clock / the force_stop flag while it waits and raises the
 *             kernel's abort flag (plis.c:263,273,371, pssubs.c:914);
 *   host      any other nlopt_func: the reference's callback contract (nlopt.h:60-62) — f is called on the caller's
 *             thread, one x at a time, in the reference's order.  The kernel runs as a coroutine (include/nlopt_amd.h
 *             "External evaluation"): every vector operation of the search stays on the device, only x travels to the
 *             host and f / gradient back, once per evaluation;
 *   user      a device objective supplied by the user as a code object (nlopt_amd_set_device_objective, userobj.c):
 *             the same coroutine, the evaluations of all waiting searches of a batch made by one launch of the user's
 *             kernel. */
#include "nla_internal.h"
#include "objfuncs.h"
#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MEMAVAIL 1310720                       /* luksan.h:149 */

int nla_lbfgs_default_mf(int n, int mf, int maxeval)        /* plis.c:441-445 */
{
    if (mf <= 0) {
        mf = MEMAVAIL / n > 10 ? MEMAVAIL / n : 10;
        if (maxeval > 0 && maxeval <= mf) mf = maxeval > 1 ? maxeval : 1;
    }
    return mf;
}

/* a reusable batch: device buffers for up to `cap` simultaneous local searches by LD_LBFGS (alg 0), LD_MMA (alg 1) or LN_COBYLA (alg 2) */
struct nla_local_ctx {
    int alg, n, ld, cap, mf;
    nla_evaluator ev;
    nla_mma_params mma;            /* alg 1: the algorithm's own parameters (the stopping values come with each run) */
    const double *d_sigma_init;    /* alg 1: initial step on the device, or NULL */
    void *st;
    const double *d_lb, *d_ub;
    double *d_X, *d_work, *d_hist;
    int *d_iwork;
    nla_lbfgs_result *d_res;
    void *ev0, *ev1;
    nlopt_amd_stats *stats;        /* optional: device time / algorithmic bytes of the launches are added here */
    /* options of a run (nla_local_ctx_set_options) */
    int exact;
    double *d_xtol_abs, *d_x_weights;
    int32_t *h_abort;              /* pinned, device-visible: 0, 100 (maxtime) or -999 (forced stop) */
    /* external evaluation */
    nla_local_ext ext;             /* device buffers */
    nla_local_req *h_req;          /* pinned */
    double *h_x, *h_g, *h_f;       /* pinned: one point, one gradient, cap values */
    int32_t *h_list, *d_list;      /* user objective: indices of the waiting searches */
    double *d_ftrace; int64_t ftrace_cap;   /* optional per-evaluation f trace (nla_local_ctx_set_ftrace) */
    int (*after_launch)(void *); void *after_arg;   /* one-shot: called right behind the next run's kernel launch (nla_local_ctx_after_launch) */
    int32_t *d_done; int32_t done_expected;         /* optional (nla_local_ctx_finished_counter): searches ended so far, on the device / launched so far */
    /* alg 2: batches too small to fill the device go through the host algorithm (cobyla_small_batch_on_host) */
    int cob_min_batch;
    double *h_lb, *h_ub, *h_dx, *h_xtol_abs, *h_rows;
};

void nla_local_ctx_destroy(nla_local_ctx *c)
{
    if (!c) return;
    nla_dev_free(c->d_X); nla_dev_free(c->d_work); nla_dev_free(c->d_iwork); nla_dev_free(c->d_hist); nla_dev_free(c->d_res);
    nla_dev_free(c->d_xtol_abs); nla_dev_free(c->d_x_weights);
    nla_dev_free(c->ext.req); nla_dev_free(c->ext.EX); nla_dev_free(c->ext.EG); nla_dev_free(c->ext.EF); nla_dev_free(c->ext.save);
    nla_dev_free(c->d_list); nla_dev_free(c->d_ftrace);
    if (c->d_done) nla_dev_free_uncached(c->d_done);
    nla_host_free(c->h_abort); nla_host_free(c->h_req); nla_host_free(c->h_x); nla_host_free(c->h_g); nla_host_free(c->h_f);
    nla_host_free(c->h_list);
    free(c->h_lb); free(c->h_ub); free(c->h_dx); free(c->h_xtol_abs); free(c->h_rows);
    nla_event_destroy(c->ev0); nla_event_destroy(c->ev1);
    free(c);
}

static int ctx_common(nla_local_ctx *c, const nla_evaluator *ev, int n, int cap, const double *d_lb, const double *d_ub, void *stream)
{
    const size_t save = c->alg == 1 ? nla_mma_save_bytes() : nla_lbfgs_save_bytes();
    c->ev = *ev; c->n = n; c->ld = (n + 1) & ~1; c->cap = cap; c->st = stream; c->d_lb = d_lb; c->d_ub = d_ub;
    c->d_X = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld * (size_t) cap);
    c->d_res = (nla_lbfgs_result *) nla_dev_malloc(sizeof(nla_lbfgs_result) * (size_t) cap);
    c->h_abort = (int32_t *) nla_host_malloc(sizeof(int32_t));
    c->ev0 = nla_event_create(); c->ev1 = nla_event_create();
    if (!c->d_X || !c->d_res || !c->h_abort || !c->ev0 || !c->ev1) return -1;
    *c->h_abort = 0;
    if (ev->kind != NLA_EVAL_DEVICE) {
        c->ext.req = (nla_local_req *) nla_dev_malloc(sizeof(nla_local_req) * (size_t) cap);
        c->ext.EX = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld * (size_t) cap);
        c->ext.EG = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld * (size_t) cap);
        c->ext.EF = (double *) nla_dev_malloc(sizeof(double) * (size_t) cap);
        c->ext.save = nla_dev_malloc(save * (size_t) cap);
        c->h_req = (nla_local_req *) nla_host_malloc(sizeof(nla_local_req) * (size_t) cap);
        c->h_x = (double *) nla_host_malloc(sizeof(double) * (size_t) c->ld);
        c->h_g = (double *) nla_host_malloc(sizeof(double) * (size_t) c->ld);
        c->h_f = (double *) nla_host_malloc(sizeof(double) * (size_t) cap);
        c->h_list = (int32_t *) nla_host_malloc(sizeof(int32_t) * (size_t) cap);
        c->d_list = (int32_t *) nla_dev_malloc(sizeof(int32_t) * (size_t) cap);
        if (!c->ext.req || !c->ext.EX || !c->ext.EG || !c->ext.EF || !c->ext.save || !c->h_req || !c->h_x || !c->h_g || !c->h_f ||
            !c->h_list || !c->d_list) return -1;
    }
    return 0;
}

nla_local_ctx *nla_local_ctx_create(const nla_evaluator *ev, int n, int cap, int mf, const double *d_lb, const double *d_ub, void *stream)
{
    nla_local_ctx *c = (nla_local_ctx *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->mf = mf;
    if (ctx_common(c, ev, n, cap, d_lb, d_ub, stream)) { nla_local_ctx_destroy(c); return NULL; }
    c->d_work = (double *) nla_dev_malloc(sizeof(double) * nla_lbfgs_work_doubles(c->ld, mf, cap));
    c->d_iwork = (int *) nla_dev_malloc(sizeof(int) * (size_t) c->ld * (size_t) cap);
    c->d_hist = (double *) nla_dev_malloc(sizeof(double) * nla_lbfgs_hist_doubles(c->ld, mf, cap));
    if (!c->d_work || !c->d_iwork || !c->d_hist) { nla_local_ctx_destroy(c); return NULL; }
    return c;
}
/* the same for LN_COBYLA with bound constraints only (alg 2; hip/cobyla_kernels.hip: GN_MLSL's default local optimiser, compiled-in device
 * objectives only): d_dx = the initial step on the device, or NULL = the default step of every start (options.c:921-946) */
nla_local_ctx *nla_local_ctx_create_cobyla(const nla_evaluator *ev, int n, int cap, const double *d_dx, const double *d_lb, const double *d_ub, void *stream)
{
    nla_local_ctx *c;
    if (ev->kind != NLA_EVAL_DEVICE) return NULL;
    c = (nla_local_ctx *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->alg = 2;
    c->d_sigma_init = d_dx;
    if (ctx_common(c, ev, n, cap, d_lb, d_ub, stream)) { nla_local_ctx_destroy(c); return NULL; }
    c->d_work = (double *) nla_dev_malloc(sizeof(double) * nla_cobyla_work_doubles(n, c->ld, cap));
    c->d_iwork = (int *) nla_dev_malloc(sizeof(int) * nla_cobyla_work_ints(n, cap));
    c->h_lb = (double *) malloc(sizeof(double) * (size_t) n); c->h_ub = (double *) malloc(sizeof(double) * (size_t) n);
    c->h_rows = (double *) malloc(sizeof(double) * (size_t) c->ld * (size_t) cap);
    if (d_dx) c->h_dx = (double *) malloc(sizeof(double) * (size_t) n);
    if (!c->d_work || !c->d_iwork || !c->h_lb || !c->h_ub || !c->h_rows || (d_dx && !c->h_dx) ||
        nla_memcpy_d2h(c->h_lb, d_lb, sizeof(double) * (size_t) n, stream) || nla_memcpy_d2h(c->h_ub, d_ub, sizeof(double) * (size_t) n, stream) ||
        (d_dx && nla_memcpy_d2h(c->h_dx, d_dx, sizeof(double) * (size_t) n, stream)) || nla_stream_sync(stream)) { nla_local_ctx_destroy(c); return NULL; }
    /* One wavefront walks a search's serial chain 4 - 20 times slower than a host core (measured per evaluation of ONE search, device
     * against the reference on the same box, profiles/r06_cobyla_batched.txt: n = 8 21.7 us against 1.0, n = 16 40.6 against 4.2,
     * n = 32 133 against 26, n = 48 380 against 88) and hundreds of them run side by side (2048 searches: 56x / 104x one core at n = 8 / 16):
     * the device is the faster place from about that many concurrent searches on, the host below */
    c->cob_min_batch = n < 12 ? 24 : n < 24 ? 12 : 6;
    return c;
}
/* ... the threshold can be set: 1 = every batch on the device, 0 / negative = the default above ("amd_cobyla_min_batch") */
void nla_local_ctx_set_cobyla_min_batch(nla_local_ctx *c, int min_batch) { if (c && c->alg == 2 && min_batch > 0) c->cob_min_batch = min_batch; }
/* the same for LD_MMA (mma_driver.c reads the parameters; sigma_init: device copy of the initial step or NULL) */
nla_local_ctx *nla_local_ctx_create_mma(const nla_evaluator *ev, int n, int cap, const nla_mma_params *alg_params, const double *d_sigma_init,
                                        const double *d_lb, const double *d_ub, void *stream)
{
    nla_local_ctx *c = (nla_local_ctx *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->alg = 1;
    c->mma = *alg_params; c->d_sigma_init = d_sigma_init;
    if (ctx_common(c, ev, n, cap, d_lb, d_ub, stream)) { nla_local_ctx_destroy(c); return NULL; }
    c->d_work = (double *) nla_dev_malloc(sizeof(double) * nla_mma_work_doubles(c->ld, cap));
    if (!c->d_work) { nla_local_ctx_destroy(c); return NULL; }
    return c;
}
void nla_local_ctx_set_stats(nla_local_ctx *c, nlopt_amd_stats *stats) { if (c) c->stats = stats; }
/* work the caller wants on the context's stream RIGHT BEHIND the searches' kernel — enqueued before the host waits for the kernel, so that
 * it starts the moment the last search ends (MLSL: the minimisers' distances to the point set).  One-shot; device objectives only (a
 * run that goes through host evaluations calls it before it returns, with the searches finished).  A nonzero return fails the run. */
void nla_local_ctx_after_launch(nla_local_ctx *c, int (*fn)(void *), void *arg) { if (c) { c->after_launch = fn; c->after_arg = arg; } }

/* device objectives: a counter on the device that every search of this context's launches adds 1 to when it ends (never reset).
 * nla_local_ctx_count_finished switches it on (0, or -1: no memory / searches that evaluate on the host); nla_local_ctx_finished_counter
 * returns it (NULL: not switched on) and in *from its value once everything launched so far has ended. */
int nla_local_ctx_count_finished(nla_local_ctx *c)
{
    if (!c || c->ev.kind != NLA_EVAL_DEVICE) return -1;
    if (c->d_done) return 0;
    c->d_done = (int32_t *) nla_dev_malloc_uncached(sizeof(int32_t));
    if (!c->d_done) return -1;
    if (nla_memset(c->d_done, 0, sizeof(int32_t), c->st) || nla_stream_sync(c->st)) { nla_dev_free_uncached(c->d_done); c->d_done = NULL; return -1; }
    c->done_expected = 0;
    return 0;
}
const int32_t *nla_local_ctx_finished_counter(nla_local_ctx *c, int32_t *from)
{
    if (!c || !c->d_done) return NULL;
    *from = c->done_expected;
    return c->d_done;
}
int nla_local_ctx_alg(const nla_local_ctx *c) { return c->alg; }
double *nla_local_ctx_X(nla_local_ctx *c) { return c->d_X; }

/* exact-order sums ("amd_exact_dot"), per-coordinate x tolerances and weights (host arrays of n or NULL; stop.c:98-120) */
int nla_local_ctx_set_options(nla_local_ctx *c, int exact, const double *xtol_abs, const double *x_weights)
{
    c->exact = exact;
    nla_dev_free(c->d_xtol_abs); nla_dev_free(c->d_x_weights);
    c->d_xtol_abs = c->d_x_weights = NULL;
    free(c->h_xtol_abs); c->h_xtol_abs = NULL;
    if (xtol_abs && c->alg == 2) {
        c->h_xtol_abs = (double *) malloc(sizeof(double) * (size_t) c->n);
        if (!c->h_xtol_abs) return -1;
        memcpy(c->h_xtol_abs, xtol_abs, sizeof(double) * (size_t) c->n);
    }
    if (xtol_abs) {
        c->d_xtol_abs = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld);
        if (!c->d_xtol_abs || nla_memcpy_h2d(c->d_xtol_abs, xtol_abs, sizeof(double) * (size_t) c->n, c->st)) return -1;
    }
    if (x_weights) {
        c->d_x_weights = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld);
        if (!c->d_x_weights || nla_memcpy_h2d(c->d_x_weights, x_weights, sizeof(double) * (size_t) c->n, c->st)) return -1;
    }
    return nla_stream_sync(c->st);
}

/* record f of every evaluation of each search (cap per search); read back with nla_local_ctx_read_ftrace */
int nla_local_ctx_set_ftrace(nla_local_ctx *c, int64_t cap)
{
    nla_dev_free(c->d_ftrace);
    c->d_ftrace = NULL; c->ftrace_cap = 0;
    if (cap <= 0) return 0;
    c->d_ftrace = (double *) nla_dev_malloc(sizeof(double) * (size_t) cap * (size_t) c->cap);
    if (!c->d_ftrace) return -1;
    c->ftrace_cap = cap;
    return 0;
}
int nla_local_ctx_read_ftrace(nla_local_ctx *c, int inst, int64_t count, double *h_out)
{
    if (!c->d_ftrace || count > c->ftrace_cap) return -1;
    if (nla_memcpy_d2h(h_out, c->d_ftrace + (size_t) inst * (size_t) c->ftrace_cap, sizeof(double) * (size_t) count, c->st)) return -1;
    return nla_stream_sync(c->st);
}

static int launch(nla_local_ctx *c, int count, const nla_lbfgs_params *prm, const nla_local_ext *ext)
{
    const int obj = c->ev.kind == NLA_EVAL_DEVICE ? c->ev.obj : NLA_OBJ_EXTERNAL;
    if (c->alg == 2) {
        nla_cobyla_params P;
        memset(&P, 0, sizeof P);
        P.minf_max = prm->minf_max; P.ftol_rel = prm->ftol_rel; P.ftol_abs = prm->ftol_abs; P.xtol_rel = prm->xtol_rel; P.maxeval = prm->maxeval;
        P.exact = (c->exact & 1); P.sign = c->ev.sign; P.xtol_abs = c->d_xtol_abs; P.abort = c->h_abort; P.done = c->d_done;
        if (ext) return -1;
        return nla_k_cobyla_batch(obj, c->n, c->ld, count, c->d_lb, c->d_ub, c->d_sigma_init, c->d_X, c->d_work, c->d_iwork, &P, c->d_res, c->st);
    } else if (c->alg == 1) {
        nla_mma_params P = c->mma;
        P.minf_max = prm->minf_max; P.ftol_rel = prm->ftol_rel; P.ftol_abs = prm->ftol_abs; P.xtol_rel = prm->xtol_rel; P.maxeval = prm->maxeval;
        P.exact = (c->exact & 1); P.sign = c->ev.sign; P.xtol_abs = c->d_xtol_abs; P.x_weights = c->d_x_weights; P.abort = c->h_abort;
        P.ftrace = c->d_ftrace; P.ftrace_cap = c->ftrace_cap; P.done = ext ? NULL : c->d_done;
        return nla_k_mma_batch(obj, c->n, c->ld, count, c->d_lb, c->d_ub, c->d_sigma_init, c->d_X, c->d_work, &P, c->d_res, ext, c->st);
    } else {
        nla_lbfgs_params P = *prm;
        P.exact = c->exact; P.sign = c->ev.sign; P.xtol_abs = c->d_xtol_abs; P.x_weights = c->d_x_weights; P.abort = c->h_abort;
        P.ftrace = c->d_ftrace; P.ftrace_cap = c->ftrace_cap; P.done = ext ? NULL : c->d_done;
        return nla_k_lbfgs_batch(obj, c->n, c->ld, c->mf, count, c->d_lb, c->d_ub, c->d_X, c->d_work, c->d_iwork, c->d_hist, &P, c->d_res, ext, c->st);
    }
}

/* LN_COBYLA, a batch of fewer searches than the device needs to beat a host core (cob_min_batch): the rows come to the host, each search
 * runs through the library's own nlopt_optimize(LN_COBYLA) (cobyla_host.c: the reference's run evaluation by evaluation) on the
 * objective's host twin, the results go back to where the kernel would have left them.  In exact-order mode the twin's values ARE the
 * kernel's (same summation order, IEEE arithmetic), so where a batch runs changes nothing; in the default mode they differ by the
 * rounding of the tree sum. */
typedef struct { int obj; double sign; const nla_stopping *stop; nlopt_opt loc; int timed; } cob_host_obj;
static double cob_host_f(unsigned n, const double *x, double *g, void *p)
{
    cob_host_obj *o = (cob_host_obj *) p;
    (void) g;
    if (o->stop) {
        if (nla_stop_forced(o->stop)) nlopt_force_stop(o->loc);
        else if (nla_stop_time(o->stop)) { o->timed = 1; nlopt_force_stop(o->loc); }
    }
    return o->sign * nla_obj_eval_seq(o->obj, n, x, NULL);
}
static int cobyla_small_batch_on_host(nla_local_ctx *c, int count, const nla_lbfgs_params *prm, nla_lbfgs_result *h_res, const nla_stopping *stop)
{
    const size_t rows = sizeof(double) * (size_t) c->ld * (size_t) count;
    int rc, i;
    if ((rc = nla_memcpy_d2h(c->h_rows, c->d_X, rows, c->st)) || (rc = nla_stream_sync(c->st))) return rc;
    for (i = 0; i < count; ++i) {
        cob_host_obj o = { c->ev.obj & 0xff, (c->ev.sign == 0. ? 1. : c->ev.sign) * ((c->ev.obj & 0x100) ? -1. : 1.), stop, NULL, 0 };
        nlopt_opt loc = nlopt_create(NLOPT_LN_COBYLA, (unsigned) c->n);
        double minf = HUGE_VAL;
        if (!loc) return -1;
        o.loc = loc;
        nlopt_set_min_objective(loc, cob_host_f, &o);
        nlopt_set_lower_bounds(loc, c->h_lb); nlopt_set_upper_bounds(loc, c->h_ub);
        nlopt_set_stopval(loc, prm->minf_max); nlopt_set_ftol_rel(loc, prm->ftol_rel); nlopt_set_ftol_abs(loc, prm->ftol_abs); nlopt_set_xtol_rel(loc, prm->xtol_rel);
        if (c->h_xtol_abs) nlopt_set_xtol_abs(loc, c->h_xtol_abs);
        nlopt_set_maxeval(loc, prm->maxeval);
        if (c->h_dx) nlopt_set_initial_step(loc, c->h_dx);
        h_res[i].ret = nlopt_optimize(loc, c->h_rows + (size_t) i * (size_t) c->ld, &minf);
        if (o.timed && h_res[i].ret == NLOPT_FORCED_STOP) h_res[i].ret = NLOPT_MAXTIME_REACHED;
        h_res[i].f = minf; h_res[i].nevals = h_res[i].iterm = nlopt_get_numevals(loc); h_res[i].cols = 0;
        nlopt_destroy(loc);
    }
    if ((rc = nla_memcpy_h2d(c->d_X, c->h_rows, rows, c->st))) return rc;
    if (c->after_launch) { int (*fn)(void *) = c->after_launch; c->after_launch = NULL; if ((rc = fn(c->after_arg))) return rc; }
    if ((rc = nla_stream_sync(c->st))) return rc;
    if (c->stats) c->stats->cobyla_host_searches += (uint64_t) count;
    return 0;
}

/* wait for the stream; meanwhile forward the caller's force_stop flag and maxtime to the kernel's abort flag */
static int wait_watching(nla_local_ctx *c, const nla_stopping *stop)
{
    int rc;
    if (!stop || (!stop->force_stop && stop->maxtime <= 0)) return nla_stream_sync(c->st);
    while ((rc = nla_stream_query(c->st)) == -1) {
        if (nla_stop_forced(stop)) *(volatile int32_t *) c->h_abort = -999;
        else if (nla_stop_time(stop)) *(volatile int32_t *) c->h_abort = 100;
        sched_yield();
    }
    return rc;
}

/* run `count` searches from the rows already in ctx X; minimisers stay there, results come to the host.  stop (may be
 * NULL): the caller's force_stop flag and time limit, observed during the searches; live_nevals (may be NULL; host
 * objective only): incremented at every counted evaluation as the reference does (plis.c:261,391; mma.c:219,298). */
int nla_local_ctx_run(nla_local_ctx *c, int count, const nla_lbfgs_params *prm, nla_lbfgs_result *h_res, const nla_stopping *stop,
                      int *live_nevals)
{
    int rc, i;
    if (count > c->cap) return -1;
    if (c->alg == 2 && count < c->cob_min_batch) return cobyla_small_batch_on_host(c, count, prm, h_res, stop);
    *c->h_abort = 0;
    nla_event_record(c->ev0, c->st);
    if (c->ev.kind == NLA_EVAL_DEVICE) {
        if ((rc = launch(c, count, prm, NULL))) return rc;
        if (c->d_done) c->done_expected = (int32_t) ((uint32_t) c->done_expected + (uint32_t) count);
        nla_event_record(c->ev1, c->st);
        if (c->after_launch) { int (*fn)(void *) = c->after_launch; c->after_launch = NULL; if ((rc = fn(c->after_arg))) return rc; }
        /* watch first, copy afterwards: a device-to-host copy into pageable memory (the caller's result record may live on its
         * stack) does not return before the kernel has finished, and nobody would raise the abort flag meanwhile */
        if ((rc = wait_watching(c, stop))) return rc;
        if ((rc = nla_memcpy_d2h(h_res, c->d_res, sizeof(nla_lbfgs_result) * (size_t) count, c->st))) return rc;
        if ((rc = nla_stream_sync(c->st))) return rc;
    } else {
        nla_local_ext E = c->ext;
        const size_t rowb = sizeof(double) * (size_t) c->n;
        if ((rc = nla_memset(E.req, 0, sizeof(nla_local_req) * (size_t) count, c->st))) return rc;
        E.resume = 0;
        for (;;) {
            int waiting = 0;
            E.forced = stop ? nla_stop_forced(stop) : 0;
            E.timeout = stop ? nla_stop_time(stop) : 0;
            if ((rc = launch(c, count, prm, &E))) return rc;
            if ((rc = nla_memcpy_d2h(c->h_req, E.req, sizeof(nla_local_req) * (size_t) count, c->st))) return rc;
            if ((rc = nla_stream_sync(c->st))) return rc;
            for (i = 0; i < count; ++i) waiting += c->h_req[i].state == 1;
            if (!waiting) break;
            if (c->ev.kind == NLA_EVAL_USER) {
                /* every waiting search is evaluated by one launch of the user's kernel; list entry i: search i wants its
                 * gradient, -(i+1): the value only */
                int m = 0;
                for (i = 0; i < count; ++i) if (c->h_req[i].state == 1) c->h_list[m++] = (c->h_req[i].want_grad & 1) ? i : -(i + 1);
                if ((rc = nla_memcpy_h2d(c->d_list, c->h_list, sizeof(int32_t) * (size_t) m, c->st))) return rc;
                if ((rc = nla_userobj_evalgrad_list(c->ev.user, c->n, c->ld, m, c->d_list, E.EX, E.EF, E.EG, c->ev.sign, c->st))) return rc;
            } else {
                for (i = 0; i < count; ++i) {
                    const int want = c->h_req[i].want_grad;          /* bit 0: gradient wanted; bit 1: LD_MMA's uncounted call (mma.c:337-339) */
                    if (c->h_req[i].state != 1) continue;
                    if ((rc = nla_memcpy_d2h(c->h_x, E.EX + (size_t) i * c->ld, rowb, c->st)) || (rc = nla_stream_sync(c->st))) return rc;
                    c->h_f[i] = c->ev.f((unsigned) c->n, c->h_x, (want & 1) ? c->h_g : NULL, c->ev.f_data);
                    if (live_nevals && !(want & 2)) ++*live_nevals;
                    if ((rc = nla_memcpy_h2d(E.EF + i, c->h_f + i, sizeof(double), c->st))) return rc;
                    if ((want & 1) && (rc = nla_memcpy_h2d(E.EG + (size_t) i * c->ld, c->h_g, rowb, c->st))) return rc;
                    if ((rc = nla_stream_sync(c->st))) return rc;         /* h_g is reused by the next evaluation */
                }
            }
            E.resume = 1;
        }
        nla_event_record(c->ev1, c->st);
        if (c->after_launch) { int (*fn)(void *) = c->after_launch; c->after_launch = NULL; if ((rc = fn(c->after_arg))) return rc; }
        if ((rc = nla_memcpy_d2h(h_res, c->d_res, sizeof(nla_lbfgs_result) * (size_t) count, c->st))) return rc;
        if ((rc = nla_stream_sync(c->st))) return rc;
    }
    if (c->stats) {
        ++c->stats->lbfgs_launches;
        c->stats->t_lbfgs_ms += (double) nla_event_elapsed_ms(c->ev0, c->ev1);
        for (i = 0; i < count; ++i)
            c->stats->lbfgs_bytes += c->alg == 2 ? (uint64_t) c->n * 16ULL * (uint64_t) h_res[i].nevals    /* (the point written and read once per evaluation) */
                                   : c->alg == 1 ? (uint64_t) c->n * 64ULL * (uint64_t) h_res[i].nevals    /* mma_kernels.hip header */
                                                 : (uint64_t) c->n * (32ULL * (uint64_t) h_res[i].cols + 16ULL * (uint64_t) h_res[i].nevals);
        if (c->alg == 0) {
            uint64_t longest = 0;
            for (i = 0; i < count; ++i) {
                const uint64_t steps = (uint64_t) c->n * (2ULL * (uint64_t) h_res[i].cols + 4ULL * (uint64_t) h_res[i].nevals);
                if (steps > longest) longest = steps;
            }
            c->stats->lbfgs_longest_chain_steps += longest;
        }
    }
    return 0;
}

/* `count` local searches from the rows of h_X (count x n, host), results back in h_X / res.
 * alg 0: LD_LBFGS with mf history pairs; alg 1: LD_MMA with its parameters and optional initial step (host, n) */
int nla_local_run_batch(int alg, const nla_evaluator *ev, int n, int count, const double *lb, const double *ub, double *h_X, int mf,
                        const nla_mma_params *mma, const double *sigma_init, const nla_lbfgs_params *prm, nla_lbfgs_result *res,
                        const nla_stopping *stop, int exact, int *live_nevals, nlopt_opt trace_to, char *err, size_t errlen)
{
    const int ld = (n + 1) & ~1;
    void *st = nla_stream_create();
    double *d_lb = NULL, *d_ub = NULL, *d_si = NULL;
    nla_local_ctx *c = NULL;
    int rc = -1, i;
    if (!st) { snprintf(err, errlen, "stream creation failed"); return -1; }
    d_lb = (double *) nla_dev_malloc(sizeof(double) * (size_t) ld);
    d_ub = (double *) nla_dev_malloc(sizeof(double) * (size_t) ld);
    if (alg == 1 && sigma_init) {
        d_si = (double *) nla_dev_malloc(sizeof(double) * (size_t) ld);
        if (!d_si || nla_memcpy_h2d(d_si, sigma_init, sizeof(double) * (size_t) n, st)) { snprintf(err, errlen, "upload failed"); goto done; }
    }
    if (d_lb && d_ub) c = alg == 1 ? nla_local_ctx_create_mma(ev, n, count, mma, d_si, d_lb, d_ub, st) : nla_local_ctx_create(ev, n, count, mf, d_lb, d_ub, st);
    if (!c) { snprintf(err, errlen, "out of device memory"); goto done; }
    if (nla_memcpy_h2d(d_lb, lb, sizeof(double) * (size_t) n, st) || nla_memcpy_h2d(d_ub, ub, sizeof(double) * (size_t) n, st)) { snprintf(err, errlen, "upload failed"); goto done; }
    if (nla_local_ctx_set_options(c, exact, stop ? stop->xtol_abs : NULL, stop ? stop->x_weights : NULL)) { snprintf(err, errlen, "upload failed"); goto done; }
    for (i = 0; i < count; ++i)
        if (nla_memcpy_h2d(c->d_X + (size_t) i * ld, h_X + (size_t) i * n, sizeof(double) * (size_t) n, st)) { snprintf(err, errlen, "upload failed"); goto done; }
    if (trace_to && trace_to->trace && count == 1 && nla_local_ctx_set_ftrace(c, (int64_t) trace_to->trace_cap)) { snprintf(err, errlen, "out of device memory"); goto done; }
    if ((i = nla_local_ctx_run(c, count, prm, res, stop, live_nevals))) { snprintf(err, errlen, "local-search batch failed: %s", nla_dev_error_string(i)); goto done; }
    if (trace_to && trace_to->trace && count == 1) {
        /* per-evaluation trace (nlopt_amd_set_trace): kind 5 = an objective call of a local search */
        const int64_t calls = alg == 1 ? res[0].iterm : res[0].nevals;
        const int64_t k = calls < (int64_t) trace_to->trace_cap ? calls : (int64_t) trace_to->trace_cap;
        double *tmp = (double *) malloc(sizeof(double) * (size_t) (k > 0 ? k : 1));
        if (!tmp || (k > 0 && nla_local_ctx_read_ftrace(c, 0, k, tmp))) { free(tmp); snprintf(err, errlen, "trace read-back failed"); goto done; }
        for (i = 0; i < (int) k; ++i) { nlopt_amd_trace_rec *tr = trace_to->trace + i; tr->f = tmp[i]; tr->row = i; tr->kind = 5; tr->accepted = 0; }
        trace_to->trace_len = (size_t) calls;
        free(tmp);
    }
    for (i = 0; i < count; ++i)
        if (nla_memcpy_d2h(h_X + (size_t) i * n, c->d_X + (size_t) i * ld, sizeof(double) * (size_t) n, st)) { snprintf(err, errlen, "read-back failed"); goto done; }
    if ((i = nla_stream_sync(st))) { snprintf(err, errlen, "read-back failed: %s", nla_dev_error_string(i)); goto done; }
    rc = 0;
done:
    nla_local_ctx_destroy(c);
    nla_dev_free(d_lb); nla_dev_free(d_ub); nla_dev_free(d_si);
    nla_stream_destroy(st);
    return rc;
}

/* reference-shaped entry: luksan_plis(n, f, f_data, lb, ub, x, minf, stop, mf, tolg) (plis.c:420-426) */
nlopt_result nla_lbfgs_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                                nla_stopping *stop, int mf, double tolg)
{
    nla_evaluator ev;
    nla_lbfgs_params prm;
    nla_lbfgs_result res;
    char err[200];
    if (nla_dev_count() <= 0) { nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
    nla_evaluator_resolve(&ev, opt, f, f_data);
    memset(&prm, 0, sizeof prm);
    prm.minf_max = stop->minf_max; prm.ftol_rel = stop->ftol_rel; prm.ftol_abs = stop->ftol_abs; prm.xtol_rel = stop->xtol_rel;
    prm.tolg = tolg; prm.maxeval = stop->maxeval;
    mf = nla_lbfgs_default_mf(n, mf, stop->maxeval);
    if (nla_local_run_batch(0, &ev, n, 1, lb, ub, x, mf, NULL, NULL, &prm, &res, stop, nla_exact_mode_for(opt, NULL, &ev),
                            ev.kind == NLA_EVAL_HOST ? stop->nevals_p : NULL, opt, err, sizeof err)) { nla_stop_msg(stop, "device engine: %s", err); return NLOPT_FAILURE; }
    *minf = res.f;
    if (ev.kind != NLA_EVAL_HOST) *stop->nevals_p += res.nevals;       /* host objective: counted call by call */
    return (nlopt_result) res.ret;
}
