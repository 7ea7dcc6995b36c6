/* lbfgs_driver.c — NLOPT_LD_LBFGS behind the reference's entry point luksan_plis (plis.c:420-510):
 * argument handling on the host, the optimisation itself in one launch of the batched device
 * kernel (hip/lbfgs_kernels.hip) — used directly by nlopt_optimize(LD_LBFGS) with count = 1 and by
 * MLSL (mlsl_driver.c) with one workgroup per start point.  The batch context (nla_local_ctx) also
 * carries the second local optimiser, LD_MMA (mma_driver.c, hip/mma_kernels.hip).
 *
 * Provided for device objectives (nlopt_amd_objective): the objective and its gradient are
 * evaluated inside the kernel.  A host callback would need a PCIe round trip per evaluation; that
 * path is not provided and says so. */
#include "nla_internal.h"
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

/* a reusable batch: device buffers for up to `cap` simultaneous local searches by LD_LBFGS (alg 0) or LD_MMA (alg 1) */
struct nla_local_ctx {
    int alg, obj, n, ld, cap, mf;
    nla_mma_params mma;            /* alg 1: the algorithm's own parameters (the stopping values come with each run) */
    const double *d_sigma_init;    /* alg 1: initial step on the device, or NULL */
    void *st;
    const double *d_lb, *d_ub;
    double *d_X, *d_work, *d_hist;
    int *d_iwork;
    nla_lbfgs_result *d_res;
    void *ev0, *ev1;
    nlopt_amd_stats *stats;        /* optional: device time / algorithmic bytes of the launches are added here */
};

void nla_local_ctx_destroy(nla_local_ctx *c)
{
    if (!c) return;
    nla_dev_free(c->d_X); nla_dev_free(c->d_work); nla_dev_free(c->d_iwork); nla_dev_free(c->d_hist); nla_dev_free(c->d_res);
    nla_event_destroy(c->ev0); nla_event_destroy(c->ev1);
    free(c);
}

nla_local_ctx *nla_local_ctx_create(int obj, int n, int cap, int mf, const double *d_lb, const double *d_ub, void *stream)
{
    nla_local_ctx *c = (nla_local_ctx *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->obj = obj; c->n = n; c->ld = (n + 1) & ~1; c->cap = cap; c->mf = mf; c->st = stream; c->d_lb = d_lb; c->d_ub = d_ub;
    c->d_X = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld * (size_t) cap);
    c->d_work = (double *) nla_dev_malloc(sizeof(double) * nla_lbfgs_work_doubles(c->ld, mf, cap));
    c->d_iwork = (int *) nla_dev_malloc(sizeof(int) * (size_t) c->ld * (size_t) cap);
    c->d_hist = (double *) nla_dev_malloc(sizeof(double) * nla_lbfgs_hist_doubles(c->ld, mf, cap));
    c->d_res = (nla_lbfgs_result *) nla_dev_malloc(sizeof(nla_lbfgs_result) * (size_t) cap);
    c->ev0 = nla_event_create(); c->ev1 = nla_event_create();
    if (!c->d_X || !c->d_work || !c->d_iwork || !c->d_hist || !c->d_res || !c->ev0 || !c->ev1) { nla_local_ctx_destroy(c); return NULL; }
    return c;
}
/* the same for LD_MMA (mma_driver.c reads the parameters; sigma_init: device copy of the initial step or NULL) */
nla_local_ctx *nla_local_ctx_create_mma(int obj, int n, int cap, const nla_mma_params *alg_params, const double *d_sigma_init,
                                        const double *d_lb, const double *d_ub, void *stream)
{
    nla_local_ctx *c = (nla_local_ctx *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->alg = 1; c->obj = obj; c->n = n; c->ld = (n + 1) & ~1; c->cap = cap; c->st = stream; c->d_lb = d_lb; c->d_ub = d_ub;
    c->mma = *alg_params; c->d_sigma_init = d_sigma_init;
    c->d_X = (double *) nla_dev_malloc(sizeof(double) * (size_t) c->ld * (size_t) cap);
    c->d_work = (double *) nla_dev_malloc(sizeof(double) * nla_mma_work_doubles(c->ld, cap));
    c->d_res = (nla_lbfgs_result *) nla_dev_malloc(sizeof(nla_lbfgs_result) * (size_t) cap);
    c->ev0 = nla_event_create(); c->ev1 = nla_event_create();
    if (!c->d_X || !c->d_work || !c->d_res || !c->ev0 || !c->ev1) { nla_local_ctx_destroy(c); return NULL; }
    return c;
}
void nla_local_ctx_set_stats(nla_local_ctx *c, nlopt_amd_stats *stats) { if (c) c->stats = stats; }
int nla_local_ctx_alg(const nla_local_ctx *c) { return c->alg; }

double *nla_local_ctx_X(nla_local_ctx *c) { return c->d_X; }

/* run `count` searches from the rows already in ctx X; minimisers stay there, results come to the host */
int nla_local_ctx_run(nla_local_ctx *c, int count, const nla_lbfgs_params *prm, nla_lbfgs_result *h_res)
{
    int rc, i;
    if (count > c->cap) return -1;
    nla_event_record(c->ev0, c->st);
    if (c->alg == 1) {
        nla_mma_params P = c->mma;
        P.minf_max = prm->minf_max; P.ftol_rel = prm->ftol_rel; P.ftol_abs = prm->ftol_abs; P.xtol_rel = prm->xtol_rel; P.maxeval = prm->maxeval;
        if ((rc = nla_k_mma_batch(c->obj, c->n, c->ld, count, c->d_lb, c->d_ub, c->d_sigma_init, c->d_X, c->d_work, &P, c->d_res, c->st))) return rc;
    } else if ((rc = nla_k_lbfgs_batch(c->obj, c->n, c->ld, c->mf, count, c->d_lb, c->d_ub, c->d_X, c->d_work, c->d_iwork, c->d_hist, prm, c->d_res, c->st))) return rc;
    nla_event_record(c->ev1, c->st);
    if ((rc = nla_memcpy_d2h(h_res, c->d_res, sizeof(nla_lbfgs_result) * (size_t) count, c->st))) return rc;
    if ((rc = nla_stream_sync(c->st))) return rc;
    if (c->stats) {
        ++c->stats->lbfgs_launches;
        c->stats->t_lbfgs_ms += (double) nla_event_elapsed_ms(c->ev0, c->ev1);
        for (i = 0; i < count; ++i)
            c->stats->lbfgs_bytes += c->alg == 1 ? (uint64_t) c->n * 64ULL * (uint64_t) h_res[i].nevals    /* mma_kernels.hip header */
                                                 : (uint64_t) c->n * (32ULL * (uint64_t) h_res[i].cols + 16ULL * (uint64_t) h_res[i].nevals);
    }
    return 0;
}

/* `count` local searches from the rows of h_X (count x n, host), results back in h_X / res */
int nla_lbfgs_run_batch(int obj, int n, int count, const double *lb, const double *ub, double *h_X, int mf,
                        const nla_lbfgs_params *prm, nla_lbfgs_result *res, char *err, size_t errlen)
{
    return nla_local_run_batch(0, obj, n, count, lb, ub, h_X, mf, NULL, NULL, prm, res, err, errlen);
}

/* alg 0: LD_LBFGS with mf history pairs; alg 1: LD_MMA with its parameters and optional initial step (host, n) */
int nla_local_run_batch(int alg, int obj, int n, int count, const double *lb, const double *ub, double *h_X, int mf,
                        const nla_mma_params *mma, const double *sigma_init, const nla_lbfgs_params *prm, nla_lbfgs_result *res,
                        char *err, size_t errlen)
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
    if (d_lb && d_ub) c = alg == 1 ? nla_local_ctx_create_mma(obj, n, count, mma, d_si, d_lb, d_ub, st) : nla_local_ctx_create(obj, n, count, mf, d_lb, d_ub, st);
    if (!c) { snprintf(err, errlen, "out of device memory"); goto done; }
    if (nla_memcpy_h2d(d_lb, lb, sizeof(double) * (size_t) n, st) || nla_memcpy_h2d(d_ub, ub, sizeof(double) * (size_t) n, st)) { snprintf(err, errlen, "upload failed"); goto done; }
    for (i = 0; i < count; ++i)
        if (nla_memcpy_h2d(c->d_X + (size_t) i * ld, h_X + (size_t) i * n, sizeof(double) * (size_t) n, st)) { snprintf(err, errlen, "upload failed"); goto done; }
    if ((i = nla_local_ctx_run(c, count, prm, res))) { snprintf(err, errlen, "local-search batch failed: %s", nla_dev_error_string(i)); goto done; }
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
nlopt_result nla_lbfgs_minimize(int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                                nla_stopping *stop, int mf, double tolg)
{
    const int obj = nlopt_amd_objective_id(f);
    nla_lbfgs_params prm;
    nla_lbfgs_result res;
    char err[200];
    (void) f_data;
    if (nla_dev_count() <= 0) { nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
    if (obj < 0) {
        nla_stop_msg(stop, "nlopt_amd: LD_LBFGS is provided for device objectives (nlopt_amd_objective) only");
        return NLOPT_INVALID_ARGS;
    }
    if (stop->xtol_abs) { nla_stop_msg(stop, "nlopt_amd: LD_LBFGS on the device does not take xtol_abs"); return NLOPT_INVALID_ARGS; }
    memset(&prm, 0, sizeof prm);
    prm.minf_max = stop->minf_max; prm.ftol_rel = stop->ftol_rel; prm.ftol_abs = stop->ftol_abs; prm.xtol_rel = stop->xtol_rel;
    prm.tolg = tolg; prm.maxeval = stop->maxeval;
    mf = nla_lbfgs_default_mf(n, mf, stop->maxeval);
    if (nla_lbfgs_run_batch(obj, n, 1, lb, ub, x, mf, &prm, &res, err, sizeof err)) { nla_stop_msg(stop, "device engine: %s", err); return NLOPT_FAILURE; }
    *minf = res.f;
    *stop->nevals_p += res.nevals;
    return (nlopt_result) res.ret;
}
