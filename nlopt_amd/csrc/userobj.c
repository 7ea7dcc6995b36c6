/* userobj.c — user-supplied device objectives (include/nlopt_amd.h part 1, include/nlopt_amd_device.h).
 *
 * The reference has no counterpart: its objective is a host callback (src/api/nlopt.h:60-62).  SURVEY.md §8b names "an
 * additive setter" as the extension a GPU implementation needs; this is it.  A bound objective is a loaded code object
 * and its <name>_evalgrad kernel; the algorithms reach it through nla_evaluator (kind NLA_EVAL_USER):
 *   populations / samples   nla_userobj_eval_rows     one launch over all rows, one wavefront per row
 *   local searches          nla_userobj_evalgrad_list  one launch per step over the searches that wait (lbfgs_driver.c)
 *   anything else           the adapter callback below: a single point through the same kernel (exact, slow) */
#include "nla_internal.h"
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct nla_userobj {
    int refs;
    void *module, *fn;
    nlopt_func twin; void *twin_data;
    /* single-point evaluation through the kernel (adapter): one scratch set per object — and the object is shared by nlopt_copy
     * (per-thread copies are the normal NLopt pattern), so the adapter serialises on `lock` */
    pthread_mutex_t lock;
    void *st;
    double *d_x, *d_g, *d_f, *h_buf;       /* h_buf pinned: x | g | f */
    int cap;
};

nla_userobj *nla_userobj_retain(nla_userobj *u) { if (u) __atomic_add_fetch(&u->refs, 1, __ATOMIC_RELAXED); return u; }
void nla_userobj_release(nla_userobj *u)
{
    if (!u || __atomic_sub_fetch(&u->refs, 1, __ATOMIC_ACQ_REL) > 0) return;
    if (u->st) nla_stream_sync(u->st);
    nla_dev_free(u->d_x); nla_dev_free(u->d_g); nla_dev_free(u->d_f); nla_host_free(u->h_buf);
    if (u->st) nla_stream_destroy(u->st);
    nla_module_unload(u->module);
    pthread_mutex_destroy(&u->lock);
    free(u);
}

static int launch(nla_userobj *u, int n, int ld, int64_t count, const int32_t *list, const double *X, double *F, double *G, double sign, void *stream)
{
    /* <name>_evalgrad(int n, int ld, long count, const int *list, const double *X, double *F, double *G, double sign) */
    int32_t an = n, ald = ld;
    int64_t acount = count;
    double asign = sign == 0. ? 1. : sign;
    void *params[8];
    if (count <= 0) return 0;
    params[0] = &an; params[1] = &ald; params[2] = &acount; params[3] = &list; params[4] = &X; params[5] = &F; params[6] = &G; params[7] = &asign;
    return nla_module_launch(u->fn, (unsigned) ((count + 3) / 4), 256, params, stream);
}

int nla_userobj_eval_rows(nla_userobj *u, int n, int ld, int64_t count, const double *X, double *F, double *G, double sign, void *stream)
{
    return launch(u, n, ld, count, NULL, X, F, G, sign, stream);
}
int nla_userobj_evalgrad_list(nla_userobj *u, int n, int ld, int m, const int32_t *d_list, const double *X, double *F, double *G,
                              double sign, void *stream)
{
    return launch(u, n, ld, m, d_list, X, F, G, sign, stream);
}

/* the objective as an ordinary nlopt_func: the host twin if the user gave one, else one point through the kernel */
static double adapter(unsigned n, const double *x, double *grad, void *data)
{
    nla_userobj *u = (nla_userobj *) data;
    const int ld = (int) ((n + 1) & ~1u);
    double fv = HUGE_VAL;
    if (u->twin) return u->twin(n, x, grad, u->twin_data);
    pthread_mutex_lock(&u->lock);
    if ((int) n > u->cap) {
        nla_dev_free(u->d_x); nla_dev_free(u->d_g); nla_host_free(u->h_buf);
        u->d_x = (double *) nla_dev_malloc(sizeof(double) * (size_t) ld);
        u->d_g = (double *) nla_dev_malloc(sizeof(double) * (size_t) ld);
        u->h_buf = (double *) nla_host_malloc(sizeof(double) * (2 * (size_t) ld + 1));
        u->cap = (u->d_x && u->d_g && u->h_buf) ? (int) n : 0;
    }
    if (u->cap >= (int) n) {
        memcpy(u->h_buf, x, sizeof(double) * n);
        if (!(nla_memcpy_h2d(u->d_x, u->h_buf, sizeof(double) * n, u->st) ||
              launch(u, (int) n, ld, 1, NULL, u->d_x, u->d_f, grad ? u->d_g : NULL, 1., u->st) ||
              nla_memcpy_d2h(u->h_buf + 2 * (size_t) ld, u->d_f, sizeof(double), u->st) ||
              (grad && nla_memcpy_d2h(u->h_buf + ld, u->d_g, sizeof(double) * n, u->st)) || nla_stream_sync(u->st))) {
            if (grad) memcpy(grad, u->h_buf + ld, sizeof(double) * n);
            fv = u->h_buf[2 * (size_t) ld];
        }
    }
    pthread_mutex_unlock(&u->lock);
    return fv;
}
int nla_userobj_is_adapter(nlopt_func f) { return f == adapter; }

static nlopt_result bind(nlopt_opt opt, const char *code_object, const char *name, nlopt_func twin, void *f_data, int maximize)
{
    nla_userobj *u;
    char sym[256];
    nlopt_result r;
    if (!opt) return NLOPT_INVALID_ARGS;
    nla_unset_errmsg(opt);
    if (!code_object || !name || strlen(name) > 200) { nla_set_errmsg(opt, "nlopt_amd: device objective needs a code object path and a name"); return NLOPT_INVALID_ARGS; }
    if (nla_dev_count() <= 0) { nla_set_errmsg(opt, "nlopt_amd: no HIP device visible (this library has no CPU fallback)"); return NLOPT_FAILURE; }
    u = (nla_userobj *) calloc(1, sizeof *u);
    if (!u) return NLOPT_OUT_OF_MEMORY;
    u->refs = 1; u->twin = twin; u->twin_data = f_data;
    pthread_mutex_init(&u->lock, NULL);
    if (!(u->module = nla_module_load_file(code_object))) {
        nla_set_errmsg(opt, "nlopt_amd: could not load code object %s (build it for gfx950 with hipcc --genco)", code_object);
        nla_userobj_release(u); return NLOPT_INVALID_ARGS;
    }
    snprintf(sym, sizeof sym, "%s_evalgrad", name);
    if (!(u->fn = nla_module_function(u->module, sym))) {
        nla_set_errmsg(opt, "nlopt_amd: kernel %s not found in %s (NLOPT_AMD_DEVICE_OBJECTIVE(%s, ...) of nlopt_amd_device.h defines it)", sym, code_object, name);
        nla_userobj_release(u); return NLOPT_INVALID_ARGS;
    }
    u->st = nla_stream_create();
    u->d_f = (double *) nla_dev_malloc(sizeof(double));
    if (!u->st || !u->d_f) { nla_userobj_release(u); return NLOPT_OUT_OF_MEMORY; }
    {   /* ABI check: <name>_abi writes the header's version */
        void *abi;
        int32_t *d_v = (int32_t *) nla_dev_malloc(sizeof(int32_t)), v = -1;
        void *params[1];
        snprintf(sym, sizeof sym, "%s_abi", name);
        abi = nla_module_function(u->module, sym);
        params[0] = &d_v;
        if (!abi || !d_v || nla_module_launch(abi, 1, 1, params, u->st) || nla_memcpy_d2h(&v, d_v, sizeof v, u->st) || nla_stream_sync(u->st) || v != 1) {
            nla_dev_free(d_v);
            nla_set_errmsg(opt, "nlopt_amd: %s in %s was not built with this library's nlopt_amd_device.h (ABI %d, expected 1)", name, code_object, (int) v);
            nla_userobj_release(u); return NLOPT_INVALID_ARGS;
        }
        nla_dev_free(d_v);
    }
    r = maximize ? nlopt_set_max_objective(opt, adapter, u) : nlopt_set_min_objective(opt, adapter, u);   /* releases a previous binding */
    if (r != NLOPT_SUCCESS) { nla_userobj_release(u); return r; }
    opt->userobj = u;
    return NLOPT_SUCCESS;
}

nlopt_result nlopt_amd_set_min_device_objective(nlopt_opt opt, const char *code_object, const char *name, nlopt_func host_twin, void *f_data)
{ return bind(opt, code_object, name, host_twin, f_data, 0); }
nlopt_result nlopt_amd_set_max_device_objective(nlopt_opt opt, const char *code_object, const char *name, nlopt_func host_twin, void *f_data)
{ return bind(opt, code_object, name, host_twin, f_data, 1); }
int nlopt_amd_has_device_objective(const nlopt_opt opt) { return opt && opt->userobj ? 1 : 0; }

/* how an objective handed to an algorithm driver is to be evaluated */
void nla_evaluator_resolve(nla_evaluator *ev, nlopt_opt opt, nlopt_func f, void *f_data)
{
    memset(ev, 0, sizeof *ev);
    ev->f = f; ev->f_data = f_data;
    ev->sign = (opt && opt->dev_sign < 0) ? -1. : 1.;
    ev->obj = nlopt_amd_objective_id(f);
    if (ev->obj >= 0) ev->kind = NLA_EVAL_DEVICE;
    else if (f == adapter && f_data) { ev->kind = NLA_EVAL_USER; ev->user = (nla_userobj *) f_data; }
    else { ev->kind = NLA_EVAL_HOST; ev->sign = 1.; }
}

int nla_exact_mode(nlopt_opt opt) { return opt && nlopt_get_param(opt, "amd_exact_dot", 0.) != 0.; }

/* Which summation order a local search uses.  "amd_exact_dot" set on the object (or on MLSL's local optimiser, `also`) decides;
 * unset, a client's own nlopt_func gets the reference's sequential order — its callback sees exactly the points the reference
 * would hand it, which is what a drop-in owes it, and the callback's round trip dwarfs the cost of the ordered sums — while
 * device objectives get the tree reductions (results to rounding). */
int nla_exact_mode_for(nlopt_opt opt, nlopt_opt also, const nla_evaluator *ev)
{
    const int set_a = opt && nlopt_has_param(opt, "amd_exact_dot"), set_b = also && nlopt_has_param(also, "amd_exact_dot");
    int exact = (set_a || set_b) ? ((set_a && nla_exact_mode(opt)) || (set_b && nla_exact_mode(also))) : (ev && ev->kind == NLA_EVAL_HOST);
    /* "amd_lbfgs_streaming" != 0: LD_LBFGS with tree sums on the streaming kernel (hip/lbfgs_kernels.hip) instead of the resident one
     * (hip/lbfgs_resident.hip) — the two are bit-identical, which is what the switch exists to show (tests/test_gpu_lbfgs.py) */
    if ((opt && nlopt_get_param(opt, "amd_lbfgs_streaming", 0) != 0) || (also && nlopt_get_param(also, "amd_lbfgs_streaming", 0) != 0)) return exact ? 3 : 2;
    return exact;
}
