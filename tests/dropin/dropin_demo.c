/* dropin_demo.c — a plain NLopt client program, used to show that libnlopt_amd.so is a drop-in for the reference's
 * shared library at the C ABI: the SAME binary is pointed at either library (resolved with dlopen so
 * that one executable serves both), makes only reference-API calls (src/api/nlopt.h:203-284) and prints what it got.
 *
 *   dropin_demo <lib.so> <algorithm id> <n> <population> <maxeval> <seed> [device|host] [local algorithm id] [exact]
 *
 * Objective: Rastrigin as a C callback of the program (nlopt_func, nlopt.h:60-62); every call is counted and the bits of
 * x are folded into a hash, so two runs print the same line only if they evaluated the same points in the same order.
 * With the 7th argument "device" and a library that has nlopt_amd_objective (ours), the registered device objective
 * is used instead (then there are no callbacks to hash).  With a local algorithm id (11 = LD_LBFGS, 24 = LD_MMA) the
 * object gets that local optimiser (ftol_rel 1e-8) through nlopt_set_local_optimizer — the G_MLSL case; the callback
 * then also serves gradients.  "exact" sets the generic parameter amd_exact_dot = 1 (nlopt_set_param; the reference
 * stores and ignores unknown names): libnlopt_amd then accumulates its sums in the reference's order. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct nlopt_opt_s *nlopt_opt;
typedef double (*nlopt_func)(unsigned n, const double *x, double *gradient, void *func_data);

static unsigned long ncalls;
static uint64_t xhash = 1469598103934665603ULL;

static double rastrigin(unsigned n, const double *x, double *grad, void *data)
{
    double s = 10.0 * n;
    unsigned i;
    (void) data;
    for (i = 0; i < n; ++i) {
        uint64_t b;
        memcpy(&b, x + i, 8);
        xhash = (xhash ^ b) * 1099511628211ULL;
        s += x[i] * x[i] - 10.0 * cos(6.283185307179586 * x[i]);
        if (grad) grad[i] = 2 * x[i] + 10.0 * 6.283185307179586 * sin(6.283185307179586 * x[i]);
    }
    xhash = (xhash ^ (grad ? 0x9e3779b97f4a7c15ULL : 0)) * 1099511628211ULL;     /* whether a gradient was asked for is part of the record */
    ++ncalls;
    return s;
}

typedef nlopt_opt (*create_t)(int, unsigned);
typedef void (*destroy_t)(nlopt_opt);
typedef int (*set_d_t)(nlopt_opt, double);
typedef int (*set_obj_t)(nlopt_opt, nlopt_func, void *);
typedef int (*set_u_t)(nlopt_opt, unsigned);
typedef int (*set_i_t)(nlopt_opt, int);
typedef int (*get_i_t)(nlopt_opt);
typedef void (*srand_t)(unsigned long);
typedef int (*optimize_t)(nlopt_opt, double *, double *);
typedef const char *(*algname_t)(int);
typedef const char *(*errmsg_t)(nlopt_opt);
typedef int (*set_local_t)(nlopt_opt, nlopt_opt);
typedef int (*set_param_t)(nlopt_opt, const char *, double);
#define SYM(T, name) T name = (T) dlsym(h, #name); if (!name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }

int main(int argc, char **argv)
{
    void *h;
    int alg, n, pop, maxeval, i, ret, device, local_alg, exact;
    unsigned long seed;
    double *x, minf = 0;
    nlopt_opt opt;
    nlopt_func f = rastrigin;
    if (argc < 7) { fprintf(stderr, "usage: %s lib.so algorithm n population maxeval seed [device]\n", argv[0]); return 2; }
    h = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    alg = atoi(argv[2]); n = atoi(argv[3]); pop = atoi(argv[4]); maxeval = atoi(argv[5]); seed = strtoul(argv[6], NULL, 10);
    device = argc > 7 && !strcmp(argv[7], "device");
    local_alg = argc > 8 ? atoi(argv[8]) : 0;
    exact = argc > 9 && !strcmp(argv[9], "exact");
    {
        SYM(create_t, nlopt_create)
        SYM(destroy_t, nlopt_destroy)
        SYM(set_d_t, nlopt_set_lower_bounds1)
        SYM(set_d_t, nlopt_set_upper_bounds1)
        SYM(set_obj_t, nlopt_set_min_objective)
        SYM(set_u_t, nlopt_set_population)
        SYM(set_i_t, nlopt_set_maxeval)
        SYM(get_i_t, nlopt_get_numevals)
        SYM(srand_t, nlopt_srand)
        SYM(optimize_t, nlopt_optimize)
        SYM(algname_t, nlopt_algorithm_name)
        SYM(errmsg_t, nlopt_get_errmsg)
        SYM(set_local_t, nlopt_set_local_optimizer)
        SYM(set_d_t, nlopt_set_ftol_rel)
        SYM(set_param_t, nlopt_set_param)
        if (device) {
            nlopt_func (*amd_obj)(int) = (nlopt_func (*)(int)) dlsym(h, "nlopt_amd_objective");
            if (!amd_obj) { fprintf(stderr, "this library has no device objectives\n"); return 2; }
            f = amd_obj(0);                                   /* NLOPT_AMD_OBJ_RASTRIGIN */
        }
        opt = nlopt_create(alg, (unsigned) n);
        if (!opt) { fprintf(stderr, "nlopt_create failed\n"); return 1; }
        x = (double *) malloc(sizeof(double) * (size_t) n);
        for (i = 0; i < n; ++i) x[i] = -5.12 + 10.24 * fmod((i + 1) * 0.6180339887498949, 1.0);
        nlopt_set_lower_bounds1(opt, -5.12);
        nlopt_set_upper_bounds1(opt, 5.12);
        nlopt_set_min_objective(opt, f, NULL);
        if (pop) nlopt_set_population(opt, (unsigned) pop);
        nlopt_set_maxeval(opt, maxeval);
        if (local_alg) {
            nlopt_opt loc = nlopt_create(local_alg, (unsigned) n);
            if (!loc) { fprintf(stderr, "nlopt_create(local) failed\n"); return 1; }
            nlopt_set_ftol_rel(loc, 1e-8);
            if (nlopt_set_local_optimizer(opt, loc) < 0) { fprintf(stderr, "nlopt_set_local_optimizer failed\n"); return 1; }
            nlopt_destroy(loc);
        } else if (alg == 11 || alg == 24) nlopt_set_ftol_rel(opt, 1e-10);
        if (exact) nlopt_set_param(opt, "amd_exact_dot", 1.0);
        nlopt_srand(seed);
        ret = nlopt_optimize(opt, x, &minf);
        printf("%s: result %d, minf %.17g, x[0] %.17g, x[n-1] %.17g, numevals %d, callbacks %lu, xhash %016llx\n",
               nlopt_algorithm_name(alg), ret, minf, x[0], x[n - 1], nlopt_get_numevals(opt), ncalls, (unsigned long long) xhash);
        if (ret < 0 && nlopt_get_errmsg(opt)) fprintf(stderr, "errmsg: %s\n", nlopt_get_errmsg(opt));
        nlopt_destroy(opt);
        free(x);
    }
    return ret < 0 ? 1 : 0;
}
