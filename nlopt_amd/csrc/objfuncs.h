/* objfuncs.h — the synthetic objective zoo of the hot path, written ONCE and compiled three ways:
 *   - by gcc into the product library as host callbacks (nlopt_func signature) — these are the
 *     function pointers a user passes to nlopt_set_min_objective(); the dispatcher recognises them
 *     by pointer identity and runs the device version instead (SURVEY.md §8b "required extension");
 *   - by gcc into oracle/liboracle.so and the oracle harness as the CPU reference's callbacks;
 *   - by hipcc (gfx950) as the per-element terms of the wavefront-parallel device evaluators.
 *
 * Style and formulae follow the reference's objective zoo test/testfuncs.c (n-general Griewank
 * :250-266, Levy :218-240, generalized Rosenbrock :124-140 with the hard-coded 29 -> n-1, the
 * 2*pi literal :32); Rastrigin and Ackley do not exist in the reference zoo and
 * are defined here in the same style.  No global eval counter / printf (testfuncs_status :17-30).
 *
 * Everything is fp64, compiled with -ffp-contract=off on every side (CMakeLists.txt:280-284).
 */
#ifndef NLA_OBJFUNCS_H
#define NLA_OBJFUNCS_H

#include <math.h>

#if defined(__HIPCC__)
#define NLA_HD __host__ __device__ static inline
#else
#define NLA_HD static inline
#endif

/* objective ids (ABI: used by nlopt_amd_objective(), the kernels and the tests) */
#define NLA_OBJ_RASTRIGIN  0
#define NLA_OBJ_ACKLEY     1
#define NLA_OBJ_GRIEWANK   2
#define NLA_OBJ_ROSENBROCK 3   /* generalized (chained) Rosenbrock, n >= 2 */
#define NLA_OBJ_LEVY       4
#define NLA_OBJ_SPHERE     5
#define NLA_OBJ_COUNT      6

/* constraint ids */
#define NLA_CON_BLOCKSUM   0   /* g_q(x) = sum_{i in block q of Q} x_i - 1 <= 0 */

/* sin / cos as the objectives call them.  On the host they go through functions the optimiser cannot look into: glibc's
 * sincos() and cos() (or sin()) round differently for some arguments (glibc 2.35: cos(2 pi 0.5939350286162490) differs in the
 * last bit), and whether gcc merges a sin and a cos of one argument into a sincos call depends on inlining, loop versioning and
 * the -O level — so this same source gave f values one ulp apart in different translation units (the product's callbacks vs
 * the oracle's).  Behind these wrappers every host build calls libm's plain cos / sin.  Device code is unaffected. */
#if defined(__HIPCC__)
#define NLA_COS(x) cos(x)
#define NLA_SIN(x) sin(x)
#else
static __attribute__((noinline, noclone, unused)) double nla_libm_cos(double x) { return cos(x); }
static __attribute__((noinline, noclone, unused)) double nla_libm_sin(double x) { return sin(x); }
#define NLA_COS(x) nla_libm_cos(x)
#define NLA_SIN(x) nla_libm_sin(x)
#endif

#define NLA_PI2 6.283185307179586   /* 2*pi, same literal as test/testfuncs.c:32 */
#define NLA_PI3 9.424777960769379   /* 3*pi, test/testfuncs.c:33 */
#define NLA_E   2.718281828459045

NLA_HD double nla_sqr(double x) { return x * x; }

/* ---- per-element terms (shared host/device) ------------------------------------------------- */
NLA_HD double nla_rastrigin_term(double x) { return x * x - 10.0 * NLA_COS(NLA_PI2 * x); }
NLA_HD double nla_ackley_cos_term(double x) { return NLA_COS(NLA_PI2 * x); }
NLA_HD double nla_griewank_sum_term(double x) { return nla_sqr(x) * 0.00025; }
NLA_HD double nla_griewank_prod_term(double x, unsigned i) { return NLA_COS(x / sqrt(i + 1.)); }
NLA_HD double nla_rosenbrock_term(double xi, double xi1)
{
    double a = xi1 - xi * xi, b = 1 - xi;
    return 100 * nla_sqr(a) + nla_sqr(b);
}
/* Levy body term for i < n-1: (x_i - 1)^2 (1 + sin^2(3 pi x_{i+1})) */
NLA_HD double nla_levy_term(double xi, double xi1)
{
    double a = xi - 1, b = 1 + nla_sqr(NLA_SIN(NLA_PI3 * xi1));
    return nla_sqr(a) * b;
}
/* Levy head: sin^2(3 pi x_0) + (x_{n-1} - 1)(1 + sin^2(2 pi x_{n-1})) */
NLA_HD double nla_levy_head(double x0, double xl)
{
    double a = xl - 1, b = 1 + nla_sqr(NLA_SIN(NLA_PI2 * xl));
    return nla_sqr(NLA_SIN(NLA_PI3 * x0)) + a * b;
}
NLA_HD double nla_ackley_finish(double sumsq, double sumcos, unsigned n)
{
    return -20.0 * exp(-0.2 * sqrt(sumsq / n)) - exp(sumcos / n) + 20.0 + NLA_E;
}

/* default search boxes (per coordinate) */
NLA_HD void nla_obj_box(int id, double *lo, double *hi)
{
    switch (id) {
    case NLA_OBJ_RASTRIGIN:  *lo = -5.12;   *hi = 5.12;   break;
    case NLA_OBJ_ACKLEY:     *lo = -32.768; *hi = 32.768; break;
    case NLA_OBJ_GRIEWANK:   *lo = -500;    *hi = 600;    break;  /* testfuncs.c:268-269 */
    case NLA_OBJ_ROSENBROCK: *lo = -30;     *hi = 30;     break;  /* testfuncs.c:142-143 */
    case NLA_OBJ_LEVY:       *lo = -10;     *hi = 10;     break;  /* testfuncs.c:244-245 (levy4) */
    default:                 *lo = -10;     *hi = 10;     break;
    }
}

/* ---- sequential host evaluators (sum order i = 0..n-1, one accumulator, as the zoo does) ----- */
#if !defined(__HIP_DEVICE_COMPILE__)
static inline double nla_obj_eval_seq(int id, unsigned n, const double *x, double *grad)
{
    unsigned i;
    switch (id) {
    case NLA_OBJ_RASTRIGIN: {
        double f = 10.0 * n;
        for (i = 0; i < n; ++i) {
            f += nla_rastrigin_term(x[i]);
            if (grad) grad[i] = 2 * x[i] + 10.0 * NLA_PI2 * NLA_SIN(NLA_PI2 * x[i]);
        }
        return f;
    }
    case NLA_OBJ_ACKLEY: {
        double s = 0, c = 0, r, e1, e2;
        for (i = 0; i < n; ++i) { s += nla_sqr(x[i]); c += nla_ackley_cos_term(x[i]); }
        if (grad) {
            r = sqrt(s / n);
            e1 = exp(-0.2 * r);
            e2 = exp(c / n);
            for (i = 0; i < n; ++i) {
                double g = e2 * NLA_PI2 * NLA_SIN(NLA_PI2 * x[i]) / n;
                if (r > 0) g += 4.0 * e1 * x[i] / (n * r);   /* d/dx of -20 exp(-0.2 r); r=0 guarded */
                grad[i] = g;
            }
        }
        return nla_ackley_finish(s, c, n);
    }
    case NLA_OBJ_GRIEWANK: {   /* test/testfuncs.c:250-266, serial product order */
        double f = 1, p = 1;
        for (i = 0; i < n; ++i) {
            f += nla_griewank_sum_term(x[i]);
            p *= nla_griewank_prod_term(x[i], i);
            if (grad) grad[i] = x[i] * 0.0005;
        }
        f -= p;
        if (grad)
            for (i = 0; i < n; ++i)
                grad[i] += p * tan(x[i] / sqrt(i + 1.)) / sqrt(i + 1.);
        return f;
    }
    case NLA_OBJ_ROSENBROCK: { /* test/testfuncs.c:124-140 with 29 -> n-1 */
        double f = 0;
        if (grad) grad[0] = 0;
        for (i = 0; i + 1 < n; ++i) {
            double a = x[i + 1] - x[i] * x[i], b = 1 - x[i];
            if (grad) {
                grad[i] += -400 * a * x[i] - 2 * b;
                grad[i + 1] = 200 * a;
            }
            f += nla_rosenbrock_term(x[i], x[i + 1]);
        }
        return f;
    }
    case NLA_OBJ_LEVY: {       /* test/testfuncs.c:218-240 */
        double f = nla_levy_head(x[0], x[n - 1]);
        if (grad) {
            double a = x[n - 1] - 1, b = 1 + nla_sqr(NLA_SIN(NLA_PI2 * x[n - 1]));
            for (i = 0; i < n; ++i) grad[i] = 0;
            grad[0] = 2 * NLA_PI3 * NLA_SIN(NLA_PI3 * x[0]) * NLA_COS(NLA_PI3 * x[0]);
            grad[n - 1] += b + a * 2 * NLA_PI2 * NLA_SIN(NLA_PI2 * x[n - 1]) * NLA_COS(NLA_PI2 * x[n - 1]);
        }
        for (i = 0; i + 1 < n; ++i) {
            f += nla_levy_term(x[i], x[i + 1]);
            if (grad) {
                double a = x[i] - 1, b = 1 + nla_sqr(NLA_SIN(NLA_PI3 * x[i + 1]));
                grad[i] += 2 * a * b;
                grad[i + 1] += 2 * NLA_PI3 * nla_sqr(a) * NLA_SIN(NLA_PI3 * x[i + 1]) * NLA_COS(NLA_PI3 * x[i + 1]);
            }
        }
        return f;
    }
    case NLA_OBJ_SPHERE: {
        double f = 0;
        for (i = 0; i < n; ++i) { f += nla_sqr(x[i]); if (grad) grad[i] = 2 * x[i]; }
        return f;
    }
    default:
        return HUGE_VAL;
    }
}

/* block-sum inequality constraint q of Q (SURVEY.md §8d proposal): sum_{i in block} x_i - 1 <= 0 */
static inline double nla_con_blocksum_seq(unsigned n, const double *x, double *grad, unsigned q, unsigned Q)
{
    unsigned i, lo = (unsigned) (((unsigned long long) q * n) / Q), hi = (unsigned) (((unsigned long long) (q + 1) * n) / Q);
    double s = 0;
    if (grad) for (i = 0; i < n; ++i) grad[i] = 0;
    for (i = lo; i < hi; ++i) { s += x[i]; if (grad) grad[i] = 1; }
    return s - 1;
}
#endif /* !__HIP_DEVICE_COMPILE__ */

#endif /* NLA_OBJFUNCS_H */
