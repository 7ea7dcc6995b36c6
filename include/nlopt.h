/* nlopt.h — public C API of libnlopt_amd, the MI355X-native drop-in for NLopt's stochastic
 * population-based global optimisers.
 *
 * This header declares the same C ABI as the reference's src/api/nlopt.h (NLopt 2.11.0): the
 * same function names and signatures, the same enum *values* (they are ABI: nlopt.h:72-154 for
 * nlopt_algorithm, :162-176 for nlopt_result), the same callback typedefs (:60-70).  A program
 * compiled against the reference header links against this library unchanged.
 *
 * Implemented algorithms (everything else returns NLOPT_INVALID_ARGS from nlopt_optimize with an
 * errmsg saying so — see DESIGN.md "out of scope"):
 *     NLOPT_GN_CRS2_LM (19)   NLOPT_GN_ISRES (35)   NLOPT_GN_ESCH (42)
 *     NLOPT_GN_MLSL / GD_MLSL / GN_MLSL_LDS / GD_MLSL_LDS (20-23), NLOPT_G_MLSL / G_MLSL_LDS (38,39)
 *     the local optimisers MLSL runs (and nlopt_optimize serves on their own): NLOPT_LD_LBFGS (11), NLOPT_LD_MMA (24; with
 *     nonlinear inequality constraints too), NLOPT_LN_COBYLA (25)
 *     as callers of the above: NLOPT_AUGLAG / AUGLAG_EQ and the LN_ / LD_ variants (36, 37, 30-33)
 *     the pre-2.0 one-call interface nlopt_minimize* maps onto the same
 *
 * Objectives.  A GPU cannot call a host callback, so there are three kinds: (1) one of the function pointers returned by
 * nlopt_amd_objective() (include/nlopt_amd.h) — the population / the local searches are evaluated inside the HIP kernels;
 * (2) a device objective of your own, compiled as a code object (include/nlopt_amd_device.h, nlopt_amd_set_min_device_objective);
 * (3) any other nlopt_func: served by every algorithm above through the exact host-callback path — candidates and iterates are built
 * on the device, x is copied back and f is called on the caller's thread, one x at a time, in the reference's order (the
 * reference's callback contract, nlopt.h:60-62).
 */
#ifndef NLOPT_H
#define NLOPT_H

#include <stddef.h>

#define NLOPT_STDCALL
#define NLOPT_EXTERN(T) extern T NLOPT_STDCALL

#ifdef __cplusplus
extern "C" {
#endif

/* reference: src/api/nlopt.h:60-70 */
typedef double (*nlopt_func)(unsigned n, const double *x, double *gradient /* NULL if not needed */, void *func_data);
typedef void (*nlopt_mfunc)(unsigned m, double *result, unsigned n, const double *x,
                            double *gradient /* NULL if not needed */, void *func_data);
typedef void (*nlopt_precond)(unsigned n, const double *x, const double *v, double *vpre, void *data);

/* reference: src/api/nlopt.h:72-154.  Values are ABI and must not be renumbered. */
typedef enum {
    NLOPT_GN_DIRECT = 0, NLOPT_GN_DIRECT_L = 1, NLOPT_GN_DIRECT_L_RAND = 2, NLOPT_GN_DIRECT_NOSCAL = 3,
    NLOPT_GN_DIRECT_L_NOSCAL = 4, NLOPT_GN_DIRECT_L_RAND_NOSCAL = 5, NLOPT_GN_ORIG_DIRECT = 6,
    NLOPT_GN_ORIG_DIRECT_L = 7, NLOPT_GD_STOGO = 8, NLOPT_GD_STOGO_RAND = 9, NLOPT_LD_LBFGS_NOCEDAL = 10,
    NLOPT_LD_LBFGS = 11, NLOPT_LN_PRAXIS = 12, NLOPT_LD_VAR1 = 13, NLOPT_LD_VAR2 = 14, NLOPT_LD_TNEWTON = 15,
    NLOPT_LD_TNEWTON_RESTART = 16, NLOPT_LD_TNEWTON_PRECOND = 17, NLOPT_LD_TNEWTON_PRECOND_RESTART = 18,
    NLOPT_GN_CRS2_LM = 19, NLOPT_GN_MLSL = 20, NLOPT_GD_MLSL = 21, NLOPT_GN_MLSL_LDS = 22, NLOPT_GD_MLSL_LDS = 23,
    NLOPT_LD_MMA = 24, NLOPT_LN_COBYLA = 25, NLOPT_LN_NEWUOA = 26, NLOPT_LN_NEWUOA_BOUND = 27,
    NLOPT_LN_NELDERMEAD = 28, NLOPT_LN_SBPLX = 29, NLOPT_LN_AUGLAG = 30, NLOPT_LD_AUGLAG = 31,
    NLOPT_LN_AUGLAG_EQ = 32, NLOPT_LD_AUGLAG_EQ = 33, NLOPT_LN_BOBYQA = 34, NLOPT_GN_ISRES = 35,
    NLOPT_AUGLAG = 36, NLOPT_AUGLAG_EQ = 37, NLOPT_G_MLSL = 38, NLOPT_G_MLSL_LDS = 39, NLOPT_LD_SLSQP = 40,
    NLOPT_LD_CCSAQ = 41, NLOPT_GN_ESCH = 42, NLOPT_GN_AGS = 43,
    NLOPT_NUM_ALGORITHMS
} nlopt_algorithm;

NLOPT_EXTERN(const char *) nlopt_algorithm_name(nlopt_algorithm a);
NLOPT_EXTERN(const char *) nlopt_algorithm_to_string(nlopt_algorithm algorithm);
NLOPT_EXTERN(nlopt_algorithm) nlopt_algorithm_from_string(const char *name);

/* reference: src/api/nlopt.h:162-176 */
typedef enum {
    NLOPT_FAILURE = -1, NLOPT_INVALID_ARGS = -2, NLOPT_OUT_OF_MEMORY = -3, NLOPT_ROUNDOFF_LIMITED = -4,
    NLOPT_FORCED_STOP = -5, NLOPT_NUM_FAILURES = -6,
    NLOPT_SUCCESS = 1, NLOPT_STOPVAL_REACHED = 2, NLOPT_FTOL_REACHED = 3, NLOPT_XTOL_REACHED = 4,
    NLOPT_MAXEVAL_REACHED = 5, NLOPT_MAXTIME_REACHED = 6, NLOPT_NUM_RESULTS
} nlopt_result;
#define NLOPT_MINF_MAX_REACHED NLOPT_STOPVAL_REACHED

NLOPT_EXTERN(const char *) nlopt_result_to_string(nlopt_result result);
NLOPT_EXTERN(nlopt_result) nlopt_result_from_string(const char *name);

/* reference: src/api/nlopt.h:184-187 */
NLOPT_EXTERN(void) nlopt_srand(unsigned long seed);
NLOPT_EXTERN(void) nlopt_srand_time(void);
NLOPT_EXTERN(void) nlopt_version(int *major, int *minor, int *bugfix);

struct nlopt_opt_s;
typedef struct nlopt_opt_s *nlopt_opt;

/* reference: src/api/nlopt.h:203-218 */
NLOPT_EXTERN(nlopt_opt) nlopt_create(nlopt_algorithm algorithm, unsigned n);
NLOPT_EXTERN(void) nlopt_destroy(nlopt_opt opt);
NLOPT_EXTERN(nlopt_opt) nlopt_copy(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_optimize(nlopt_opt opt, double *x, double *opt_f);
NLOPT_EXTERN(nlopt_result) nlopt_set_min_objective(nlopt_opt opt, nlopt_func f, void *f_data);
NLOPT_EXTERN(nlopt_result) nlopt_set_max_objective(nlopt_opt opt, nlopt_func f, void *f_data);
NLOPT_EXTERN(nlopt_result) nlopt_set_precond_min_objective(nlopt_opt opt, nlopt_func f, nlopt_precond pre, void *f_data);
NLOPT_EXTERN(nlopt_result) nlopt_set_precond_max_objective(nlopt_opt opt, nlopt_func f, nlopt_precond pre, void *f_data);
NLOPT_EXTERN(nlopt_algorithm) nlopt_get_algorithm(const nlopt_opt opt);
NLOPT_EXTERN(unsigned) nlopt_get_dimension(const nlopt_opt opt);
NLOPT_EXTERN(const char *) nlopt_get_errmsg(nlopt_opt opt);

/* reference: src/api/nlopt.h:221-225 */
NLOPT_EXTERN(nlopt_result) nlopt_set_param(nlopt_opt opt, const char *name, double val);
NLOPT_EXTERN(double) nlopt_get_param(const nlopt_opt opt, const char *name, double defaultval);
NLOPT_EXTERN(int) nlopt_has_param(const nlopt_opt opt, const char *name);
NLOPT_EXTERN(unsigned) nlopt_num_params(const nlopt_opt opt);
NLOPT_EXTERN(const char *) nlopt_nth_param(const nlopt_opt opt, unsigned n);

/* reference: src/api/nlopt.h:229-246 */
NLOPT_EXTERN(nlopt_result) nlopt_set_lower_bounds(nlopt_opt opt, const double *lb);
NLOPT_EXTERN(nlopt_result) nlopt_set_lower_bounds1(nlopt_opt opt, double lb);
NLOPT_EXTERN(nlopt_result) nlopt_set_lower_bound(nlopt_opt opt, int i, double lb);
NLOPT_EXTERN(nlopt_result) nlopt_get_lower_bounds(const nlopt_opt opt, double *lb);
NLOPT_EXTERN(nlopt_result) nlopt_set_upper_bounds(nlopt_opt opt, const double *ub);
NLOPT_EXTERN(nlopt_result) nlopt_set_upper_bounds1(nlopt_opt opt, double ub);
NLOPT_EXTERN(nlopt_result) nlopt_set_upper_bound(nlopt_opt opt, int i, double ub);
NLOPT_EXTERN(nlopt_result) nlopt_get_upper_bounds(const nlopt_opt opt, double *ub);
NLOPT_EXTERN(nlopt_result) nlopt_remove_inequality_constraints(nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_add_inequality_constraint(nlopt_opt opt, nlopt_func fc, void *fc_data, double tol);
NLOPT_EXTERN(nlopt_result) nlopt_add_precond_inequality_constraint(nlopt_opt opt, nlopt_func fc, nlopt_precond pre, void *fc_data, double tol);
NLOPT_EXTERN(nlopt_result) nlopt_add_inequality_mconstraint(nlopt_opt opt, unsigned m, nlopt_mfunc fc, void *fc_data, const double *tol);
NLOPT_EXTERN(nlopt_result) nlopt_remove_equality_constraints(nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_add_equality_constraint(nlopt_opt opt, nlopt_func h, void *h_data, double tol);
NLOPT_EXTERN(nlopt_result) nlopt_add_precond_equality_constraint(nlopt_opt opt, nlopt_func h, nlopt_precond pre, void *h_data, double tol);
NLOPT_EXTERN(nlopt_result) nlopt_add_equality_mconstraint(nlopt_opt opt, unsigned m, nlopt_mfunc h, void *h_data, const double *tol);

/* reference: src/api/nlopt.h:250-277 */
NLOPT_EXTERN(nlopt_result) nlopt_set_stopval(nlopt_opt opt, double stopval);
NLOPT_EXTERN(double) nlopt_get_stopval(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_ftol_rel(nlopt_opt opt, double tol);
NLOPT_EXTERN(double) nlopt_get_ftol_rel(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_ftol_abs(nlopt_opt opt, double tol);
NLOPT_EXTERN(double) nlopt_get_ftol_abs(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_xtol_rel(nlopt_opt opt, double tol);
NLOPT_EXTERN(double) nlopt_get_xtol_rel(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_xtol_abs1(nlopt_opt opt, double tol);
NLOPT_EXTERN(nlopt_result) nlopt_set_xtol_abs(nlopt_opt opt, const double *tol);
NLOPT_EXTERN(nlopt_result) nlopt_get_xtol_abs(const nlopt_opt opt, double *tol);
NLOPT_EXTERN(nlopt_result) nlopt_set_x_weights1(nlopt_opt opt, double w);
NLOPT_EXTERN(nlopt_result) nlopt_set_x_weights(nlopt_opt opt, const double *w);
NLOPT_EXTERN(nlopt_result) nlopt_get_x_weights(const nlopt_opt opt, double *w);
NLOPT_EXTERN(nlopt_result) nlopt_set_maxeval(nlopt_opt opt, int maxeval);
NLOPT_EXTERN(int) nlopt_get_maxeval(const nlopt_opt opt);
NLOPT_EXTERN(int) nlopt_get_numevals(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_maxtime(nlopt_opt opt, double maxtime);
NLOPT_EXTERN(double) nlopt_get_maxtime(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_force_stop(nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_force_stop(nlopt_opt opt, int val);
NLOPT_EXTERN(int) nlopt_get_force_stop(const nlopt_opt opt);

/* reference: src/api/nlopt.h:281-301 */
NLOPT_EXTERN(nlopt_result) nlopt_set_local_optimizer(nlopt_opt opt, const nlopt_opt local_opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_population(nlopt_opt opt, unsigned pop);
NLOPT_EXTERN(unsigned) nlopt_get_population(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_vector_storage(nlopt_opt opt, unsigned dim);
NLOPT_EXTERN(unsigned) nlopt_get_vector_storage(const nlopt_opt opt);
NLOPT_EXTERN(nlopt_result) nlopt_set_default_initial_step(nlopt_opt opt, const double *x);
NLOPT_EXTERN(nlopt_result) nlopt_set_initial_step(nlopt_opt opt, const double *dx);
NLOPT_EXTERN(nlopt_result) nlopt_set_initial_step1(nlopt_opt opt, double dx);
NLOPT_EXTERN(nlopt_result) nlopt_get_initial_step(const nlopt_opt opt, const double *x, double *dx);

typedef void *(*nlopt_munge)(void *p);
NLOPT_EXTERN(void) nlopt_set_munge(nlopt_opt opt, nlopt_munge munge_on_destroy, nlopt_munge munge_on_copy);
typedef void *(*nlopt_munge2)(void *p, void *data);
NLOPT_EXTERN(void) nlopt_munge_data(nlopt_opt opt, nlopt_munge2 munge, void *data);

/* the pre-2.0 one-call interface (reference: src/api/nlopt.h:317-337; behaviour src/api/deprecated.c:65-189) */
typedef double (*nlopt_func_old)(int n, const double *x, double *gradient, void *func_data);
NLOPT_EXTERN(nlopt_result) nlopt_minimize(nlopt_algorithm algorithm, int n, nlopt_func_old f, void *f_data,
                                          const double *lb, const double *ub, double *x, double *minf,
                                          double minf_max, double ftol_rel, double ftol_abs, double xtol_rel, const double *xtol_abs,
                                          int maxeval, double maxtime);
NLOPT_EXTERN(nlopt_result) nlopt_minimize_constrained(nlopt_algorithm algorithm, int n, nlopt_func_old f, void *f_data,
                                                      int m, nlopt_func_old fc, void *fc_data, ptrdiff_t fc_datum_size,
                                                      const double *lb, const double *ub, double *x, double *minf,
                                                      double minf_max, double ftol_rel, double ftol_abs, double xtol_rel,
                                                      const double *xtol_abs, int maxeval, double maxtime);
NLOPT_EXTERN(nlopt_result) nlopt_minimize_econstrained(nlopt_algorithm algorithm, int n, nlopt_func_old f, void *f_data,
                                                       int m, nlopt_func_old fc, void *fc_data, ptrdiff_t fc_datum_size,
                                                       int p, nlopt_func_old h, void *h_data, ptrdiff_t h_datum_size,
                                                       const double *lb, const double *ub, double *x, double *minf,
                                                       double minf_max, double ftol_rel, double ftol_abs, double xtol_rel,
                                                       const double *xtol_abs, double htol_rel, double htol_abs,
                                                       int maxeval, double maxtime);

/* deprecated-API globals whose semantics the dispatcher still honours
 * (reference: src/api/deprecated.c:28-61, read by POP() at src/api/optimize.c:511) */
NLOPT_EXTERN(int) nlopt_get_stochastic_population(void);
NLOPT_EXTERN(void) nlopt_set_stochastic_population(int pop);
NLOPT_EXTERN(void) nlopt_get_local_search_algorithm(nlopt_algorithm *deriv, nlopt_algorithm *nonderiv, int *maxeval);
NLOPT_EXTERN(void) nlopt_set_local_search_algorithm(nlopt_algorithm deriv, nlopt_algorithm nonderiv, int maxeval);

#ifdef __cplusplus
}
#endif
#endif /* NLOPT_H */
