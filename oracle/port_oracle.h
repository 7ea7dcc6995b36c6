/* port_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY; never linked into the product).
 *
 * A plain-C, 64-bit-index restatement of the reference's stochastic-global hot path, used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the *checker* for the HIP
 * path.  Each function cites the reference file:line it follows.  Pinned against the real
 * reference (oracle/_ref/libnlopt_ref.so, built by `make ref`) and against the golden vectors in
 * tests/golden/ by tests/test_oracle_pins.py.
 *
 * Naming: orc_* .  The product library (nlopt_amd/lib/libnlopt_amd.so) exports nlopt_* / nla_*
 * and shares no object code with this directory (objfuncs.h, the objective formulae, is the one
 * shared *source* header — SURVEY.md §7.1 step 1 requires identical objective C on both sides).
 */
#ifndef PORT_ORACLE_H
#define PORT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef double (*orc_func)(unsigned n, const double *x, double *grad, void *data);
typedef void (*orc_mfunc)(unsigned m, double *result, unsigned n, const double *x, double *grad, void *data);

/* result codes = nlopt_result values (src/api/nlopt.h:162-176) */
enum { ORC_FAILURE = -1, ORC_INVALID_ARGS = -2, ORC_OUT_OF_MEMORY = -3, ORC_ROUNDOFF_LIMITED = -4,
       ORC_FORCED_STOP = -5, ORC_SUCCESS = 1, ORC_STOPVAL_REACHED = 2, ORC_FTOL_REACHED = 3,
       ORC_XTOL_REACHED = 4, ORC_MAXEVAL_REACHED = 5, ORC_MAXTIME_REACHED = 6 };

/* ---- MT19937 (src/util/mt19937ar.c) ---------------------------------------------------------- */
void orc_srand(unsigned long seed);                 /* nlopt_init_genrand :80-93 */
uint32_t orc_genrand_int32(void);                   /* :97-131 */
double orc_urand(double a, double b);               /* :194-206 */
int orc_iurand(int n);                              /* :209-212 */
double orc_nrand(double mean, double stddev);       /* :216-232 */
void orc_mt_get_state(uint32_t mt[624], int *mti);
void orc_mt_set_state(const uint32_t mt[624], int mti);
uint64_t orc_mt_words_drawn(void);                  /* words drawn since the last orc_srand */

/* ---- stopping (src/util/stop.c:81-159, nlopt-util.h:79-91) ----------------------------------- */
typedef struct {
    unsigned n;
    double minf_max, ftol_rel, ftol_abs, xtol_rel;
    const double *xtol_abs, *x_weights;
    long nevals, maxeval;
    double maxtime, start;
    int force_stop;
} orc_stop;
void orc_stop_default(orc_stop *s, unsigned n);
int orc_stop_ftol(const orc_stop *s, double f, double oldf);
int orc_stop_f(const orc_stop *s, double f, double oldf);
int orc_stop_x(const orc_stop *s, const double *x, const double *oldx);
int orc_stop_dx(const orc_stop *s, const double *x, const double *dx);
int orc_stop_evals(const orc_stop *s);
int orc_stop_time(const orc_stop *s);
double orc_seconds(void);

/* ---- per-evaluation trace --------------------------------------------------------------------
 * kind: 0 = init row, 1 = reflection trial (T), 2 = local mutation (M)
 * row : init -> the row written; accepted trial -> the row replaced (the then-worst); else -1 */
typedef struct { double f; int64_t row; int32_t kind; int32_t accepted; } orc_trace_rec;
typedef struct { orc_trace_rec *rec; size_t cap, len; } orc_trace;

/* ---- CRS2_LM (src/algs/crs/crs.c) ------------------------------------------------------------ */
int orc_crs_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub,
                     double *x, double *minf, orc_stop *stop, long population, orc_trace *trace);

/* ---- ISRES (src/algs/isres/isres.c) ---------------------------------------------------------- */
typedef struct { orc_func f; void *f_data; double tol; } orc_constraint;      /* scalar, nlopt-util.h:119-126 */
typedef struct { double *f, *pen; size_t cap, len; long generations; } orc_isres_trace;   /* per evaluation */
int orc_isres_minimize(int n, orc_func f, void *f_data, int m, const orc_constraint *fc, int p, const orc_constraint *h,
                       const double *lb, const double *ub, double *x, double *minf, orc_stop *stop, long population,
                       orc_isres_trace *trace);

/* ---- LD_LBFGS (src/algs/luksan/plis.c) -------------------------------------------------------- */
int orc_lbfgs_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                       orc_stop *stop, int mf, double tolg);

/* ---- LD_MMA without nonlinear constraints (src/algs/mma/mma.c) -------------------------------- */
typedef struct { double rho_init, sigma_min; int inner_maxeval, inner_gradients, always_improve, pad;
                 const double *sigma_init; /* the initial step (opt->dx), or NULL */ } orc_mma_params;
int orc_mma_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                     orc_stop *stop, const orc_mma_params *prm);

/* ---- MLSL (src/algs/mlsl/mlsl.c) with LD_LBFGS (alg 0) or LD_MMA (alg 1) as the local optimiser - */
typedef struct { double ftol_rel, ftol_abs, xtol_rel, tolg; long maxeval; int mf; int alg; orc_mma_params mma; } orc_local_params;
typedef struct { double *fsamp, *floc; int *eloc; size_t cap, nsamp, nloc; long iterations;
                 /* optional (NULL: not recorded): per local search the creation index of its start point (0 = the caller's x, 1.. = the samples
                  * in the order drawn); per finished iteration the searches and evaluations so far and the stream words drawn so far */
                 int *sloc; long *it_nloc, *it_nevals; unsigned long long *it_words; size_t it_cap; } orc_mlsl_trace;
/* Sobol LDS (port_sobol.c; sobolseq.c:109-264) */
typedef struct orc_sobol_s orc_sobol;
orc_sobol *orc_sobol_create(unsigned sdim);
void orc_sobol_destroy(orc_sobol *s);
void orc_sobol_next(orc_sobol *s, double *x, const double *lb, const double *ub);
void orc_sobol_next01(orc_sobol *s, double *x);
void orc_sobol_skip(orc_sobol *s, unsigned n, double *x);

int orc_mlsl_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                      orc_stop *stop, int Nsamples, const orc_local_params *loc, orc_mlsl_trace *trace);

/* ---- ESCH (src/algs/esch/esch.c) ---------------------------------------------------------------- */
/* np parents, no offspring (0, 0 -> 40, 60; the dispatcher passes pop and (unsigned)(pop*1.5), optimize.c:946-949);
 * trace: one record per evaluation (kind 0 parent, 1 offspring; row = physical row) */
int orc_esch_minimize(int n, orc_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf, orc_stop *stop,
                      long np, long no, orc_trace *trace);

/* ---- objective zoo callbacks (objfuncs.h compiled for the host) ------------------------------ */
orc_func orc_objective(int id);                     /* f_data ignored */
double orc_con_blocksum(unsigned n, const double *x, double *grad, void *data); /* data -> unsigned[2]={q,Q} */

/* recording wrapper: calls inner, appends f to buf (for pinning against the real reference) */
typedef struct { orc_func inner; void *inner_data; double *fbuf; uint64_t *xhash; size_t cap, len; } orc_recorder;
double orc_recording_callback(unsigned n, const double *x, double *grad, void *data);

#ifdef __cplusplus
}
#endif
#endif
