/* nla_internal.h — internal declarations of libnlopt_amd's C host side.
 * Mirrors the roles of the reference's src/api/nlopt-internal.h (struct nlopt_opt_s :40-88) and
 * src/util/nlopt-util.h (nlopt_stopping :79-91, nlopt_constraint :119-126); own layout. */
#ifndef NLA_INTERNAL_H
#define NLA_INTERNAL_H

#include <stddef.h>
#include <stdint.h>
#include "../../include/nlopt.h"
#include "../../include/nlopt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NLA_MT_M 397
#define NLA_MT_DEG 19937
#define NLA_MT_MAXPOW2 48

#define NLA_VERSION_MAJOR 2     /* API level of the reference this library is a drop-in for */
#define NLA_VERSION_MINOR 11
#define NLA_VERSION_BUGFIX 0

/* ---- stopping criteria (reference: nlopt-util.h:79-91, stop.c:81-159) ------------------------ */
typedef struct {
    unsigned n;
    double minf_max, ftol_rel, ftol_abs, xtol_rel;
    const double *xtol_abs, *x_weights;
    int *nevals_p, maxeval;
    double maxtime, start;
    int *force_stop;
    char **stop_msg;
} nla_stopping;

int nla_stop_ftol(const nla_stopping *s, double f, double oldf);
int nla_stop_f(const nla_stopping *s, double f, double oldf);
int nla_stop_x(const nla_stopping *s, const double *x, const double *oldx);
int nla_stop_dx(const nla_stopping *s, const double *x, const double *dx);
int nla_stop_evals(const nla_stopping *s);
int nla_stop_time(const nla_stopping *s);
int nla_stop_forced(const nla_stopping *s);
void nla_stop_msg(const nla_stopping *s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int nla_isinf(double x);
int nla_istiny(double x);
double nla_seconds(void);
unsigned long nla_time_seed(void);
void nla_init_genrand(unsigned long seed);
long nla_thread_id(void);

/* ---- constraints (reference: nlopt-util.h:119-126) ------------------------------------------- */
typedef struct {
    unsigned m;
    nlopt_func f;
    nlopt_mfunc mf;
    nlopt_precond pre;
    void *f_data;
    double *tol;
} nla_constraint;

typedef struct { char *name; double val; } nla_param;

/* ---- the optimiser object (reference: nlopt-internal.h:40-88) -------------------------------- */
struct nlopt_opt_s {
    nlopt_algorithm algorithm;
    unsigned n;
    nlopt_func f; void *f_data; nlopt_precond pre; int maximize;
    nla_param *params; unsigned nparams;
    double *lb, *ub;
    unsigned m, m_alloc; nla_constraint *fc;
    unsigned p, p_alloc; nla_constraint *h;
    nlopt_munge munge_on_destroy, munge_on_copy;
    double stopval, ftol_rel, ftol_abs, xtol_rel, *xtol_abs, *x_weights;
    int maxeval, numevals;
    double maxtime;
    int force_stop;
    struct nlopt_opt_s *force_stop_child;
    nlopt_opt local_opt;
    unsigned stochastic_population;
    double *dx;
    unsigned vector_storage;
    char *errmsg;
    /* --- libnlopt_amd additions (not in the reference) --- */
    nlopt_amd_trace_rec *trace; size_t trace_cap, trace_len;
    nlopt_amd_stats stats;
    nlopt_amd_progress_fn progress; void *progress_data;
    nlopt_amd_comm *comm;           /* multi-GPU run: borrowed communicator (comm.c), NULL = single process */
    struct nla_userobj *userobj;    /* user-supplied device objective (userobj.c), shared by copies (reference counted) */
    int dev_sign;                   /* -1 while a maximisation runs on a device objective without the host flip wrapper */
};

/* ---- how an objective is evaluated ----------------------------------------------------------------------- */
enum { NLA_EVAL_DEVICE = 0, NLA_EVAL_HOST = 1, NLA_EVAL_USER = 2 };
typedef struct nla_userobj nla_userobj;
typedef struct {
    int kind, obj;                  /* NLA_EVAL_DEVICE: obj = compiled-in objective id */
    nlopt_func f; void *f_data;     /* NLA_EVAL_HOST: the caller's callback */
    nla_userobj *user;              /* NLA_EVAL_USER */
    double sign;                    /* +1, or -1: minimise -f (device / user objectives under nlopt_set_max_objective) */
} nla_evaluator;
void nla_evaluator_resolve(nla_evaluator *ev, nlopt_opt opt, nlopt_func f, void *f_data);
int nla_exact_mode(nlopt_opt opt);                    /* nlopt_set_param(opt, "amd_exact_dot", 1) */
int nla_exact_mode_for(nlopt_opt opt, nlopt_opt also, const nla_evaluator *ev);   /* unset: exact order iff the objective is a host callback */

/* user device objectives (userobj.c) */
nla_userobj *nla_userobj_retain(nla_userobj *u);
void nla_userobj_release(nla_userobj *u);
int nla_userobj_is_adapter(nlopt_func f);
/* F[c] = sign * f(row c of X), G row c = sign * gradient (G may be NULL), c < count; rows ld apart */
int nla_userobj_eval_rows(nla_userobj *u, int n, int ld, int64_t count, const double *X, double *F, double *G, double sign, void *stream);
/* the same for the rows named by list[0..m): entry i >= 0: row i with its gradient; entry -(i+1): row i, value only */
int nla_userobj_evalgrad_list(nla_userobj *u, int n, int ld, int m, const int32_t *d_list, const double *X, double *F, double *G,
                              double sign, void *stream);

const char *nla_set_errmsg(nlopt_opt opt, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void nla_unset_errmsg(nlopt_opt opt);
char *nla_vsprintf(char *p, const char *fmt, __builtin_va_list ap);

extern int nla_stochastic_population;                 /* deprecated.c:61 semantics */
extern nlopt_algorithm nla_local_search_alg_deriv, nla_local_search_alg_nonderiv;
extern int nla_local_search_maxeval;

nlopt_result nla_optimize_limited(nlopt_opt opt, double *x, double *minf, int maxeval, double maxtime);

/* ---- collectives (comm.c) ---------------------------------------------------------------------- */
int nla_comm_allgather_dev(nlopt_amd_comm *c, const void *d_send, void *d_recv, size_t bytes, void *stream);
int nla_comm_allgather_host(nlopt_amd_comm *c, const void *h_send, void *h_recv, size_t bytes, void *stream);
void nla_comm_partition(const nlopt_amd_comm *c, int64_t count, int64_t *per, int64_t *first, int64_t *mine);
void nla_stop_view(const nla_stopping *stop, int forced, int timed, nla_stopping *view, int *force_store);
int nla_comm_reserve(nlopt_amd_comm *c, size_t bytes);     /* set-up: staging for all-gathers of up to `bytes` per rank */
void nla_comm_abort(nlopt_amd_comm *c);                    /* a rank leaves a multi-rank job out of band (shm transport: the others' barriers fail) */
int nla_comm_agree_ready(nlopt_amd_comm *c, int ok);      /* end of a multi-rank set-up: 1 iff every rank is ready */
int nla_comm_agree_same(nlopt_amd_comm *c, int ok, uint64_t fingerprint);   /* 1 ready and identical jobs, 0 some rank not ready, -1 ranks were given different jobs */
uint64_t nla_params_fingerprint(const nlopt_opt opt);             /* every nlopt_set_param value of opt (the options that shape the pass structure / the collectives' sizes) */
uint64_t nla_problem_fingerprint(int algorithm, int n, int population, int obj, const double *lb, const double *ub, const double *x,
                                 const nla_stopping *stop);
#define NLA_MSG_RANKS_DIFFER "nlopt_amd: the ranks of this communicator were given different problems (dimension, population, bounds, starting point, stopping criteria or nlopt_srand seed): one job needs them identical"
const nla_stopping *nla_comm_agree_stop(nlopt_amd_comm *c, const nla_stopping *stop, nla_stopping *view, int *force_store);

/* ---- MT19937 host side (mt_host.c) ------------------------------------------------------------ */
void nla_mt_seed_array(uint32_t mt[NLA_MT_N], unsigned long seed);
void nla_mt_regen(uint32_t mt[NLA_MT_N]);
uint32_t nla_mt_temper(uint32_t y);
uint32_t nla_genrand_int32(void);
double nlopt_urand(double a, double b);
int nlopt_iurand(int n);
double nlopt_nrand(double mean, double stddev);
void nla_srand_time_default(void);
void nla_mt_export(uint32_t mt[NLA_MT_N], int *consumed);
void nla_mt_import(const uint32_t mt[NLA_MT_N], int consumed);
int nla_mt_charpoly_terms(const int **exps);
void nla_mt_jump_poly_words(uint64_t J, uint64_t g[NLA_MT_POLYWORDS]);
const uint64_t *nla_mt_jump_poly_pow2(int k);
void nla_mt_apply_jump_host(const uint64_t g[NLA_MT_POLYWORDS], const uint32_t src[NLA_MT_N], uint32_t dst[NLA_MT_N]);
void nla_mt_advance_blocks_host(const uint32_t src[NLA_MT_N], uint64_t regens, uint32_t dst[NLA_MT_N]);

/* ---- device word stream (mtstream.c): the host generator continued on the GPU ---------------- */
typedef struct nla_mtstream nla_mtstream;
nla_mtstream *nla_mtstream_create(void *stream);     /* snapshots the calling thread's generator */
nla_mtstream *nla_mtstream_create_seg(void *stream, int seg_regens);   /* ... cut into segments of seg_regens regenerations (power of two <= NLA_MT_SEG_REGENS); fill only */
void nla_mtstream_destroy(nla_mtstream *s);
void nla_mtstream_host_state(nla_mtstream *s, uint64_t rel_word, uint32_t mt[NLA_MT_N], int *pos);
uint64_t nla_mtstream_origin(const nla_mtstream *s); /* global index of the first unconsumed word */
/* out[i] = stream word (origin + rel_first + i), i < count; device pointer, async on `stream` */
int nla_mtstream_fill(nla_mtstream *s, uint64_t rel_first, uint64_t count, uint32_t *d_out);
int nla_mtstream_rankbits(nla_mtstream *s, uint64_t rel_rank0, uint64_t rel_first, uint64_t count, int64_t popm1, int64_t rowwords, uint64_t *d_bits);
int nla_mtstream_rankbits_gated(nla_mtstream *s, uint64_t rel_rank0, uint64_t rel_first, uint64_t count, int64_t popm1, int64_t rowwords,
                                uint64_t *d_bits, int *d_gate, int *d_ticket, int waves_per_cu);   /* in-order gates: nla_k_mt_rankbits_gated */
int nla_mtstream_reserve(nla_mtstream *s, uint64_t rel_last);
/* room for the segment states of `words` words of stream from the start (within 2^17 states = 330 MB): the array doubles as the run
 * consumes its stream, and every doubling frees the old block — a device-wide wait in the middle of whatever runs (ISRES config 3: 3900
 * states per generation, 1.8 ms per free: host-side API trace, round 5) */
int nla_mtstream_expect(nla_mtstream *s, uint64_t words);
/* leave the calling thread's generator as if it had drawn `consumed` words since create */
int nla_mtstream_finish(nla_mtstream *s, uint64_t consumed);

/* ---- CRS engine interface (the device work the algorithm driver asks for) --------------------
 * The driver (crs_driver.c) owns the algorithm: ordered set, accept/reject chain, stopping.
 * The engine owns device memory and kernels.  A *slot* is one 2n-word stream block speculated as a
 * reflection trial; it is named by its block index.  `kind`: 1 = reflection trial, 2 = mutation. */
typedef struct {
    /* population rows 1..N-1 from the stream (2n words each), row 0 = x0; F[0..N-1] on the host.
     * obj < 0 (host-callback mode): rows generated but not evaluated. */
    int (*init_population)(void *e, const double *x0, double *F);
    /* how many consecutive blocks starting at `first_block` one window may hold */
    int (*max_slots)(void *e, uint64_t first_block);
    /* one pass over the window of K blocks first_block .. first_block+K-1 (see nla_k_crs_advance):
     * blocks >= fresh_from have no state yet (start at pick 0); every slot advances its gather-sum
     * to its first pick among W[0..d), d = its distance from the window front, W = the nW rows
     * that may be overwritten next, worst first.  status[a] = (f of the finished trial, f of the
     * mutation that would follow its rejection (w from block+1), picks summed so far). */
    int (*advance)(void *e, uint64_t first_block, int K, uint64_t fresh_from, int64_t i0, const int64_t *W, int nW,
                   nla_crs_slot_status *status);
    /* the same window in ONE launch with the accept/reject chain resolved on the device (hip/crs_chain.hip; K, nW <= 256):
     * every slot is computed from pick 0 and finishes; a pick of row W[j], j < a, is read as the device's resolution of the
     * chain says the row stands at the slot's turn.  Wf = f of the rows W, f_best = f of row i0.  fwcnt[a] / fwrec[a * fwcap ..]:
     * one record per such pick, j | producer window slot << 8 | kind << 16 (kind 0: the row itself; 1 / 2: the producer's trial
     * point / mutation), fwcnt[a] > fwcap: more than recorded.  NULL where the engine has no such kernel. */
    int (*chain)(void *e, uint64_t first_block, int K, int64_t i0, double f_best, const int64_t *W, const double *Wf, int nW,
                 nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap);
    /* forget the state of one in-flight block: it is recomputed from pick 0 by the next pass that holds it */
    int (*reset_slot)(void *e, uint64_t block);
    /* X[row[c]] := point of slot block[c] (kind[c]: 1 trial, 2 its mutation); rows distinct */
    int (*commit)(void *e, int ncommit, const uint64_t *block, const int32_t *kind, const int64_t *row);
    int (*read_slot)(void *e, uint64_t block, int kind, double *x);
    int (*read_row)(void *e, int64_t row, double *x);
    /* host-callback mode: mutate the slot's finished trial in place with the words of block+1 */
    int (*mutate_slot)(void *e, uint64_t block, int64_t i0);
    const char *(*last_error)(void *e);
    /* collective passes (a column-sharded run): what this rank sees of the per-process stop conditions goes INTO the next pass
     * (force_stop raised, clock run out) and comes back OR-ed over all ranks with that pass's status — the ranks' agreement rides
     * on the candidates' all-gather instead of costing a collective of its own.  NULL where passes are not collective. */
    void (*stop_flags_in)(void *e, int forced, int timed);
    void (*stop_flags_out)(void *e, int *forced, int *timed);
    /* fn(arg) is called by advance / chain after the pass has been handed to the device and before the host starts waiting for it:
     * host work that does not depend on the pass's results runs beside it (crs_driver.c: the ordered set's upkeep).  NULL: no such call. */
    void (*set_idle)(void *e, void (*fn)(void *arg), void *arg);
} nla_crs_engine_ops;

typedef struct {
    int n; int64_t N;
    const double *lb, *ub;
    int obj;                        /* device objective id, or -1: call f on the host */
    nlopt_func f; void *f_data;
    nla_stopping *stop;
    nlopt_amd_trace_rec *trace; size_t trace_cap, *trace_len;
    nlopt_amd_stats *stats;
    int max_spec;                   /* cap on slots per round (0 = default) */
    int forward;                    /* 1: windows through the engine's chain op (device-resolved dependences), 0: the conservative passes */
    double window_factor;           /* window = factor x (blocks consumed per pass, smoothed) + 4 (0 = default 1.5) */
    nlopt_amd_comm *comm;           /* non-NULL: the engine's passes are collective (column-sharded run) — the per-process stop conditions
                                     * (clock, force_stop) are then agreed by all ranks once per pass (nla_comm_agree_stop) */
} nla_crs_problem;

/* the algorithm, resumable between speculation rounds (bench steps, sessions);
 * *words_used = stream words consumed (2n per row / block) */
typedef struct nla_crs_session nla_crs_session;
nla_crs_session *nla_crs_begin(const nla_crs_engine_ops *ops, void *engine, const nla_crs_problem *pb,
                               double *x, double *minf, nlopt_result *ret_out);
nlopt_result nla_crs_advance(nla_crs_session *S, int64_t eval_budget);
nlopt_result nla_crs_end(nla_crs_session *S, uint64_t *words_used);
nlopt_result nla_crs_run(const nla_crs_engine_ops *ops, void *engine, const nla_crs_problem *pb,
                         double *x, double *minf, uint64_t *words_used);

/* reference-shaped entry (src/algs/isres/isres.h:34-41) */
int nla_isres_constraints_on_device(unsigned m, const nla_constraint *fc, unsigned p, const nla_constraint *h);
nlopt_result nla_isres_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, int m, nla_constraint *fc, int p, nla_constraint *h,
                                const double *lb, const double *ub, double *x, double *minf, nla_stopping *stop, int population);

/* LD_LBFGS (lbfgs_driver.c) */
int nla_lbfgs_default_mf(int n, int mf, int maxeval);
typedef struct nla_local_ctx nla_local_ctx;
nla_local_ctx *nla_local_ctx_create(const nla_evaluator *ev, int n, int cap, int mf, const double *d_lb, const double *d_ub, void *stream);
nla_local_ctx *nla_local_ctx_create_mma(const nla_evaluator *ev, int n, int cap, const nla_mma_params *alg_params, const double *d_sigma_init,
                                        const double *d_lb, const double *d_ub, void *stream);
nla_local_ctx *nla_local_ctx_create_cobyla(const nla_evaluator *ev, int n, int cap, const double *d_dx, const double *d_lb, const double *d_ub, void *stream);
void nla_local_ctx_set_cobyla_min_batch(nla_local_ctx *c, int min_batch);
int nla_local_ctx_alg(const nla_local_ctx *c);
void nla_local_ctx_destroy(nla_local_ctx *c);
double *nla_local_ctx_X(nla_local_ctx *c);
void nla_local_ctx_set_stats(nla_local_ctx *c, nlopt_amd_stats *stats);
void nla_local_ctx_after_launch(nla_local_ctx *c, int (*fn)(void *), void *arg);
int nla_local_ctx_count_finished(nla_local_ctx *c);
const int32_t *nla_local_ctx_finished_counter(nla_local_ctx *c, int32_t *from);
int nla_local_ctx_set_options(nla_local_ctx *c, int exact, const double *xtol_abs, const double *x_weights);
int nla_local_ctx_run(nla_local_ctx *c, int count, const nla_lbfgs_params *prm, nla_lbfgs_result *h_res, const nla_stopping *stop,
                      int *live_nevals);
int nla_local_run_batch(int alg, const nla_evaluator *ev, int n, int count, const double *lb, const double *ub, double *h_X, int mf,
                        const nla_mma_params *mma, const double *sigma_init, const nla_lbfgs_params *prm, nla_lbfgs_result *res,
                        const nla_stopping *stop, int exact, int *live_nevals, nlopt_opt trace_to, char *err, size_t errlen);
int nla_local_ctx_set_ftrace(nla_local_ctx *c, int64_t cap);
int nla_local_ctx_read_ftrace(nla_local_ctx *c, int inst, int64_t count, double *h_out);
/* LD_MMA without nonlinear constraints (mma_driver.c) */
int nla_mma_read_params(nlopt_opt opt, nla_mma_params *out);     /* optimize.c:798-815; 0 or an nlopt_result < 0 with errmsg set */
/* NLOPT_LN_COBYLA on the host (cobyla_host.c; reference entry cobyla_minimize, cobyla.c:181-271) */
nlopt_result nla_mma_constrained(nlopt_opt opt, unsigned n, nlopt_func f, void *f_data, const double *lb, const double *ub,
                                 double *x, double *minf, nla_stopping *stop, const nla_mma_params *prm);   /* mma_host.c */
nlopt_result nla_auglag_minimize(unsigned n, nlopt_func f, void *f_data, unsigned m, const nla_constraint *fc, unsigned p, const nla_constraint *h,
                                 const double *lb, const double *ub, double *x, double *minf, nla_stopping *stop, nlopt_opt sub_opt, int sub_has_fc);   /* auglag_host.c */
nlopt_result nla_cobyla_minimize(unsigned n, nlopt_func f, void *f_data, unsigned m, const nla_constraint *fc, unsigned p, const nla_constraint *h,
                                 const double *lb, const double *ub, double *x, double *minf, nla_stopping *stop, const double *dx);
nlopt_result nla_mma_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x,
                              double *minf, nla_stopping *stop);
nlopt_result nla_lbfgs_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                                nla_stopping *stop, int mf, double tolg);

/* Sobol LDS (sobol.c; src/util/sobolseq.c) */
int nla_sobol_directions(unsigned sdim, uint32_t *V);      /* V: 32*sdim u32; 0 = no generator for this dimension */
uint32_t nla_sobol_skip_count(unsigned n);
void nla_sobol_point01(unsigned sdim, const uint32_t *V, uint32_t index, double *x);

/* reference-shaped entry (src/algs/mlsl/mlsl.h:34-41) */
nlopt_result nla_mlsl_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x,
                               double *minf, nla_stopping *stop, nlopt_opt local_opt, int Nsamples, int lds);

/* reference-shaped entry (src/algs/esch/esch.h) */
nlopt_result nla_esch_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, double *x, double *minf,
                               nla_stopping *stop, unsigned np, unsigned no);

/* HIP engine (crs_engine.c) */
typedef struct nla_crs_hip_engine nla_crs_hip_engine;
nla_crs_hip_engine *nla_crs_hip_engine_create(int n, int64_t N, const double *lb, const double *ub, int obj, int forward,
                                              nlopt_amd_comm *comm, int shard, int cu_parts, nlopt_amd_stats *stats, char **errmsg);
void nla_crs_hip_engine_destroy(nla_crs_hip_engine *e, uint64_t words_used);
int nla_crs_can_shard(int n, int world);
int nla_crs_can_shard_windows(int n, int world);
extern const nla_crs_engine_ops nla_crs_hip_ops;

/* reference-shaped entry (src/algs/crs/crs.h:34-40) */
nlopt_result nla_crs_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub,
                              double *x, double *minf, nla_stopping *stop, int population);

#ifdef __cplusplus
}
#endif
#endif
