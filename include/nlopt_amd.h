/* nlopt_amd.h — additive extension of the NLopt C API plus the kernel-level C-ABI of libnlopt_amd.
 *
 * Part 1 (user-facing): the device-objective registry the reference cannot have (its objective
 *   is a host callback, src/api/nlopt.h:60-62; SURVEY.md §8b "required extension"), run
 *   statistics and the per-evaluation trace used by the parity tests.  All symbols are new
 *   (nlopt_amd_*), none changes the reference ABI.
 *
 * Part 2 (nla_k_*): one `extern "C"` launcher per HIP kernel.  Plain pointers and sizes only;
 *   every pointer is a DEVICE pointer unless its name starts with h_; `stream` is a hipStream_t
 *   passed as void*; return value 0 = launched, otherwise the hipError_t.  These are what a
 *   maintainer of the reference would bind to replace the CPU loops listed in SURVEY.md §2.3
 *   (each launcher cites the loop it replaces).  The parity tests call through them.
 */
#ifndef NLOPT_AMD_H
#define NLOPT_AMD_H

#include <stddef.h>
#include <stdint.h>
#include "nlopt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part 1 — extension API
 * ---------------------------------------------------------------------------------------------- */
#define NLOPT_AMD_OBJ_RASTRIGIN  0
#define NLOPT_AMD_OBJ_ACKLEY     1
#define NLOPT_AMD_OBJ_GRIEWANK   2   /* test/testfuncs.c:250-266 */
#define NLOPT_AMD_OBJ_ROSENBROCK 3   /* test/testfuncs.c:124-140, n-general */
#define NLOPT_AMD_OBJ_LEVY       4   /* test/testfuncs.c:218-240 */
#define NLOPT_AMD_OBJ_SPHERE     5
#define NLOPT_AMD_OBJ_COUNT      6

/* Host callback (sequential, same formulae) for device objective `id`; passing exactly this
 * pointer to nlopt_set_min/max_objective selects the HIP evaluator (pointer identity).  Under nlopt_set_max_objective the local
 * optimisers and MLSL keep the device evaluator (f and gradient are negated on the device, as the reference's f_max wrapper
 * does on the host, optimize.c:970-980), and so do CRS2_LM / ISRES / ESCH (their evaluation kernels multiply f by -1: the flag
 * NLA_OBJ_NEGATE).  Where part of the run has to call f on the HOST — fixed coordinates (the elimination wrapper), the parameter
 * amd_host_eval, an ISRES constraint that is not a device constraint, LD_MMA with nonlinear constraints — the whole run goes through
 * the reference's f_max wrapper instead, i.e. takes the exact host-callback path (api_optimize.c: objective_stays_on_device). */
nlopt_func nlopt_amd_objective(int id);
int nlopt_amd_objective_id(nlopt_func f);              /* -1: not a device objective */
const char *nlopt_amd_objective_name(int id);
void nlopt_amd_objective_box(int id, double *lo, double *hi);

/* block-sum inequality constraint g(x) = sum_{i in block q of Q} x_i - 1 <= 0 (ISRES config,
 * SURVEY.md §8d); func_data must point at `unsigned qQ[2] = {q, Q}` that outlives the run. */
nlopt_func nlopt_amd_constraint_blocksum(void);
int nlopt_amd_constraint_id(nlopt_func f);             /* 0: block sum, -1: not a device constraint */

int nlopt_amd_device_count(void);                      /* visible HIP devices (0 => optimize fails loudly) */

/* ---- user-supplied device objectives (required extension (ii): an additive setter) -----------------
 * code_object: path of a gfx950 code object built from a source that describes the objective with
 * include/nlopt_amd_device.h (NLOPT_AMD_DEVICE_OBJECTIVE(name, ...)); `name` selects the kernel <name>_evalgrad.
 * host_twin (may be NULL): the same function as an ordinary nlopt_func; it is what nlopt_get_... style introspection and
 * code outside the device paths see (e.g. the fixed-coordinate wrapper, optimize.c:219-445); when NULL the library
 * evaluates single points through the device kernel instead.  The objective then runs on the device inside every algorithm of this
 * library: population / sample evaluation in one launch, local searches as device coroutines with one launch of the
 * user's kernel per step of a whole batch.  nlopt_set_min/max_objective afterwards unbinds it.  Returns NLOPT_SUCCESS,
 * NLOPT_INVALID_ARGS (errmsg: the file / symbol that could not be loaded, or an ABI mismatch) or NLOPT_FAILURE (no device). */
nlopt_result nlopt_amd_set_min_device_objective(nlopt_opt opt, const char *code_object, const char *name, nlopt_func host_twin, void *f_data);
nlopt_result nlopt_amd_set_max_device_objective(nlopt_opt opt, const char *code_object, const char *name, nlopt_func host_twin, void *f_data);
int nlopt_amd_has_device_objective(const nlopt_opt opt);   /* 1: a user objective is bound, 0: not */

/* per-evaluation trace (same record as the oracle's): kind 0 = initial row, 1 = reflection trial,
 * 2 = local mutation; row = row written (init / accepted) or -1. */
typedef struct { double f; int64_t row; int32_t kind; int32_t accepted; } nlopt_amd_trace_rec;
nlopt_result nlopt_amd_set_trace(nlopt_opt opt, nlopt_amd_trace_rec *buf, size_t cap);
size_t nlopt_amd_trace_len(const nlopt_opt opt);       /* records produced by the last nlopt_optimize */

typedef struct {
    uint64_t rounds;            /* advance/commit passes */
    uint64_t slots_launched;    /* stream blocks started on the device as reflection trials */
    uint64_t slots_used;        /* ... consumed by the in-order commit walk */
    uint64_t slots_invalid;     /* device-resolved windows: slots recomputed because a row taken from a producer slot was not what the chain wrote */
    uint64_t slots_newbest;     /* ... discarded in flight: the best point changed */
    uint64_t slots_role;        /* ... discarded in flight: their block was consumed as a mutation block */
    uint64_t evals_init, evals_trial, evals_mutation;
    uint64_t accepted;
    uint64_t mt_words;          /* MT19937 words consumed (== the reference's count) */
    double t_init_s, t_trial_s; /* wall seconds in the two phases */
    double t_gather_ms;         /* sum of device time of the gather-sum (advance) kernel (HIP events) */
    uint64_t gather_launches;
    uint64_t gather_bytes;      /* algorithmic bytes those launches summed: 8n per row, n+1 rows per trial */
    /* ISRES (evals are counted in evals_trial) */
    uint64_t generations, rank_sweeps;
    double t_eval_s, t_rank_s, t_evolve_s, t_rng_s;   /* wall seconds per phase; t_rng_s = stream-word generation inside rank/evolve */
    /* batched L-BFGS inside MLSL: launches of lbfgs_batch_kernel, their device time (HIP events) and the algorithmic
     * bytes they streamed: 32 n per history column used (two matrices, a dot and an axpy pass each) + 16 n per f/grad evaluation */
    uint64_t lbfgs_launches, lbfgs_bytes;
    double t_lbfgs_ms;
    /* ISRES stochastic ranking (isres_stochrank_kernel): device time (HIP events), launches, and the serial ticks of the
     * systolic pipeline those launches had to make: pop + 2 sweeps + 63 ceil(sweeps / 64) each */
    double t_stochrank_ms;
    uint64_t stochrank_launches, stochrank_ticks;
    /* multi-rank CRS2_LM initialisation: device time (HIP events) of the all-gather of the rows and their f, and the bytes
     * every rank received (0 on a single rank) */
    double t_allgather_ms;
    uint64_t allgather_bytes;
    /* ISRES evolve (hip/isres_evolve2.hip): rounds of stage / scan / chain / write enqueued, and those of them that had
     * individuals left to resolve (a batch of rounds is enqueued before the host looks at the state again) */
    uint64_t evolve_rounds_enqueued, evolve_rounds;
    /* CRS2_LM trial phase, where the wall time of a pass / window goes on the HOST's side: inside the engine call (uploads, launches,
     * waiting for the device, copies out of pinned memory) and in the driver's in-order walk over the finished slots (verification of
     * the device's chain, the order statistics, stop tests, staging of the commits).  t_trial_s - the two = preparing the pass. */
    double t_engine_s, t_walk_s;
    /* ... and the driver's ordered set: how often the heap was brought up to date and the list of worst rows redrawn from it, how many
     * of those times that happened beside the device (inside the engine call, between launch and wait) rather than between two launches,
     * and the time it took (seconds; the part beside the device is inside t_engine_s, the rest outside both) */
    uint64_t list_refreshes, list_refreshes_beside_device;
    double t_list_refresh_s;
    /* ISRES gated ranking: launches of the pipeline in which a unit gave up waiting for its block of ranking bits (hip/isres_stochrank.h:
     * 4 s) — the ranking was then redone without gates after the generator's stream had been waited for.  0 in a healthy run */
    uint64_t isres_gate_timeouts;
    /* MLSL: iterations whose sampling phase (points, values, distances) had been computed beside the local phase before them
     * (mlsl_driver.c, mlsl_enqueue_ahead): every iteration but the first of a one-rank run with a compiled-in objective */
    uint64_t mlsl_sampled_ahead;
    uint64_t cobyla_host_searches;   /* GN_MLSL + LN_COBYLA on a device objective: searches of batches too small for the device, run by the host algorithm */
    /* batched L-BFGS: sum over its launches of the LONGEST search's ordered-sum steps, n (2 cols + 4 nevals) — two dot products per history
     * column used, about four passes of ordered sums per evaluation (objective, g.s; per iteration the norms, p, the stop test's) —
     * what a launch in the reference's summation order ("amd_exact_dot" = 1) lasts: one dependent addition per step (bench.py's model of that mode) */
    uint64_t lbfgs_longest_chain_steps;
    /* MLSL: gates in front of the sampling phase enqueued ahead that gave up waiting (50 ms) for the searches they start behind — the two
     * streams did not run beside each other; the run stops gating after the first.  0 in a healthy run */
    uint64_t mlsl_gate_timeouts;
} nlopt_amd_stats;
nlopt_result nlopt_amd_get_stats(const nlopt_opt opt, nlopt_amd_stats *out);
/* ... for a client that may have been built against an older or newer header: writes at most `bytes` bytes of the structure (fields are
 * only ever appended) and returns the library's sizeof(nlopt_amd_stats); 0 = invalid arguments */
size_t nlopt_amd_get_stats_sized(const nlopt_opt opt, void *out, size_t bytes);

/* generation hook: called on the caller's thread by ISRES at the start of every generation (isres.c:130) and by
 * MLSL at the start of every iteration (mlsl.c:345) — the device is idle at those points — with the number of
 * generations completed and the evaluations made so far.  bench.py times exactly K generations with it. */
typedef void (*nlopt_amd_progress_fn)(void *data, long generations_done, long numevals);
nlopt_result nlopt_amd_set_progress(nlopt_opt opt, nlopt_amd_progress_fn fn, void *data);

/* ---- multi-GPU (one process per GPU): the collective the sharded runs use --------------------------
 * Every rank creates the same optimiser, seeds the same nlopt_srand() and calls nlopt_optimize() with
 * the same arguments; the library partitions the data-parallel device work of each step over the
 * ranks and exchanges results with an ALL-GATHER: ISRES — the candidates' f/penalty;
 * MLSL — the minimisers found by the local searches dealt to each rank; CRS2_LM — the rows of the
 * initial population.  Every rank returns the same x, f and nlopt_result.  Without a communicator
 * (default) the run is single-process.
 *   RCCL transport: the id comes from nlopt_amd_rccl_unique_id() on rank 0 and reaches the other ranks
 *   through the launcher (torch.distributed / MPI broadcast, a file, ...); create after hipSetDevice.
 *   Host transport: an all-gather on host buffers supplied by the caller (MPI_Allgather, gloo, ...):
 *   h_recv[r*bytes ..] := rank r's h_send[0..bytes), return 0 on success. */
typedef struct nlopt_amd_comm_s nlopt_amd_comm;
typedef int (*nlopt_amd_allgather_fn)(void *ctx, const void *h_send, void *h_recv, size_t bytes);
int nlopt_amd_rccl_unique_id(void *id128);                                   /* 0 = ok */
nlopt_amd_comm *nlopt_amd_comm_create_rccl(int rank, int world, const void *id128);
nlopt_amd_comm *nlopt_amd_comm_create_host(int rank, int world, nlopt_amd_allgather_fn fn, void *ctx);
/* ranks on ONE node without a collective library: a POSIX shared-memory segment `name` ("/...", the same on every rank, unique to the
 * job) with per-rank slots of slot_bytes (0: 16 MiB; larger contributions travel in pieces) and a barrier; device data are copied
 * straight into / out of the (registered) slots.  Rank 0 creates the segment (removing one a crashed run left under the name), the
 * others wait for it and attach only to a segment whose creating process is alive — the ranks may start in any order. */
nlopt_amd_comm *nlopt_amd_comm_create_shm(int rank, int world, const char *name, size_t slot_bytes);
/* shm transport: how long a rank waits in a barrier for the others before it reports an error (default 600 s) */
void nlopt_amd_comm_set_timeout(nlopt_amd_comm *c, double seconds);
void nlopt_amd_comm_destroy(nlopt_amd_comm *c);
int nlopt_amd_comm_rank(const nlopt_amd_comm *c);
int nlopt_amd_comm_world(const nlopt_amd_comm *c);
const char *nlopt_amd_comm_error(const nlopt_amd_comm *c);
void nlopt_amd_comm_counters(const nlopt_amd_comm *c, uint64_t *calls, uint64_t *bytes);   /* collectives issued / bytes gathered */
nlopt_result nlopt_amd_set_comm(nlopt_opt opt, nlopt_amd_comm *c);            /* borrowed, not copied; NULL = single process */

/* Stepwise CRS2_LM: the same run as nlopt_optimize(), paused between passes (used by
 * bench.py to time exactly K steps, and by callers that want to poll).  open() performs the
 * population initialisation (crs_init, crs.c:165-229); step() runs until the algorithm stops or at
 * least `eval_budget` more objective evaluations were made (<=0: until it stops); close() writes
 * the best point to the x/minf given to open(), advances the thread's RNG exactly as a full
 * nlopt_optimize() would have, and returns the final nlopt_result. */
typedef struct nlopt_amd_crs_session nlopt_amd_crs_session;
nlopt_amd_crs_session *nlopt_amd_crs_open(nlopt_opt opt, double *x, double *minf, nlopt_result *ret);
nlopt_result nlopt_amd_crs_step(nlopt_amd_crs_session *s, long eval_budget);
nlopt_result nlopt_amd_crs_close(nlopt_amd_crs_session *s);

/* ------------------------------------------------------------------------------------------------
 * Part 2 — kernel-level C-ABI (device pointers; see file header)
 * ---------------------------------------------------------------------------------------------- */
#define NLA_MT_N 624
#define NLA_MT_POLYWORDS 312           /* GF(2) polynomial of degree < 19937 as 64-bit words */
#define NLA_MT_SEG_REGENS 1024         /* one stream segment = 1024 regenerations (4096: the generator loses parallelism — ISRES config 3 549 k -> 438 k evals/s) */
#define NLA_MT_SEG_WORDS (624ULL * NLA_MT_SEG_REGENS)

/* replaces: the serial generator src/util/mt19937ar.c:102-120 advanced J words.
 * dst[i] (624 words) = block array `J` words after src[i], g = t^J mod phi (312 u64). */
int nla_k_mt_jump(const uint64_t *poly, const uint32_t *src_states, uint32_t *dst_states, int count, void *stream);

/* replaces: nlopt_genrand_int32, src/util/mt19937ar.c:97-131, called count times.
 * out[i] = tempered stream word g_first+i; seg_states + 624*s = block array at the start of
 * segment seg_first+s; the nseg segments must cover [g_first, g_first+count). */
int nla_k_mt_generate(const uint32_t *seg_states, uint64_t seg_first, int nseg,
                      uint64_t g_first, uint64_t count, uint32_t *out, void *stream);
/* the same for a stream cut into segments of seg_regens regenerations (a power of two <= NLA_MT_SEG_REGENS; seg_states / seg_first in
 * units of THAT segment length): more, shorter segments = more wavefronts for a caller that fills few words at a time */
int nla_k_mt_generate_seg(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_first, uint64_t count, uint32_t *out,
                          int seg_regens, void *stream);
/* replaces: the ranking's nlopt_urand calls (isres.c:210, mt19937ar.c:194-198) — words [g_first, g_first + count) of the stream
 * (global word indices; g_rank0 = the ranking's first word, (g_first - g_rank0) even) reduced on the fly to the bits u < 0.45:
 * step s = (g - g_rank0) / 2 sets bit (s % popm1) of row (s / popm1) of `bits` (rows of `rowwords` u64, ZEROED by the caller).
 * seg_states: block arrays of segments seg_first .. seg_first + nseg - 1 as for nla_k_mt_generate. */
int nla_k_mt_rankbits(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first,
                      uint64_t count, int64_t popm1, int64_t rowwords, uint64_t *bits, void *stream);
/* the same with IN-ORDER GATES: segments are claimed front first through *ticket (0 before the launch) and a wavefront that has produced
 * everything it owes to the block of sweeps 64 c .. 64 c + 63 adds 1 to gate[c] (zero before the launch; its stores landed first):
 * block c is complete when gate[c] == nla_rankbits_gate_target(g_rank0, popm1, nrows, c), nrows = the sweeps [g_first, g_first + count)
 * covers counted from g_rank0.  waves_per_cu > 0 bounds the wavefronts a CU holds at a time (the segments are then worked off front
 * to back in waves of that size and the first blocks complete early); 0: as many as fit. */
int nla_k_mt_rankbits_gated(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first, uint64_t count,
                            int64_t popm1, int64_t rowwords, uint64_t *bits, int *gate, int *ticket, int waves_per_cu, void *stream);
int nla_rankbits_gate_target(uint64_t g_rank0, int64_t popm1, int64_t nrows, int64_t c);

/* replaces: crs_init's row loop, src/algs/crs/crs.c:211-226 (K1+K2 of SURVEY.md §2.3).
 * rows row_first .. row_first+nrows-1 of X (leading dimension ld) := lb + (ub-lb)*res53(words),
 * row r uses words[(r-row_first)*2n ...]; F[r] := objective(X[r]). obj < 0: skip evaluation. */
int nla_k_crs_init_rows(int obj, int n, int ld, const double *lb, const double *ub, const uint32_t *words,
                        int64_t row_first, int64_t nrows, double *X, double *F, void *stream);

/* replaces: one objective call per candidate (crs.c:133,205,220; isres.c:138; mlsl.c:335,360).
 * F[c] = objective(P + c*ld), c < count; one wavefront per candidate. */
int nla_k_eval(int obj, int n, int ld, const double *P, int64_t count, double *F, void *stream);

/* replaces: the Vitter method-A chain of random_trial, crs.c:89-109 (K3), for nblocks
 * consecutive 2n-word blocks of the stream, one lane per block.  Outputs per block b:
 *   jn[b]          which pick is reflected (crs.c:72)
 *   pos[b*n + t]   t < n-1: position of pick t among the N-1 non-best rows ("reduced" index,
 *                  the actual row is r + (r >= i0)); pos[b*n + n-1] = reduced position from
 *                  which the last pick jumps
 *   last[b]        d = iurand(Nleft) of the last pick (crs.c:109), resolved against i0 at use. */
int nla_k_crs_vitter(int n, int64_t N, const uint32_t *words, int nblocks,
                     int32_t *jn, int32_t *pos, int32_t *last, void *stream);

/* status of one window slot after a pass (read back by the host's in-order commit walk) */
typedef struct { double fT, fM; int32_t t, pad; } nla_crs_slot_status;

/* replaces: the centroid/reflection gather-sum of random_trial, crs.c:69,101-120 (K4), as a
 * resumable operation on a window of K consecutive stream blocks first_block .. first_block+K-1.
 * Block b uses entry b % ring_blocks of jn_ring/pos_ring/last_ring (as written by
 * nla_k_crs_vitter) and slot q = b & slot_mask of TX.  Slot state: picks [0,t) are summed into
 * TX[q] (t == n: TX[q] is the finished trial point x, scaled and clamped; row order, one
 * accumulator per coordinate, no FMA: bit-identical to the reference's x).  Window slot a
 * (block first_block+a) resumes at t_in[a] and advances to the first pick whose row is among
 * W[0 .. min(a,nW)) — the rows that may be overwritten before the slot's turn — or to n;
 * t_out[a] = new t.  t_in and t_out must be different arrays.  variant 0 = automatic tiling. */
int nla_k_crs_advance(int n, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                      const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                      uint64_t first_block, int K, const int64_t *W, int nW,
                      const int32_t *t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                      double *TX, int variant, void *stream);

/* replaces: crs_trial's whole loop body (crs.c:125-156) for a window of K <= 256 consecutive stream blocks in ONE launch
 * (hip/crs_chain.hip): the gather-sum of every block speculated as a reflection trial, its evaluation, the mutation that would
 * follow its rejection (words of the next block) with its evaluation — and the accept/reject chain itself, resolved on the
 * device in block order against the nW worst rows W (worst first) and their f values Wf, so that a slot whose pick is one of
 * those rows reads what the chain says the row holds at the slot's turn: the point (TX or TM) of the block that overwrote it,
 * or the row as it is.  Every slot finishes.  Outputs: TX / TM / status[a] = (fT, fM, n) as nla_k_crs_advance + nla_k_crs_finish
 * would leave them; fwcnt[a] and fwrec[a * fwcap ..] = one record per pick of a row W[j], j < a:  j | producer slot << 8 |
 * kind << 16  (kind 0: the row itself was read; 1 / 2: TX / TM of window slot `producer`) — the caller verifies them against
 * the chain it replays (a stop, or a value landing among the worst rows twice, are not modelled on the device).
 * ctrl: nla_crs_chain_ctrl_bytes(K, nW) bytes of device memory, zero before the first launch (the launcher re-zeroes all of it
 * but its ticket counter); ticket_base = workgroups launched by earlier calls on this ctrl = sum of nla_crs_chain_tickets.
 * A slot behind a trial that became the new best point may come back with status.t = 0 (not computed: it started from the old best row).
 * w_on_host != 0 (nW <= 128): W / Wf are host arrays and travel as kernel arguments.  f_best = f of row i0.
 * TX, TM and ctrl MUST be nla_dev_malloc_uncached memory: the workgroups hand trial points to each other through them; and
 * ld % 16 == 0 with TX / TM 128-byte aligned (no cache line shared by two slots) — the launcher refuses anything else. */
size_t nla_crs_chain_ctrl_bytes(int K, int nW);
int nla_crs_chain_chunks(int n, int ld);
int nla_k_crs_chain(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                    const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                    uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                    const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                    nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, void *stream);
/* the same; ctrl_is_zero != 0 — the caller cleared the control block behind its ticket word on this stream already
 * (nla_k_crs_commit_zero: no fill operations in front of the window) */
int nla_k_crs_chain_lean(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                         const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                         uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                         const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                         nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero, void *stream);
/* ---- the same window on a COLUMN-SHARDED population, one rank per device (replaces crs.c:125-156 for a population
 * dealt over several GPUs).  Rank r holds columns [c0, c0 + nc) of every row (X: N x ld); its launch forms ITS columns of every slot's
 * trial point and stores them into the slot's row of EVERY rank's TX (rows of ldf >= n doubles; the peers' TX mapped with nla_ipc_open),
 * raises per-chunk flag words on every rank, and evaluates / resolves exactly as the single-device launch does once a slot's chunks
 * from all ranks are in: identical f, identical decisions on every rank, no collective call.
 *   nla_crs_chain_sh_chunks      chunks a rank with `ncols` columns contributes per slot (ncols even where n >= 128: pad column zero)
 *   nla_crs_chain_sh_table       host image (nla_crs_chain_sh_table_bytes) of the table the kernel reads: rank r's TX / flags / stop words
 *                                as mapped HERE for every r (own entries: the own buffers), the whole best row, the whole bounds;
 *                                flags: K_max x chunks_total u32, stop words: nla_crs_chain_sh_stop_bytes — both zero before the first
 *                                launch, in the same peer-mapped allocation kind as TX
 *   nla_k_crs_chain_sh           the launch; seq = 1, 2, ... identical on every rank; stopbits: bit 0 force_stop, bit 1 maxtime as this
 *                                rank sees them; status[K] = (any rank forced, any rank timed, a rank's chunks never arrived);
 *                                grid_cap >= 2: the launch's workgroups take ticket after ticket, at most that many resident (ranks that
 *                                SHARE a device — a test box — must all fit the chip at once); 0: one workgroup per ticket
 *   nla_k_crs_commit_sh          in front of it: accepted whole points -> rows of the slice, control block cleared, whole best row
 *                                refreshed from a slot (best_slot >= 0) */
int nla_crs_chain_sh_chunks(int n, int ncols);
size_t nla_crs_chain_sh_table_bytes(void);
size_t nla_crs_chain_sh_stop_bytes(void);
int nla_crs_chain_sh_table(void *host_image, int world, int rank, int c0, int ldf, int chunks_total, int chunk0, void *const *peerTX,
                           void *const *peerflags, void *const *peerstop, const double *xbest, const double *lbf, const double *ubf);
int nla_k_crs_chain_sh(int obj, int n, int ncols, int ld, int ldf, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                       const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                       uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                       const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                       nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero,
                       const void *table, uint32_t seq, uint32_t stopbits, int grid_cap, void *stream);
uint32_t nla_crs_chain_sh_tickets(int n, int ncols, int K, int grid_cap);     /* tickets one launch draws (ticket_base of the next) */
int nla_k_crs_commit_sh(int nc, int ld, int ldf, int c0, double *X, const double *TX, const double *TM, int ncommit,
                        const int32_t *h_slot, const int32_t *h_kind, const int64_t *h_row, void *zero, size_t zero_bytes,
                        int n, int best_slot, int best_kind, double *xbest, void *stream);
/* workgroups one launch draws tickets for: K * nla_crs_chain_chunks + 1 (the resolver wavefront's workgroup, hip/crs_chain_resolver.h:
 * the chain — crs.c:135-156, the decisions between evaluations — is advanced by one dedicated wavefront out of registers) */
uint32_t nla_crs_chain_tickets(int n, int ld, int K);

/* replaces: the evaluation of the trial (crs.c:133) and the local mutation + its evaluation
 * (crs.c:139-146, K5) for the slots completed by the preceding nla_k_crs_advance (same window):
 * fT_ring[q] = f(TX[q]); TM[q] = clamp(best*(1+w) - w*TX[q]), w from the words of block b+1
 * (words_ring entry (b+1) % ring_blocks); fM_ring[q] = f(TM[q]); status[a] = (fT, fM, t) of
 * window slot a for all a < K.  obj == -1: no evaluation / no mutation (host-callback mode); obj == -2: the mutation only (the
 * objective is a user-supplied kernel the caller runs on TX / TM afterwards). */
int nla_k_crs_finish(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                     const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                     const int32_t *t_in, const int32_t *t_out, int slot_mask,
                     const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                     nla_crs_slot_status *status, void *stream);

/* nla_k_crs_advance / nla_k_crs_finish / nla_k_crs_commit with their small per-pass lists (W, t_in, the commit list; at most 128
 * entries each) given as HOST arrays: the lists travel as kernel arguments, so a pass needs no host-to-device copy in front of it
 * (one dependent stream operation less).  Everything else as in the pointer forms. */
int nla_k_crs_advance_args(int n, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                           const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                           uint64_t first_block, int K, const int64_t *h_W, int nW,
                           const int32_t *h_t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                           double *TX, int variant, void *stream);
int nla_k_crs_finish_args(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                          const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                          const int32_t *h_t_in, const int32_t *t_out, int slot_mask,
                          const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                          nla_crs_slot_status *status, void *stream);
/* the same with a doorbell for a spinning host: status and bell in pinned host memory, bell_count a zeroed device word */
int nla_k_crs_finish_args_bell(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM,
                          const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                          const int32_t *h_t_in, const int32_t *t_out, int slot_mask,
                          const double *lb, const double *ub, double *fT_ring, double *fM_ring,
                          nla_crs_slot_status *status, uint32_t *bell_count, uint32_t *bell, uint32_t bell_seq, void *stream);
int nla_k_crs_commit_args(int n, int ld, double *X, const double *TX, const double *TM, int ncommit,
                          const int32_t *h_slot, const int32_t *h_kind, const int64_t *h_row, void *stream);
/* nla_k_crs_commit_args + nla_k_crs_advance_args in ONE launch: the ncommit (<= 16) staged commits — distinct rows, none of their
 * source slots among this pass's own K slots — are copied by extra workgroups at the end of the grid, and every read of such a row
 * by the gather (a pick, or the best row a fresh slot starts from) is forwarded to the slot it is being copied from */
int nla_k_crs_advance_commit_args(int n, int ld, double *X, int64_t i0, const int32_t *jn_ring,
                                  const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                                  uint64_t first_block, int K, const int64_t *h_W, int nW,
                                  const int32_t *h_t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                                  double *TX, const double *TM, int ncommit, const int32_t *h_slot, const int32_t *h_kind,
                                  const int64_t *h_row, int variant, void *stream);

/* ---- CRS2_LM over several GPUs: the population sharded by COORDINATE (hip/crs_shard.hip) -----------------------------------------
 * Rank r holds columns [r * colper, r * colper + nc) of every row (colper = ceil(n / world); local index i = global column c0 + i;
 * rows ld >= nc apart, pad columns zero).  The reference's trial point (crs.c:63-121), mutation (:139-146) and row replacement
 * (:153) are per coordinate, so each rank runs them on its slice unchanged; the candidates of a pass cross the ranks (all-gather
 * of their slices) and every rank evaluates the assembled points with the single-GPU reduction: identical f everywhere. */
/* nla_k_crs_advance on a column slice: n rows are summed, ncol <= ld coordinates of each are held (X, TX, lb, ub: the slice's) */
int nla_k_crs_advance_cols(int n, int ncol, int ld, const double *X, int64_t i0, const int32_t *jn_ring,
                           const int32_t *pos_ring, const int32_t *last_ring, uint32_t ring_blocks,
                           uint64_t first_block, int K, const int64_t *W, int nW,
                           const int32_t *t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub,
                           double *TX, int variant, void *stream);
/* replaces: crs.c:211-219 for the slice — X[row_first + r][i] = urand(lb[i], ub[i]) from the words of (row, column c0 + i), i < nc */
int nla_k_crs_sh_init_rows(int n, int c0, int nc, int ld, const double *lb, const double *ub, const uint32_t *words,
                           int64_t row_first, int64_t nrows, double *X, void *stream);
/* the mutation half of nla_k_crs_finish on the slice, for the window slots completed by the preceding advance: TM[q] = slice of the
 * mutation (crs.c:139-146), and both slices packed for the all-gather: SEND[(2a) * colper + i] = T_i, SEND[(2a+1) * colper + i] = M_i;
 * SEND[2 K colper], [+1] = this rank's stop flags (force_stop raised, clock run out): a rank's block is 2 K colper + 2 doubles */
int nla_k_crs_sh_mutate_pack(int n, int c0, int nc, int ld, int colper, const double *X, int64_t i0, const double *TX, double *TM,
                             const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *t_in,
                             const int32_t *t_out, int slot_mask, const double *lb, const double *ub, double *SEND,
                             int flag_forced, int flag_timed, void *stream);
/* the evaluation half (crs.c:133, :146) on the gathered candidates: RECV rank-major, each rank's block = 2K slices of colper doubles + 2;
 * fT_ring / fM_ring / status as nla_k_crs_finish writes them; status[K].fT / .fM = the stop flags OR-ed over the ranks */
int nla_k_crs_sh_eval(int obj, int n, int colper, uint64_t first_block, int K, const int32_t *t_in, const int32_t *t_out, int slot_mask,
                      const double *RECV, int world, double *fT_ring, double *fM_ring, nla_crs_slot_status *status, void *stream);

/* replaces: memcpy(worst->k, d->p, ...) at crs.c:153 for a batch of accepted candidates.
 * X[row[c]] := (kind[c] == 1 ? TX : TM)[slot[c]];  rows must be distinct within one call. */
int nla_k_crs_commit(int n, int ld, double *X, const double *TX, const double *TM, int ncommit,
                     const int32_t *slot, const int32_t *kind, const int64_t *row, void *stream);

/* nla_k_crs_commit (lists_on_host == 0) / nla_k_crs_commit_args (!= 0, ncommit <= 128) that also clear zero_bytes (a multiple of 4) at
 * `zero`: the control block of the window launched next on the stream.  ncommit >= 1. */
int nla_k_crs_commit_zero(int n, int ld, double *X, const double *TX, const double *TM, int ncommit, const int32_t *slot,
                          const int32_t *kind, const int64_t *row, int lists_on_host, void *zero, size_t zero_bytes, void *stream);

/* single local mutation of one candidate in place (host-callback mode, crs.c:139-146) */
int nla_k_crs_mutate(int n, const double *best, double *p, const uint32_t *words,
                     const double *lb, const double *ub, void *stream);

/* ---- ISRES (src/algs/isres/isres.c) --------------------------------------------------------------
 * X, S (step sizes): pop x ld fp64 row-major; F/PEN/GPEN: pop fp64; FEAS: pop i32. */

/* a constraint the device can evaluate; type 0 = block sum  sum_{i in block q of Q} x_i - 1 */
typedef struct { int32_t type; uint32_t q, Q; int32_t pad; double tol; } nla_dev_constraint;

/* replaces: the initial population, isres.c:122-128, for individuals k_first .. k_first+count-1
 * (xs k-major from the stream: `words` starts at individual k_first, coordinate j of individual
 * k uses words 2((k-k_first)n+j),+1; sigma = (ub-lb)/sqrt(n); individual 0 := x0). */
int nla_k_isres_init(int n, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t k_first,
                     int64_t count, const double *x0, double *X, double *S, void *stream);

/* replaces: f and the constraint penalties of every candidate, isres.c:138-166.  con[0..m) are the
 * inequality, con[m..m+p) the equality constraints.  PEN = sum max(g,0)^2 + sum h^2 in constraint
 * order, GPEN = the inequality part, FEAS = every g <= tol and |h| <= tol.  obj < 0: the constraint part only (F untouched: the
 * objective is a user-supplied kernel, nlopt_amd_set_min_device_objective). */
int nla_k_isres_eval(int obj, int n, int ld, const double *X, int64_t pop, int m, int p, const nla_dev_constraint *con,
                     double *F, double *PEN, double *GPEN, int32_t *FEAS, void *stream);

/* replaces: nlopt_qsort_r(irank, ..., fval, key_compare), isres.c:204 (sorted[pos] = individual,
 * stable), and prepares the stochastic ranking: elems[k] = individual k packed with the dense ranks
 * of its fval and penalty (layout in isres_kernels.hip).  pop <= 2^20. */
int nla_k_isres_rank_count(int64_t pop, const double *F, const double *PEN, uint64_t *elems, int32_t *sorted, void *stream);

/* replaces: u = nlopt_urand(0,1) of every ranking step, isres.c:210, reduced to the bit u < PF.
 * words = stream words of sweeps row_first .. row_first+nrows-1 (2(pop-1) words per sweep);
 * bits row i = ceil((pop-1)/64) u64, bit j = (u_{i,j} < 0.45). */
int nla_k_isres_bits(const uint32_t *words, int64_t row_first, int nrows, int64_t pop, uint64_t *bits, void *stream);

/* replaces: the stochastic ranking sweeps, isres.c:206-228, as a systolic pipeline of nsweeps
 * stages.  streams: (ceil(nsweeps/64)+1) x pop u64, streams[0..pop) = the packed elements in
 * initial order; progress: one int per stream (progress[0] = pop, others 0); *ticket = 0.
 * Out: swapped[i] = sweep i exchanged something; irank[pos] = individual. */
int nla_isres_stochrank_handoff(void);      /* elements a pipeline unit hands on at a time (ticks of a launch: pop + 2 sweeps + (handoff - 1) units) */
int nla_k_isres_stochrank(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits,
                          int *ticket, uint8_t *swapped, int32_t *irank, void *stream);
/* the same with the rows of `bits` still being produced by nla_k_mt_rankbits_gated on another stream: unit u (sweeps 64u .. 64u+63) waits
 * until gate[u] has reached nla_rankbits_gate_target(gate_g_rank0, pop - 1, gate_nrows, u) before it reads a row.  gate == NULL: no waiting. */
int nla_k_isres_stochrank_gated(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits, int *ticket,
                                uint8_t *swapped, int32_t *irank, const int *gate, uint64_t gate_g_rank0, int64_t gate_nrows, void *stream);

/* replaces: nlopt_nrand(0,1), mt19937ar.c:216-232, for a run of 4-word attempts: appends the
 * accepted deviates of attempts [attempt_base, attempt_base+nattempts) (words = their words) to
 * z[zbase ...], zatt = their attempt indices; *ztotal += number appended.  counts: scratch,
 * ceil(nattempts/1024) ints. */
int nla_k_isres_nrand(const uint32_t *words, int64_t nattempts, int64_t attempt_base, int32_t *counts, int64_t *ztotal,
                      int64_t zbase, double *z, int64_t *zatt, void *stream);

/* replaces: the standard mutation (phase 0, isres.c:234-252) / differential variation (phase 1,
 * :253-280) loops, consuming the deviates z[state[1] ...] in the reference's order.
 * state = {next individual, next deviate, ran-out flag}; scratch = 3*ld doubles that persist
 * between calls of one generation. */
int nla_k_isres_evolve(int n, int ld, int phase, int64_t pop, int64_t survivors, int64_t zcount, double taup, double tau,
                       const double *lb, const double *ub, const double *z, const int32_t *irank, double *X, double *S,
                       double *scratch, int64_t *state, void *stream);

/* the same two loops, parallel over individuals (hip/isres_evolve2.hip: stage / multi-start scan / chain look-up /
 * write, `rounds` times; each round resolves up to 256 consecutive individuals).  state as above plus [9] individuals
 * resolved by the last round, [10] = 1: the next individual needs the serial kernel (nla_k_isres_evolve with
 * state[14] = index to stop before), [14] serial stop index; rho: 4 doubles of running redraw statistics (zero at
 * first use, kept between generations); inv: inverse of irank (nla_k_isres_inverse); x0c: copy of X row 0 taken
 * before phase 1; ws: nla_isres_evolve2_ws_bytes(n) bytes.  n <= 1150 (nla_isres_evolve2_supported). */
size_t nla_isres_evolve2_ws_bytes(int n);
int nla_isres_evolve2_supported(int n);
int nla_k_isres_inverse(int64_t pop, const int32_t *irank, int32_t *inv, void *stream);
int nla_k_isres_evolve_rounds(int n, int ld, int phase, int64_t pop, int64_t survivors, int64_t zcount, double taup, double tau,
                              const double *lb, const double *ub, const double *z, const int32_t *irank, const int32_t *inv,
                              double *X, double *S, const double *x0c, int64_t *state, double *rho, void *ws, const double *mu_rp,
                              int rounds, void *stream);
/* once per generation, before the mutation phase's rounds (phase 0 of nla_k_isres_evolve_rounds needs it): mu_rp[p], p < survivors =
 * the redraws a child of the parent at rank position p is expected to make — what the rounds predict the children's starts with */
int nla_k_isres_evolve_parent_mu(int n, int ld, int64_t survivors, const double *lb, const double *ub, const int32_t *irank,
                                 const double *X, const double *S, double *mu_rp, void *stream);

/* ---- the batched local optimisers: common pieces ---------------------------------------------------- */
/* External evaluation: the objective of a local search is not one of the compiled-in device objectives but a host callback
 * (the reference's nlopt_func contract, src/api/nlopt.h:60-62: called on the caller's thread, one x at a time) or a
 * user-supplied device module.  With obj == NLA_OBJ_EXTERNAL the batch kernels run as coroutines: at every objective
 * evaluation of a search (plis.c:260,390; mma.c:219,297,337) the kernel writes the point into row `inst` of EX, sets
 * req[inst] = {1, gradient wanted?} and returns; the caller evaluates, stores f in EF[inst] and the gradient in row inst of
 * EG, and launches the same kernel again with resume = 1; req[inst].state == 2: the search has finished (out[inst] valid).
 * All pointers are device pointers. */
#define NLA_OBJ_EXTERNAL (-1)
/* OR-ed into a compiled-in objective id given to a kernel launcher (nla_k_eval, nla_k_crs_init_rows, nla_k_crs_finish*,
 * nla_k_crs_chain, nla_k_isres_eval): the kernel delivers -f — how nlopt_set_max_objective keeps a device objective on the
 * device (the run minimises -f, optimize.c:1014-1024; the dispatcher turns the result's sign back) */
#define NLA_OBJ_NEGATE 0x100
typedef struct { int32_t state, want_grad; } nla_local_req;
typedef struct {
    nla_local_req *req;            /* count entries */
    double *EX, *EG, *EF;          /* count x ld points; count x ld gradients; count values */
    void *save;                    /* count x nla_lbfgs_save_bytes() / nla_mma_save_bytes() */
    int32_t resume;                /* 0: start the searches; 1: continue those whose evaluation was delivered */
    int32_t forced, timeout;       /* the caller's nlopt_force_stop flag / maxtime verdict at this launch (stop.c:141-159) */
    int32_t pad;
} nla_local_ext;

/* ---- LD_LBFGS (src/algs/luksan/plis.c), batched ----------------------------------------------------- */
/* exact != 0: every dot product / norm / objective sum accumulated in the reference's sequential order (mssubs.c:601-641,
 * stop.c:37-57) instead of the workgroup tree — the iterates are then the reference's bit for bit; sign = -1: minimise -f
 * (the reference's maximisation wrapper, optimize.c:970-980), 0 = +1; xtol_abs / x_weights: device arrays of n or NULL
 * (nlopt_stop_dx, stop.c:110-120); abort: NULL, or a device-visible flag the kernel polls once per iteration:
 * 100 = maxtime reached (plis.c:263,273,371), -999 = forced stop (pssubs.c:914). */
typedef struct { double minf_max, ftol_rel, ftol_abs, xtol_rel, tolg; int32_t maxeval, exact; double sign;
                 const double *xtol_abs, *x_weights; const int32_t *abort;
                 double *ftrace; int64_t ftrace_cap;     /* NULL, or count x ftrace_cap doubles: f of every evaluation of search i, in order */
                 int32_t *done;                          /* NULL, or a device counter every search adds 1 to when it ends (agent-scope atomic; never reset: the
                                                            reader keeps the count it started from) — what nla_k_gate waits on, see nla_local_ctx_finished_counter */
               } nla_lbfgs_params;
typedef struct { double f; int32_t ret, nevals, iterm, cols; } nla_lbfgs_result;   /* cols = history columns streamed: sum over iterations of k */
size_t nla_lbfgs_work_doubles(int ld, int mf, int count);     /* doubles of `work` */
size_t nla_lbfgs_hist_doubles(int ld, int mf, int count);     /* doubles of `hist` */
size_t nla_lbfgs_save_bytes(void);                            /* per search, of nla_local_ext.save */

/* replaces: luksan_plis (plis.c:420-510) for `count` independent starts at once, one workgroup per
 * start, the whole optimisation loop on the device.  X: count x ld, start points in, minimisers
 * out; lb/ub: the box (n); mf: history pairs kept (plis.c:441-445 computed by the caller); work /
 * iwork (count*ld ints) / hist: scratch sized by the helpers above; out[i] = (f, nlopt_result,
 * evaluations, PLIS termination code) of start i.  obj: a compiled-in device objective, or NLA_OBJ_EXTERNAL with `ext`
 * (NULL otherwise). */
int nla_k_lbfgs_batch(int obj, int n, int ld, int mf, int count, const double *lb, const double *ub, double *X,
                      double *work, int *iwork, double *hist, const nla_lbfgs_params *params, nla_lbfgs_result *out,
                      const nla_local_ext *ext, void *stream);

/* ---- LD_MMA without nonlinear constraints (src/algs/mma/mma.c), batched; with constraints the outer algorithm runs on the
 * host (csrc/mma_host.c) and this kernel solves its dual problems ------------------------------------------------------ */
/* stopping values as nlopt_stopping holds them (nlopt-util.h:79-91) + the algorithm's parameters as the dispatcher reads
 * them (optimize.c:798-803: rho_init 1, sigma_min 0, inner_maxeval 0, inner_gradients 1, always_improve 1); exact, sign,
 * xtol_abs, x_weights (nlopt_stop_x, stop.c:98-108), abort (mma.c:258-260,394-396) as for LD_LBFGS */
typedef struct { double minf_max, ftol_rel, ftol_abs, xtol_rel, rho_init, sigma_min;
                 int32_t maxeval, inner_maxeval, inner_gradients, always_improve; int32_t exact, pad; double sign;
                 const double *xtol_abs, *x_weights; const int32_t *abort;
                 double *ftrace; int64_t ftrace_cap;     /* as for LD_LBFGS; indexed by objective calls (the uncounted call included) */
                 int32_t *done;                          /* as for LD_LBFGS */
               } nla_mma_params;
size_t nla_mma_work_doubles(int ld, int count);               /* doubles of `work` */
size_t nla_mma_save_bytes(void);

/* ---- LN_COBYLA with bound constraints only (src/algs/cobyla/cobyla.c), batched: the default local optimiser of NLOPT_GN_MLSL(_LDS)
 * (optimize.c:763-768; MLSL strips its local optimiser's nonlinear constraints, options.c:824-846) ------------------------------------ */
/* stopping values as nlopt_stopping holds them; exact / sign / xtol_abs / abort / done as for LD_LBFGS */
typedef struct { double minf_max, ftol_rel, ftol_abs, xtol_rel; int32_t maxeval, exact; double sign;
                 const double *xtol_abs; const int32_t *abort; int32_t *done; } nla_cobyla_params;
size_t nla_cobyla_work_doubles(int n, int ld, int count);      /* doubles of `work` (a search's state is in LDS: a token size) */
size_t nla_cobyla_work_ints(int n, int count);                 /* ints of `iwork` (the same) */
size_t nla_cobyla_lds_bytes(int n);                            /* LDS one search of n variables in a finite box takes (simplex, inverse, models, LP basis) */
int nla_cobyla_fits(int n);                                    /* 1: that fits a compute unit's 160 KB (n <= 51) — beyond, the caller runs the host algorithm */
/* replaces: cobyla_minimize (cobyla.c:181-271) as nlopt_optimize(LN_COBYLA) reaches it (optimize.c:836-851, with the memoized best
 * point of :450-508,1064-1071) for `count` independent starts at once, one WAVEFRONT per start, compiled-in device objectives only.
 * X: count x ld, starts in, results out; dx: the initial step (n, device) or NULL = nlopt_set_default_initial_step per start
 * (options.c:921-946); out[i] = (f, nlopt_result, objective calls, the same, 0).  Fails (hipErrorInvalidValue) when !nla_cobyla_fits(n). */
int nla_k_cobyla_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *dx, double *X,
                       double *work, int *iwork, const nla_cobyla_params *params, nla_lbfgs_result *out, void *stream);

/* replaces: mma_minimize (mma.c:146-449) with m = 0 for `count` independent starts at once, one workgroup per start, outer
 * and inner iterations on the device (the 0-dimensional dual "solve" is dual_func's closed form, mma.c:58-137).  X: count x
 * ld, starts in, minimisers out; sigma_init: the initial step (n, device) or NULL (mma.c:203-211); out[i] = (f,
 * nlopt_result, evaluations counted by the algorithm, iterm = objective calls made, cols = outer iterations). */
int nla_k_mma_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *sigma_init, double *X,
                    double *work, const nla_mma_params *params, nla_lbfgs_result *out, const nla_local_ext *ext, void *stream);

/* ---- MLSL (src/algs/mlsl/mlsl.c) ------------------------------------------------------------------- */
/* replaces: nlopt_sobol_next (sobolseq.c:236-242) for `count` consecutive points: row r of P (count x ld) := point number
 * index_first + r (1-based call count of the reference's generator) scaled to [lb, ub]; V = 32 x n direction numbers as
 * 32-bit fractions (nla_sobol_directions).  Bit-identical to the reference's stateful Gray-code walk. */
int nla_k_mlsl_sobol_rows(int n, int ld, const double *lb, const double *ub, const uint32_t *V, uint32_t index_first,
                          int count, double *P, void *stream);
/* replaces: distance2 (mlsl.c:118-127) for all pairs: D[i*nb + j] = |A_i - B_j|^2, A: na x ld, B: nb x ld
 * (summed over k ascending without FMA: bit-identical to the reference's loop). */
int nla_k_mlsl_dist2(int n, int ld, const double *A, int na, const double *B, int nb, double *D, void *stream);
/* replaces: find_closest_pt / find_closest_lm (mlsl.c:131-155): out[i] = min(init[i], min_j {D[i*ldd+j] : FB[j] < FA[i]}) */
int nla_k_mlsl_rowmin(const double *D, int ldd, int na, int nb, const double *FA, const double *FB, const double *init, double *out, void *stream);
/* replaces: pts_update_newpt (mlsl.c:162-173): inout[j] = min(inout[j], min_i {D[i][j] : FA[i] < FB[j]}) where skip[j] == 0 */
int nla_k_mlsl_colmin(const double *D, int ldd, int na, int nb, const double *FA, const double *FB, const int32_t *skip, double *inout, void *stream);
/* replaces: the per-point memcpy of a local search's start (mlsl.c:399-404) and of an accepted minimum (mlsl.c:410-414) for a
 * whole batch: dst row c := src row idx[c] (idx on the device) */
int nla_k_mlsl_gather_rows(int n, int ld, const double *src, const int64_t *idx, int count, double *dst, void *stream);
/* out[a * nc + b] = D[rows[a] * ldd + cols[b]] (a < nr, b < nc): the entries of a batch's distance matrix the commit walk reads */
int nla_k_mlsl_gather_pairs(const double *D, int ldd, const int64_t *rows, int nr, const int64_t *cols, int nc, double *out, void *stream);
int nla_k_mlsl_gather_pairs_t(const double *D, int ldd, const int64_t *rows, int nr, const int64_t *cols, int nc, double *out, void *stream);   /* transposed: out[b * nr + a] */
/* replaces: the bound test of is_potential_minimizer (mlsl.c:211-218) for `count` points: flags[c] = 1 if row idx[c] of P has a
 * coordinate within thr (= dbound R) of a bound of a box side wider than thr */
int nla_k_mlsl_near_bound(int n, int ld, const double *P, const int64_t *idx, int count, const double *lb, const double *ub,
                          double thr, int32_t *flags, void *stream);

/* replaces: the sign flip of the maximisation wrapper f_max (optimize.c:970-980) for `count` device-evaluated values */
int nla_k_mlsl_negate(double *F, int count, void *stream);

/* ---- ESCH (src/algs/esch/esch.c) -------------------------------------------------------------------- */
/* replaces: randcauchy (esch.c:28-50) called back to back: appends the accepted values (folded to [0,1], i.e. `valor` before the
 * scaling into [lb,ub]) of attempts [attempt_base, attempt_base + nattempts) (2 words each) to v[vbase ...) (capacity vcap),
 * vatt = their attempt indices, *vtotal += number accepted; counts: scratch, ceil(nattempts/1024) ints */
int nla_k_esch_cauchy(const uint32_t *words, int64_t nattempts, int64_t attempt_base, int32_t *counts, int64_t *vtotal,
                      int64_t vbase, int64_t vcap, double *v, int64_t *vatt, void *stream);
/* replaces: the initial population loops (esch.c:133-164): element e = id*n + item of R := lb + (ub - lb) v[e - e0] */
int nla_k_esch_fill_rows(int n, int ld, const double *lb, const double *ub, const double *v, int64_t e0, int64_t count, double *R, void *stream);
/* replaces: crossover (esch.c:192-203): offspring id (individual np + id) from words 3 id .. 3 id + 2; slot[i] = row of individual i */
int nla_k_esch_crossover(int n, int ld, int64_t np, int64_t no, const uint32_t *words, const int32_t *slot, double *R, void *stream);
/* replaces: the point-mutation loop (esch.c:207-218), `total` steps from the M words W (see hip/esch_kernels.hip); last: no*n ints,
 * scratch: nla_esch_mut_scratch_bytes(M); out (device, 2 x i64): steps the segment holds (< total: too short), words consumed */
size_t nla_esch_mut_scratch_bytes(int64_t M);
int nla_k_esch_mutate(const uint32_t *W, int64_t M, int64_t total, int n, int ld, int64_t np, int64_t no, const double *lb,
                      const double *ub, const int32_t *slot, double *R, int32_t *last, void *scratch, int64_t *out, void *stream);
/* G (count x ld) := rows of individuals i0 .. i0+count-1 */
int nla_k_esch_gather_rows(int n, int ld, const int32_t *slot, int64_t i0, int64_t count, const double *R, double *G, void *stream);
/* replaces: nlopt_qsort_r(estotal, ...) + the copy back (esch.c:243-251): stable sort of (slot, fit) by fit */
size_t nla_esch_sort_scratch_bytes(int64_t count);
int nla_k_esch_select(int64_t count, const int32_t *slot_in, const double *fit_in, int32_t *slot_out, double *fit_out,
                      void *scratch, size_t scratch_bytes, void *stream);

/* thin device-runtime layer the C host code uses (no HIP types cross the boundary) */
int nla_dev_count(void);
int nla_dev_set(int dev);
void *nla_dev_malloc(size_t bytes);
void nla_dev_free(void *p);
void *nla_dev_malloc_uncached(size_t bytes);    /* MTYPE UC device memory: coherent between workgroups / XCDs without cache maintenance */
void nla_debug_uncached_stats(long out[4]);     /* [0] uncached allocations made, [1] returned to the driver (never, with the pool), [2] pooled blocks, [3] in use */
void nla_dev_free_uncached(void *p);            /* back to the library's pool (per device; nothing returns to the driver while an uncached block of the device is in use; idle blocks above 1 GiB are trimmed when the last one is released — devrt.hip) */
size_t nlopt_amd_release_device_memory(void);   /* gives every idle pooled block back to the driver (between the large runs of a long-lived process); bytes released */
void *nla_host_malloc(size_t bytes);            /* pinned */
void nla_host_free(void *p);
int nla_host_register(void *p, size_t bytes);   /* caller-owned host memory made page-locked and device-visible (the shm transport's segment); 0 = ok */
void nla_host_unregister(void *p);
int nla_memcpy_h2d(void *dst, const void *h_src, size_t bytes, void *stream);
int nla_memcpy_d2h(void *h_dst, const void *src, size_t bytes, void *stream);
int nla_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream);
int nla_memset(void *dst, int value, size_t bytes, void *stream);
void *nla_stream_create(void);
void *nla_stream_create_cu_share(int part, int parts);   /* confined to the compute units i with i mod parts == part (hipExtStreamCreateWithCUMask): ranks sharing one device */
#define NLA_IPC_BYTES 96
int nla_ipc_export(const void *p, void *blob96);    /* device memory of this process made mappable by another (hipIpcGetMemHandle); 0 = ok */
void *nla_ipc_open(const void *blob96);             /* ... mapped here (hipIpcOpenMemHandle); NULL = failed */
void nla_ipc_close(void *p);
void *nla_stream_create_background(void);       /* lowest dispatch priority of the device (hipStreamCreateWithPriority): gap-filling work */
void nla_stream_destroy(void *stream);
int nla_stream_sync(void *stream);
int nla_stream_query(void *stream);             /* 0: all work done, -1: still running, otherwise the hipError_t */
void *nla_event_create(void);
void nla_event_destroy(void *ev);
int nla_event_record(void *ev, void *stream);
int nla_event_sync(void *ev);
float nla_event_elapsed_ms(void *ev0, void *ev1);
int nla_stream_wait_event(void *stream, void *ev);
/* one wavefront that holds the stream's following work back until *counter - from >= need (agent-scope loads; counter in
 * nla_dev_malloc_uncached memory) or timeout_ms have passed — how work enqueued BESIDE a batch of local searches starts when part of
 * them have finished (mlsl_driver.c).  The emulated device returns at once. */
int nla_k_gate(const int32_t *counter, int32_t from, int32_t need, double timeout_ms, int32_t *gave_up /* device-writable (pinned) word set to 1 when the gate times out; or NULL */, void *stream);
const char *nla_dev_error_string(int err);
/* code objects loaded at run time (user device objectives) */
void *nla_module_load_file(const char *path);
void *nla_module_load_data(const void *image);
void nla_module_unload(void *module);
void *nla_module_function(void *module, const char *name);
int nla_module_launch(void *function, unsigned grid_x, unsigned block_x, void **params, void *stream);   /* params[i] = &argument i */

#ifdef __cplusplus
}
#endif
#endif /* NLOPT_AMD_H */
