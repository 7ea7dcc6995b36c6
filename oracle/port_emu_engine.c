/* port_emu_engine.c — CPU ORACLE (test infrastructure): a CPU stand-in for the HIP engine, built
 * from the per-kernel statements of port_kernels.c, so that the product's *host-side* CRS logic
 * (nlopt_amd/csrc/crs_driver.c: ordered set, window of resumable slots, in-order commit, stopping
 * quirks) can be tested against the oracle and the real reference on machines without a GPU.
 * Lives in oracle/libemu.so, which links the product library only to call nla_crs_run(); the
 * product never links or loads this file, and nlopt_optimize() has no path to it. */
#include "port_oracle.h"
#include "../nlopt_amd/csrc/nla_internal.h"
#include "objfuncs.h"
#include <stdlib.h>
#include <string.h>

void orc_k_words(uint64_t count, uint32_t *out);
void orc_k_init_rows(int n, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t nrows, double *X);
void orc_k_eval(int obj, int n, int ld, const double *P, int64_t count, double *F);
void orc_k_vitter(int n, int64_t N, const uint32_t *words, int nblocks, int32_t *jn, int32_t *pos, int32_t *last);
void orc_k_gather(int n, int ld, const double *X, int64_t i0, const int32_t *jn, const int32_t *pos, const int32_t *last,
                  int K, const double *lb, const double *ub, double *TX);
void orc_k_mutate(int n, const double *best, const double *p, const uint32_t *words, const double *lb, const double *ub, double *out);
int orc_k_advance_slot(int n, int ld, const double *X, int64_t i0, int32_t jn, const int32_t *pos, int32_t last,
                       const int64_t *W, int nun, int t0, const double *lb, const double *ub, double *acc);

typedef struct { double fT, fM; int32_t t, pad; } orc_slot_status;
void orc_k_crs_chain(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring, const int32_t *pos_ring,
                     const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                     const int64_t *W, const double *Wf, int nW, int slot_mask, const double *lb, const double *ub, double *TX, double *TM,
                     orc_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, const orc_slot_status *decide_with, uint32_t *dbg);

#define ECAP 1024                           /* slot ring, block b at slot b % ECAP */

typedef struct {
    int n, ld, obj, cap; int64_t N;
    const double *lb, *ub;
    double *X, *TX, *TM, *fT, *fM;
    uint32_t *tw; uint64_t tw_blocks;       /* trial-phase words generated so far (blocks of 2n) */
    int32_t *jn, *pos, *last, *t;           /* per ring slot */
    int max_slots;
} emu;

static void need_blocks(emu *e, uint64_t upto)     /* the oracle generator is sequential: extend on demand */
{
    if (upto <= e->tw_blocks) return;
    e->tw = (uint32_t *) realloc(e->tw, sizeof(uint32_t) * 2 * (size_t) e->n * (size_t) upto);
    orc_k_words((upto - e->tw_blocks) * 2 * (uint64_t) e->n, e->tw + e->tw_blocks * 2 * (uint64_t) e->n);
    e->tw_blocks = upto;
}

/* world_size-2 tests of the product's collective layer (comm.c) without a GPU: the initial rows are
 * produced in rank blocks and all-gathered over the communicator's host transport, the way
 * crs_engine.c's op_init_population does it on the device. */
static nlopt_amd_comm *emu_comm = NULL;
static int emu_forward = 1;                 /* device-resolved windows in the driver under test (orc_emu_set_forward) */
void orc_emu_set_forward(int on) { emu_forward = on; }
void orc_emu_set_comm(void *comm) { emu_comm = (nlopt_amd_comm *) comm; }

static int emu_init(void *ve, const double *x0, double *F)
{
    emu *e = (emu *) ve;
    size_t nw = 2 * (size_t) e->n * (size_t) (e->N - 1);
    uint32_t *w = (uint32_t *) malloc(sizeof(uint32_t) * (nw ? nw : 1));
    memcpy(e->X, x0, sizeof(double) * (size_t) e->n);
    orc_k_words(nw, w);                 /* the oracle generator is sequential: every rank draws the whole stream */
    if (nlopt_amd_comm_world(emu_comm) > 1) {
        int64_t per, first, mine;
        const size_t rowbytes = sizeof(double) * (size_t) e->ld;
        double *mineX, *allX, *mineF, *allF;
        nla_comm_partition(emu_comm, e->N - 1, &per, &first, &mine);
        mineX = (double *) calloc((size_t) per * e->ld, sizeof(double));
        allX = (double *) calloc((size_t) per * e->ld * (size_t) nlopt_amd_comm_world(emu_comm), sizeof(double));
        mineF = (double *) calloc((size_t) per, sizeof(double));
        allF = (double *) calloc((size_t) per * (size_t) nlopt_amd_comm_world(emu_comm), sizeof(double));
        orc_k_init_rows(e->n, e->ld, e->lb, e->ub, w + 2 * (size_t) e->n * (size_t) first, mine, mineX);
        if (e->obj >= 0) orc_k_eval(e->obj, e->n, e->ld, mineX, mine, mineF);
        if (nla_comm_allgather_host(emu_comm, mineX, allX, rowbytes * (size_t) per, NULL) ||
            nla_comm_allgather_host(emu_comm, mineF, allF, sizeof(double) * (size_t) per, NULL)) return -1;
        memcpy(e->X + e->ld, allX, rowbytes * (size_t) (e->N - 1));
        if (e->obj >= 0) { orc_k_eval(e->obj, e->n, e->ld, e->X, 1, F); memcpy(F + 1, allF, sizeof(double) * (size_t) (e->N - 1)); }
        free(mineX); free(allX); free(mineF); free(allF); free(w);
        return 0;
    }
    orc_k_init_rows(e->n, e->ld, e->lb, e->ub, w, e->N - 1, e->X + e->ld);
    free(w);
    if (e->obj >= 0) orc_k_eval(e->obj, e->n, e->ld, e->X, e->N, F);
    return 0;
}
static int emu_max_slots(void *ve, uint64_t first_block) { (void) first_block; return ((emu *) ve)->max_slots; }
static int emu_advance(void *ve, uint64_t first, int K, uint64_t fresh_from, int64_t i0, const int64_t *W, int nW,
                       nla_crs_slot_status *status)
{
    emu *e = (emu *) ve;
    const int n = e->n;
    if (K > e->cap) return -1;
    need_blocks(e, first + (uint64_t) K + 1);
    for (int a = 0; a < K; ++a) {
        const uint64_t b = first + (uint64_t) a;
        const int q = (int) (b % ECAP);
        double *acc = e->TX + (size_t) q * e->ld;
        int t0, t1;
        if (b >= fresh_from) {          /* a block seen for the first time: digest its words (crs.c:72,89-109) */
            orc_k_vitter(n, e->N, e->tw + b * 2 * (uint64_t) n, 1, e->jn + q, e->pos + (size_t) q * n, e->last + q);
            e->t[q] = 0;
        }
        t0 = e->t[q];
        t1 = orc_k_advance_slot(n, e->ld, e->X, i0, e->jn[q], e->pos + (size_t) q * n, e->last[q], W, a < nW ? a : nW, t0,
                                e->lb, e->ub, acc);
        e->t[q] = t1;
        if (t1 == n && t0 < n && e->obj >= 0) {
            double *m = e->TM + (size_t) q * e->ld;
            orc_k_eval(e->obj, n, e->ld, acc, 1, e->fT + q);
            orc_k_mutate(n, e->X + (size_t) i0 * e->ld, acc, e->tw + (b + 1) * 2 * (uint64_t) n, e->lb, e->ub, m);
            orc_k_eval(e->obj, n, e->ld, m, 1, e->fM + q);
        }
        status[a].t = t1; status[a].pad = 0;
        status[a].fT = t1 == n ? e->fT[q] : 0;
        status[a].fM = t1 == n ? e->fM[q] : 0;
    }
    return 0;
}
/* a whole window with the chain resolved by the "device" (port_kernels.c orc_k_crs_chain over this engine's rings, which are indexed
 * like the slot ring: block b at entry b % ECAP) */
static int emu_chain(void *ve, uint64_t first, int K, int64_t i0, double f_best, const int64_t *W, const double *Wf, int nW,
                     nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap)
{
    emu *e = (emu *) ve;
    const int n = e->n;
    uint32_t *wring;
    if (K > e->cap || e->obj < 0) return -1;
    need_blocks(e, first + (uint64_t) K + 1);
    wring = (uint32_t *) malloc(sizeof(uint32_t) * 2 * (size_t) n * ECAP);
    for (int a = 0; a <= K; ++a) {                 /* digest the window's blocks (+ the one behind it: mutation words) into the rings */
        const uint64_t b = first + (uint64_t) a;
        const int q = (int) (b % ECAP);
        memcpy(wring + (size_t) q * 2 * (size_t) n, e->tw + b * 2 * (uint64_t) n, sizeof(uint32_t) * 2 * (size_t) n);
        if (a < K) orc_k_vitter(n, e->N, e->tw + b * 2 * (uint64_t) n, 1, e->jn + q, e->pos + (size_t) q * n, e->last + q);
    }
    orc_k_crs_chain(e->obj, n, e->ld, e->X, i0, f_best, e->jn, e->pos, e->last, wring, ECAP, first, K, W, Wf, nW, ECAP - 1, e->lb, e->ub,
                    e->TX, e->TM, (orc_slot_status *) status, fwcnt, fwrec, fwcap, NULL, NULL);
    for (int a = 0; a < K; ++a) {
        const int q = (int) ((first + (uint64_t) a) % ECAP);
        e->t[q] = n; e->fT[q] = status[a].fT; e->fM[q] = status[a].fM;
    }
    free(wring);
    return 0;
}
static int emu_reset_slot(void *ve, uint64_t block) { ((emu *) ve)->t[block % ECAP] = 0; return 0; }

static int emu_commit(void *ve, int nc, const uint64_t *block, const int32_t *kind, const int64_t *row)
{
    emu *e = (emu *) ve;
    for (int c = 0; c < nc; ++c)
        memcpy(e->X + (size_t) row[c] * e->ld, (kind[c] == 1 ? e->TX : e->TM) + (size_t) (block[c] % ECAP) * e->ld,
               sizeof(double) * (size_t) e->n);
    return 0;
}
static int emu_read_slot(void *ve, uint64_t block, int kind, double *x)
{
    emu *e = (emu *) ve;
    memcpy(x, (kind == 1 ? e->TX : e->TM) + (size_t) (block % ECAP) * e->ld, sizeof(double) * (size_t) e->n);
    return 0;
}
static int emu_read_row(void *ve, int64_t row, double *x)
{
    emu *e = (emu *) ve;
    memcpy(x, e->X + (size_t) row * e->ld, sizeof(double) * (size_t) e->n);
    return 0;
}
static int emu_mutate_slot(void *ve, uint64_t block, int64_t i0)
{
    emu *e = (emu *) ve;
    double *p = e->TX + (size_t) (block % ECAP) * e->ld;
    need_blocks(e, block + 2);
    orc_k_mutate(e->n, e->X + (size_t) i0 * e->ld, p, e->tw + (block + 1) * 2 * (uint64_t) e->n, e->lb, e->ub, p);
    return 0;
}
static const char *emu_err(void *ve) { (void) ve; return "emu"; }

static const nla_crs_engine_ops emu_ops = { emu_init, emu_max_slots, emu_advance, emu_chain, emu_reset_slot, emu_commit, emu_read_slot,
                                            emu_read_row, emu_mutate_slot, emu_err, NULL, NULL };

/* Run the PRODUCT's CRS driver over the emulated engine.  RNG = the oracle generator (orc_srand
 * beforehand).  host_eval != 0 exercises the host-callback path with the zoo callback. */
int orc_emu_crs(int obj, int n, long N, const double *lb, const double *ub, double *x, double *minf,
                long maxeval, double stopval, double ftol_rel, double ftol_abs, double xtol_rel, const double *xtol_abs,
                int max_slots, int max_spec, int host_eval,
                nlopt_amd_trace_rec *trace, size_t trace_cap, size_t *trace_len, nlopt_amd_stats *stats,
                int *nevals_out, unsigned long long *words_out, double window_factor)
{
    emu e;
    nla_stopping stop;
    nla_crs_problem pb;
    int nevals = 0, force = 0, ret;
    uint64_t words = 0;
    char *msg = NULL;
    if (N == 0) N = 10 * ((long) n + 1);          /* crs.c:172-179, done by nla_crs_minimize in the product */
    if (N < n + 1) return -2;
    memset(&e, 0, sizeof e);
    e.n = n; e.ld = (n + 1) & ~1; e.N = N; e.obj = host_eval ? -1 : obj; e.lb = lb; e.ub = ub; e.cap = ECAP;
    e.max_slots = max_slots > 0 ? max_slots : 1024;
    e.X = (double *) calloc((size_t) e.ld * (size_t) N, sizeof(double));
    e.TX = (double *) calloc((size_t) e.ld * (size_t) e.cap, sizeof(double));
    e.TM = (double *) calloc((size_t) e.ld * (size_t) e.cap, sizeof(double));
    e.fT = (double *) calloc((size_t) e.cap, sizeof(double));
    e.fM = (double *) calloc((size_t) e.cap, sizeof(double));
    e.t = (int32_t *) calloc((size_t) e.cap, sizeof(int32_t));
    e.jn = (int32_t *) malloc(sizeof(int32_t) * (size_t) e.cap);
    e.last = (int32_t *) malloc(sizeof(int32_t) * (size_t) e.cap);
    e.pos = (int32_t *) malloc(sizeof(int32_t) * (size_t) e.cap * (size_t) n);
    memset(&stop, 0, sizeof stop);
    stop.n = (unsigned) n; stop.minf_max = stopval; stop.ftol_rel = ftol_rel; stop.ftol_abs = ftol_abs;
    stop.xtol_rel = xtol_rel; stop.xtol_abs = xtol_abs; stop.nevals_p = &nevals; stop.maxeval = (int) maxeval;
    stop.maxtime = 0; stop.start = 0; stop.force_stop = &force; stop.stop_msg = &msg;
    memset(&pb, 0, sizeof pb);
    pb.n = n; pb.N = N; pb.lb = lb; pb.ub = ub; pb.obj = e.obj;
    pb.f = (nlopt_func) orc_objective(obj); pb.f_data = NULL; pb.stop = &stop;
    pb.trace = trace; pb.trace_cap = trace_cap; pb.trace_len = trace_len; pb.stats = stats; pb.max_spec = max_spec; pb.window_factor = window_factor; pb.forward = emu_forward;
    if (trace_len) *trace_len = 0;
    ret = (int) nla_crs_run(&emu_ops, &e, &pb, x, minf, &words);
    *nevals_out = nevals; *words_out = words;
    free(e.X); free(e.TX); free(e.TM); free(e.fT); free(e.fM); free(e.t); free(e.jn); free(e.last); free(e.pos); free(e.tw); free(msg);
    return ret;
}
