/* crs_engine.c — the HIP engine behind the CRS2_LM driver: device memory, streams and kernel
 * sequencing.  Plain C over the C-ABI launchers of include/nlopt_amd.h (no HIP types here).
 *
 * Memory (all HBM, sized for one MI355X; N=1e5, n=4096 is 3.3 GB, N=1e6 33 GB of 288 GB):
 *   X          N x ld fp64 population              (the reference's ps matrix without the f column)
 *   block ring 2B stream blocks pre-digested ahead of use, block b at entry b % 2B:
 *              words (2n u32), jn/last (i32), pos (n i32); filled one half (= one batch of B
 *              blocks) at a time on the rng stream
 *   slot ring  KCAP slots, block b at slot b & (KCAP-1): TX (partial sum, then the finished trial
 *              point), TM (its mutation), fT/fM
 * Streams: `main` runs [upload + commit of the previous pass's accepts] -> advance -> finish ->
 * status D2H -> (host walk) each pass; `rng` runs the MT generator, the GF(2) jumps and the Vitter kernel for the *next* batch of
 * blocks concurrently (VALU-bound work hiding under the HBM-bound gather); an event per batch
 * orders main after rng.
 */
#include "nla_internal.h"
#include "nla_switches.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INIT_CHUNK_WORDS (1ULL << 28)      /* 1 GiB of stream words per init pass */
#define NLA_KC_MAX 16                 /* commits an advance launch can carry (hip/crs_kernels.hip) */
#define KCAP 1024                          /* slot ring (power of two) */
/* the objective id as the kernel launchers take it: a compiled-in objective under nlopt_set_max_objective delivers -f (e->sign) */
#define OBJK(e) (((e)->obj >= 0 && (e)->sign < 0) ? ((e)->obj | NLA_OBJ_NEGATE) : (e)->obj)
#define NLA_KARG_MAX 128                   /* list length that still travels as kernel arguments (hip/crs_kernels.hip NLA_KA_MAX) */
#define TIME_EVERY 8                       /* conservative passes below n = 2048: one in so many is timed */
#define TIME_EVERY_WINDOW 4                /* device-resolved windows below n = 2048 likewise */
#define ROWPAD 64                          /* spare rows behind X / F: the init all-gather wants equal blocks per rank (world <= 64) */

typedef struct {
    int64_t index;                 /* batch number (blocks [index*B, (index+1)*B)), -1 = empty */
    void *ev_ready;
    int waited;                    /* main stream already ordered after ev_ready */
} crs_batch;

/* one H2D per pass, packed: [W: nW i64][crow: nc i64][t_in: K i32][cslot: nc i32][ckind: nc i32][gen: K u32] */
#define UPLOAD_BYTES (KCAP * (8 + 8 + 4 + 4 + 4 + 4))
#define CHAIN_KMAX 256                     /* window slots / worst rows one device-resolved launch handles */
#define CHAIN_FWCAP 48                     /* records per slot (crs_driver.c FWCAP) */

struct nla_crs_hip_engine {
    int n, ld, obj;                /* obj: compiled-in objective id; -1: host callback; -2: user-supplied kernel (user, sign) */
    nla_userobj *user; double sign;
    int32_t *d_list, *h_list;      /* user kernel: the window's slot rows */
    double *h_fTM;                 /* user kernel: fT / fM rings read back (2 x KCAP) */
    int64_t N;
    int B;                         /* blocks per batch; the block ring holds 2B */
    int variant;                   /* advance-kernel tiling override (0 = automatic) */
    void *main, *rng;
    nla_mtstream *mts;
    double *d_lb, *d_ub, *d_X, *d_F;
    uint32_t *d_initwords; size_t initwords_cap;
    crs_batch bat[2];
    uint32_t *d_words;             /* 2B x 2n */
    int32_t *d_jn, *d_pos, *d_last;
    double *d_TX, *d_TM, *d_fT, *d_fM;
    char *d_up, *h_up;
    int32_t *d_tout;
    nla_crs_slot_status *d_status, *h_status;
    int32_t h_t[KCAP];             /* picks summed so far, per slot (host-authoritative) */
    int npending;                  /* commits staged on the host, not yet written to X */
    int fuse_commit;               /* 1: staged commits done inside the next pass's advance launch */
    int32_t pend_slot[KCAP], pend_kind[KCAP];
    int64_t pend_row[KCAP];
    void *ev0, *ev1;
    /* the gather kernel is timed with an event pair around it (the roofline figure of bench.py).  A pair costs two barrier packets
     * in front of and behind the kernel: nothing next to a 0.2-3 ms gather, a good part of a 10-20 us pass at small n — there only
     * every TIME_EVERY-th pass is timed; its bytes and its time go into the statistics together (gather_launches counts timed passes) */
    unsigned pass_no; int timed;
    int uncached;                  /* TX / TM / ctrl are uncached memory (the chain kernel may run) */
    int direct_status;             /* the finish kernel writes the status records into pinned host memory itself */
    int doorbell;                  /* ... and rings a word there when the last record is visible: the host spins on it instead of sleeping in a
                                    * stream synchronisation (conservative passes only) */
    uint32_t *h_bell, *d_bellcount, bell_seq;
    void (*idle_fn)(void *); void *idle_arg;   /* the driver's host work beside a pass (ops->set_idle) */
    /* device-resolved windows (hip/crs_chain.hip): control block, the walk's lists when they do not fit the kernel arguments,
     * what every slot took from where (pinned, written by the kernel) */
    void *d_ctrl;
    uint32_t ticket_base;
    uint32_t *h_fwcnt, *h_fwrec;
    double *d_Wf;
    int force_upload;              /* NLA_CRS_UPLOAD: the pass's lists through the H2D copy even when they fit the kernel arguments (A/B switch) */
    FILE *pass_log;                /* NLA_CRS_PASS_LOG=<file>: one line per pass (development aid, see tools/pass_log_summary.py) */
    nlopt_amd_stats *stats;
    nlopt_amd_comm *comm;          /* multi-GPU: NULL = single process */
    /* several GPUs, compiled-in objective: the population is sharded BY COORDINATE (hip/crs_shard.hip) — this rank holds columns
     * [c0, c0 + nc) of every row, c0 = rank * colper, colper = ceil(n / world); ld = the slice's row stride.  Every rank replays the
     * identical chain on identical f values: the candidates of a pass are all-gathered and evaluated by every rank */
    int sharded, world, rank, c0, nc, colper;
    int ncopy;                     /* coordinates a row copy moves: n, or nc */
    double *d_csend, *d_crecv;     /* a pass's candidates: 2 KCAP slices of colper doubles (+ 2 stop flags), world x that */
    int stop_in[2], stop_out[2];   /* the per-process stop conditions: this rank's view into a pass, all ranks' OR out of it */
    double *d_gsend, *d_grecv, *h_g;   /* a whole point from its slices (read_row / read_slot): colper, world x colper */
    const double *h_lb_full, *h_ub_full;   /* the caller's bounds (valid for the engine's life: the run's own arrays) */
    /* column-sharded population with the window resolved ON the device (hip/crs_chain.hip, SH instance; round 6): trial points are whole
     * (rows of ldf doubles) on every rank — a rank's workgroups store their chunk of a slot into every rank's TX through peer-mapped
     * memory — so every rank evaluates and resolves as a single device does and the 128-slot window survives the sharding */
    int shchain, ldf, ncolp;       /* ncolp: columns gathered = nc, even-padded from n = 128 on (the pad column of X is zero) */
    int sh_chunks, sh_chunk0, sh_chunks_total;
    uint32_t sh_seq;
    int sh_grid_cap;               /* ranks sharing one device ("amd_cu_share" = k): each rank's window kernel holds at most its k-th of the chip */
    void *d_shared;                /* ONE peer-mapped block of uncached memory: TX | flags | stop words */
    uint32_t *d_flags, *d_stopw;
    void *peer_block[8];           /* the other ranks' blocks as mapped here (nla_ipc_open) */
    void *d_table;                 /* what the kernel reads of all this (nla_crs_chain_sh_table) */
    double *d_xbest, *d_lbf, *d_ubf;   /* the whole best row, the whole bounds */
    int64_t xbest_row;             /* the row d_xbest is a copy of (-1: none yet) */
    char err[256];
};

#define NLA_CRS_FORWARD_MIN_N 1        /* device-resolved windows (hip/crs_chain.hip) from this dimension on — every dimension since round 5 (with the resolver
                                        * wavefront the windows beat the conservative passes at n = 64 / 128 / 256 too: profiles/r05_staged_ab.txt); the
                                        * conservative passes remain what host objectives, user kernels, column-sharded multi-rank jobs and "amd_forward" = 0 run on */

#define FAIL(e, ...) do { snprintf((e)->err, sizeof (e)->err, __VA_ARGS__); return -1; } while (0)
#define CK(e, call) do { int rc_ = (call); if (rc_) FAIL(e, "%.160s failed: %s", #call, nla_dev_error_string(rc_)); } while (0)

/* can n coordinates be dealt over `world` ranks in equal blocks of ceil(n / world) with nobody left empty? */
int nla_crs_can_shard(int n, int world)
{
    if (world < 2) return 0;
    return (int64_t) (world - 1) * ((n + world - 1) / world) < n;
}

/* ... and for the device-resolved windows: equal blocks of whole 128-byte lines (no line of a trial point is written by two ranks),
 * at most 8 ranks (the kernel's peer table) */
static int shchain_colper(int n, int world) { return (((n + world - 1) / world) + 15) & ~15; }
int nla_crs_can_shard_windows(int n, int world)
{
    if (world < 2 || world > 8) return 0;
    return (int64_t) (world - 1) * shchain_colper(n, world) < n;
}

static uint64_t trial_word0(const nla_crs_hip_engine *e) { return 2ULL * (uint64_t) e->n * (uint64_t) (e->N - 1); }

static int prepare_batch(nla_crs_hip_engine *e, crs_batch *b, int64_t index)
{
    const size_t half = (size_t) (index & 1) * (size_t) e->B;
    const uint64_t w0 = trial_word0(e) + 2ULL * (uint64_t) e->n * (uint64_t) index * (uint64_t) e->B;
    uint32_t *words = e->d_words + half * 2 * (size_t) e->n;
    b->index = index;
    b->waited = 0;
    if (nla_mtstream_fill(e->mts, w0, 2ULL * (uint64_t) e->n * (uint64_t) e->B, words))
        FAIL(e, "MT stream fill failed for batch %lld", (long long) index);
    CK(e, nla_k_crs_vitter(e->n, e->N, words, e->B, e->d_jn + half, e->d_pos + half * (size_t) e->n, e->d_last + half, e->rng));
    CK(e, nla_event_record(b->ev_ready, e->rng));
    return 0;
}

/* make the batches holding blocks [first, last] usable on the main stream (last < first + 2B - B
 * is guaranteed by op_max_slots: a window never reaches past the batch after first's) */
static int ensure_blocks(nla_crs_hip_engine *e, uint64_t first, uint64_t last)
{
    const int64_t i0 = (int64_t) (first / (uint64_t) e->B), i1 = (int64_t) (last / (uint64_t) e->B);
    crs_batch *cur = &e->bat[i0 & 1], *nxt = &e->bat[(i0 + 1) & 1];
    if (i1 > i0 + 1) FAIL(e, "window reaches past the prepared batches");
    /* everything that used the half being overwritten has been consumed: blocks < first are done */
    if (cur->index != i0 && prepare_batch(e, cur, i0)) return -1;
    if (nxt->index != i0 + 1 && prepare_batch(e, nxt, i0 + 1)) return -1;
    if (!cur->waited) { CK(e, nla_stream_wait_event(e->main, cur->ev_ready)); cur->waited = 1; }
    if (i1 > i0 && !nxt->waited) { CK(e, nla_stream_wait_event(e->main, nxt->ev_ready)); nxt->waited = 1; }
    return 0;
}

void nla_crs_hip_engine_destroy(nla_crs_hip_engine *e, uint64_t words_used)
{
    if (!e) return;
    if (e->main) nla_stream_sync(e->main);
    if (e->rng) nla_stream_sync(e->rng);
    if (e->mts) { nla_mtstream_finish(e->mts, words_used); nla_mtstream_destroy(e->mts); }
    if (e->pass_log) fclose(e->pass_log);
    for (int i = 0; i < 2; ++i) nla_event_destroy(e->bat[i].ev_ready);
    nla_dev_free(e->d_words); nla_dev_free(e->d_jn); nla_dev_free(e->d_pos); nla_dev_free(e->d_last);
    nla_dev_free(e->d_lb); nla_dev_free(e->d_ub); nla_dev_free(e->d_X); nla_dev_free(e->d_F);
    nla_dev_free(e->d_initwords);
    if (e->shchain) {
        for (int r = 0; r < 8; ++r) nla_ipc_close(e->peer_block[r]);
        nla_dev_free_uncached(e->d_shared); nla_dev_free_uncached(e->d_TM); nla_dev_free_uncached(e->d_ctrl);
        nla_dev_free(e->d_table); nla_dev_free(e->d_xbest); nla_dev_free(e->d_lbf); nla_dev_free(e->d_ubf);
    } else
    if (e->uncached) { nla_dev_free_uncached(e->d_TX); nla_dev_free_uncached(e->d_TM); nla_dev_free_uncached(e->d_ctrl); }
    else { nla_dev_free(e->d_TX); nla_dev_free(e->d_TM); nla_dev_free(e->d_ctrl); }
    nla_dev_free(e->d_fT);   /* d_fM aliases d_fT + KCAP */
    nla_dev_free(e->d_up); nla_dev_free(e->d_tout); nla_dev_free(e->d_status);
    nla_dev_free(e->d_list); nla_host_free(e->h_list); nla_host_free(e->h_fTM);
    nla_dev_free(e->d_Wf); nla_host_free(e->h_fwcnt); nla_host_free(e->h_fwrec);
    nla_host_free(e->h_up); nla_host_free(e->h_status); nla_host_free(e->h_bell); nla_dev_free(e->d_bellcount);
    nla_dev_free(e->d_csend); nla_dev_free(e->d_crecv); nla_dev_free(e->d_gsend); nla_dev_free(e->d_grecv); nla_host_free(e->h_g);
    nla_event_destroy(e->ev0); nla_event_destroy(e->ev1);
    if (e->rng != e->main) nla_stream_destroy(e->rng);
    nla_stream_destroy(e->main);
    free(e);
}

static int shchain_setup(nla_crs_hip_engine *e);

/* shard: 0 replicas / single process, 1 column-sharded with conservative passes, 2 column-sharded with device-resolved windows (cu_parts >= 2:
 * the ranks share one device — a test box — and each confines its window kernels to its share of the compute units) */
nla_crs_hip_engine *nla_crs_hip_engine_create(int n, int64_t N, const double *lb, const double *ub, int obj, int forward,
                                              nlopt_amd_comm *comm, int shard, int cu_parts, nlopt_amd_stats *stats, char **errmsg)
{
    nla_crs_hip_engine *e;
    size_t B;
    (void) errmsg;
    if (nla_dev_count() <= 0) return NULL;
    e = (nla_crs_hip_engine *) calloc(1, sizeof *e);
    if (!e) return NULL;
    e->n = n; e->N = N; e->obj = obj; e->stats = stats; e->comm = comm; e->h_lb_full = lb; e->h_ub_full = ub;
    e->world = nlopt_amd_comm_world(comm); e->rank = nlopt_amd_comm_rank(comm);
    e->sharded = shard && nla_crs_can_shard(n, e->world) && obj >= 0;
    e->shchain = e->sharded && shard == 2 && nla_crs_can_shard_windows(n, e->world);
    e->c0 = 0; e->nc = e->ncopy = n; e->colper = n;
    e->xbest_row = -1;
    /* (measured on this pool, tools/cu_mask_probe.hip: a CU-masked stream still runs on all 256 compute units — so what keeps the kernels of
     * ranks that share a device resident together is the cap on their workgroups: 224 of the chip's 256 one-workgroup-per-CU places dealt
     * over the ranks, the rest left to the generator's kernels) */
    e->sh_grid_cap = cu_parts >= 2 ? (224 / cu_parts >= 4 ? 224 / cu_parts : 4) : 0;
    if (e->sharded) {
        e->colper = e->shchain ? shchain_colper(n, e->world) : (n + e->world - 1) / e->world;
        e->c0 = e->rank * e->colper;
        e->nc = n - e->c0 < e->colper ? n - e->c0 : e->colper;
        e->ncopy = e->nc;
        forward = e->shchain;
    }
    e->ld = (e->nc + 15) & ~15;
    e->ldf = e->shchain ? ((n + 15) & ~15) : e->ld;
    e->ncolp = (n >= 128 && (e->nc & 1)) ? e->nc + 1 : e->nc;      /* rows start on a 128-byte line (the chain kernel's contract; coalesced row reads everywhere) */
    e->bat[0].index = e->bat[1].index = -1;
    /* blocks digested per batch (the block ring holds two batches).  The Vitter kernel walks all N rows per block, one lane per
     * block (a serial fp64 chain): a launch takes the same 40-70 ms (N = 1e5) whether it digests 4096 blocks or 32768 — it is
     * the number of lanes, not the time, that grows.  Measured with 4096-block batches at n = 4096 (profiles/r02_v1_kernel_stats.csv):
     * the kernel was running under the gather 40-60 % of the time, on a quarter of the CUs, and the gather's workgroups on those
     * CUs were the last to finish every pass.  With 2^28 words per batch (32768 blocks at n = 4096: 2.1 GB of words + 1 GB of
     * positions in the ring, of 288 GB) it runs a few per cent of the time. */
    {
        const char *lg = NLA_DBG_ENV("NLA_CRS_BATCH_LOG2");                /* development switch */
        const int lg2 = lg ? atoi(lg) : 28;
        B = (size_t) ((1ULL << (lg2 >= 20 && lg2 <= 31 ? lg2 : 28)) / (2ULL * (uint64_t) n));
    }
    if (B > 65536) B = 65536;
    if (B < 2 * KCAP) B = 2 * KCAP;
    e->B = (int) B;
    if (NLA_DBG_ENV("NLA_CRS_PASS_LOG")) e->pass_log = fopen(NLA_DBG_ENV("NLA_CRS_PASS_LOG"), "a");
    /* "amd_cu_share" = k >= 2: this process's window kernels on every k-th compute unit (ranks that share a device; a single process
     * asks for it to be measured on the same share, tools/shard_probe.py) */
    e->main = cu_parts >= 2 ? nla_stream_create_cu_share(e->rank % cu_parts, cu_parts) : nla_stream_create();
    if (!e->main && cu_parts >= 2) e->main = nla_stream_create();
    /* NLA_ONE_STREAM (A/B switch for debugging stream-ordering problems): the generator work on the main stream too */
    e->rng = NLA_DBG_ENV("NLA_ONE_STREAM") ? e->main : nla_stream_create();      /* (restricting this stream to a subset of the CUs — hipExtStreamCreateWithCUMask, every 4th / 16th CU — was measured:
                                        *  43.0 -> 43.1 k evals/s; what the digest kernels cost the gather is memory traffic, not CUs) */
    if (!e->main || !e->rng) goto fail;
    e->mts = nla_mtstream_create(e->rng);
    if (!e->mts) goto fail;
    e->d_lb = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld);
    e->d_ub = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld);
    e->d_X = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld * (size_t) (N + ROWPAD));
    e->d_F = (double *) nla_dev_malloc(sizeof(double) * (size_t) (N + ROWPAD));
    e->d_words = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * 2 * (size_t) n * 2 * B);
    e->d_jn = (int32_t *) nla_dev_malloc(sizeof(int32_t) * 2 * B);
    e->d_last = (int32_t *) nla_dev_malloc(sizeof(int32_t) * 2 * B);
    e->d_pos = (int32_t *) nla_dev_malloc(sizeof(int32_t) * 2 * B * (size_t) n);
    for (int i = 0; i < 2; ++i) {
        e->bat[i].ev_ready = nla_event_create();
        if (!e->bat[i].ev_ready) goto fail;
    }
    /* trial points and the chain kernel's control block: uncached memory where the workgroups of one launch read each other's
     * (device-resolved windows); ordinary memory for the conservative passes, whose kernels meet only at launch boundaries */
    e->uncached = (forward && obj >= 0) || NLA_DBG_ENV("NLA_UC_ALWAYS") != NULL;       /* (NLA_UC_ALWAYS: round 2's allocation pattern, A/B) */
    if (e->shchain) {
        /* whole trial points; TX, the chunk flags and the stop words in one block the other ranks map */
        const size_t txb = sizeof(double) * (size_t) e->ldf * KCAP;
        e->sh_chunks = nla_crs_chain_sh_chunks(n, e->ncolp);
        e->sh_chunks_total = 0;
        for (int r = 0; r < e->world; ++r) {
            const int c0r = r * e->colper, ncr = n - c0r < e->colper ? n - c0r : e->colper;
            if (r == e->rank) e->sh_chunk0 = e->sh_chunks_total;
            e->sh_chunks_total += nla_crs_chain_sh_chunks(n, (n >= 128 && (ncr & 1)) ? ncr + 1 : ncr);
        }
        {
            const size_t flb = (sizeof(uint32_t) * (size_t) CHAIN_KMAX * (size_t) e->sh_chunks_total + 127) & ~(size_t) 127;
            e->d_shared = nla_dev_malloc_uncached(txb + flb + nla_crs_chain_sh_stop_bytes());
            e->d_TX = (double *) e->d_shared;
            e->d_flags = e->d_shared ? (uint32_t *) ((char *) e->d_shared + txb) : NULL;
            e->d_stopw = e->d_shared ? (uint32_t *) ((char *) e->d_shared + txb + flb) : NULL;
            if (e->d_shared && nla_memset(e->d_flags, 0, flb + nla_crs_chain_sh_stop_bytes(), e->main)) goto fail;
        }
        e->d_TM = (double *) nla_dev_malloc_uncached(txb);
        e->d_table = nla_dev_malloc(nla_crs_chain_sh_table_bytes());
        e->d_xbest = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ldf);
        e->d_lbf = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ldf);
        e->d_ubf = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ldf);
    } else {
    e->d_TX = (double *) (e->uncached ? nla_dev_malloc_uncached : nla_dev_malloc)(sizeof(double) * (size_t) e->ld * KCAP);
    e->d_TM = (double *) (e->uncached ? nla_dev_malloc_uncached : nla_dev_malloc)(sizeof(double) * (size_t) e->ld * KCAP);
    }
    e->d_fT = (double *) nla_dev_malloc(sizeof(double) * 2 * KCAP);
    e->d_fM = e->d_fT ? e->d_fT + KCAP : NULL;
    e->d_up = (char *) nla_dev_malloc(UPLOAD_BYTES);
    e->d_tout = (int32_t *) nla_dev_malloc(sizeof(int32_t) * KCAP);
    e->d_status = (nla_crs_slot_status *) nla_dev_malloc(sizeof(nla_crs_slot_status) * (KCAP + 1));
    e->h_up = (char *) nla_host_malloc(UPLOAD_BYTES);
    e->h_status = (nla_crs_slot_status *) nla_host_malloc(sizeof(nla_crs_slot_status) * (KCAP + 1));
    e->ev0 = nla_event_create();
    e->ev1 = nla_event_create();
    if (obj == -2) {
        e->d_list = (int32_t *) nla_dev_malloc(sizeof(int32_t) * KCAP);
        e->h_list = (int32_t *) nla_host_malloc(sizeof(int32_t) * KCAP);
        e->h_fTM = (double *) nla_host_malloc(sizeof(double) * 2 * KCAP);
        if (!e->d_list || !e->h_list || !e->h_fTM) goto fail;
    }
    if (e->uncached && obj >= 0) {
        e->d_ctrl = nla_dev_malloc_uncached(nla_crs_chain_ctrl_bytes(CHAIN_KMAX, CHAIN_KMAX));
        e->d_Wf = (double *) nla_dev_malloc(sizeof(double) * CHAIN_KMAX);
        e->h_fwcnt = (uint32_t *) nla_host_malloc(sizeof(uint32_t) * CHAIN_KMAX);
        e->h_fwrec = (uint32_t *) nla_host_malloc(sizeof(uint32_t) * CHAIN_KMAX * CHAIN_FWCAP);
    }
    e->direct_status = !NLA_DBG_ENV("NLA_CRS_COPY_STATUS");
    e->h_bell = (uint32_t *) nla_host_malloc(64);       /* (explicitly coherent pinned memory measured equal within noise, profiles/r04_crs_doorbell_ab.txt: the
                                                          * finish kernel's system-scope fence is what makes the records visible before the bell) */
    e->d_bellcount = (uint32_t *) nla_dev_malloc(64);
    e->doorbell = 1;
    e->force_upload = NLA_DBG_ENV("NLA_CRS_UPLOAD") != NULL;
    if (!e->d_lb || !e->d_ub || !e->d_X || !e->d_F || !e->d_words || !e->d_jn || !e->d_last || !e->d_pos || !e->d_TX ||
        !e->d_TM || !e->d_fT || !e->d_up || !e->d_tout || !e->d_status || !e->h_up || !e->h_status || !e->h_bell || !e->d_bellcount || !e->ev0 || !e->ev1 ||
        (e->uncached && obj >= 0 && (!e->d_ctrl || !e->d_Wf || !e->h_fwcnt || !e->h_fwrec))) {
        /* (a window-mode rank still joins the handle exchange below, saying it failed: the others must not wait for it there) */
        if (e->shchain) shchain_setup(e);
        goto fail;
    }
    if (e->sharded) {
        /* (the candidates' all-gather buffers serve the conservative passes only: a window-mode run exchanges nothing by collective) */
        const size_t cslots = e->shchain ? 1 : 2 * KCAP;
        e->d_csend = (double *) nla_dev_malloc(sizeof(double) * (cslots * (size_t) e->colper + 2));
        e->d_crecv = (double *) nla_dev_malloc(sizeof(double) * (cslots * (size_t) e->colper + 2) * (size_t) e->world);
        e->d_gsend = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->colper);
        e->d_grecv = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->colper * (size_t) e->world);
        e->h_g = (double *) nla_host_malloc(sizeof(double) * (size_t) e->colper * (size_t) e->world);
        if (!e->d_csend || !e->d_crecv || !e->d_gsend || !e->d_grecv || !e->h_g) goto fail;
        {   /* the communicator's staging for the largest exchange of the run — a pass's candidates, or a rank's share of the initial
             * values — now, while a failure can still be agreed on (the set-up's "ready" exchange), not inside a collective */
            size_t big = sizeof(double) * (cslots * (size_t) e->colper + 2), ini = sizeof(double) * (size_t) ((e->N - 1 + e->world - 1) / e->world);
            if (nla_comm_reserve(comm, big > ini ? big : ini)) goto fail;
        }
        if (nla_memset(e->d_gsend, 0, sizeof(double) * (size_t) e->colper, e->main) ||
            nla_memset(e->d_csend, 0, sizeof(double) * (cslots * (size_t) e->colper + 2), e->main)) goto fail;
    }
    /* the slice's bounds (the whole vectors in a single-process run); pad entries are zero */
    *e->h_bell = 0;
    if (nla_memset(e->d_bellcount, 0, 64, e->main)) goto fail;
    if ((e->d_ctrl && nla_memset(e->d_ctrl, 0, nla_crs_chain_ctrl_bytes(CHAIN_KMAX, CHAIN_KMAX), e->main)) ||
        nla_memset(e->d_lb, 0, sizeof(double) * (size_t) e->ld, e->main) || nla_memset(e->d_ub, 0, sizeof(double) * (size_t) e->ld, e->main) ||
        nla_memcpy_h2d(e->d_lb, lb + e->c0, sizeof(double) * (size_t) e->nc, e->main) ||
        nla_memcpy_h2d(e->d_ub, ub + e->c0, sizeof(double) * (size_t) e->nc, e->main) || nla_stream_sync(e->main)) goto fail;
    if (e->shchain && shchain_setup(e)) goto fail;
    return e;
fail:
    nla_crs_hip_engine_destroy(e, 0);
    return NULL;
}

/* window-mode set-up: every rank exports its peer-mapped block, the handles are all-gathered (host data: 96 bytes per rank), every rank
 * maps the others' blocks and writes its table.  A rank whose allocations failed takes part with an invalid handle, so that nobody waits
 * for it; every rank then fails the set-up (the caller falls back to the conservative passes, on all ranks alike). */
static int shchain_setup(nla_crs_hip_engine *e)
{
    unsigned char mine[NLA_IPC_BYTES], all[8 * NLA_IPC_BYTES];
    void *ptx[8], *pfl[8], *pst[8];
    char *img;
    const size_t txb = sizeof(double) * (size_t) e->ldf * KCAP;
    const size_t flb = (sizeof(uint32_t) * (size_t) CHAIN_KMAX * (size_t) e->sh_chunks_total + 127) & ~(size_t) 127;
    int ok = e->d_shared && e->d_TM && e->d_table && e->d_xbest && e->d_lbf && e->d_ubf && e->d_ctrl && e->main, bad = 0;
    memset(mine, 0, sizeof mine);
    if (ok && (nla_stream_sync(e->main) || nla_ipc_export(e->d_shared, mine))) { ok = 0; memset(mine, 0, sizeof mine); }
    if (nla_comm_allgather_host(e->comm, mine, all, NLA_IPC_BYTES, e->main)) FAIL(e, "exchange of the window buffers' handles failed: %s", nlopt_amd_comm_error(e->comm));
    for (int r = 0; r < e->world; ++r) {
        int zero = 1;
        for (int i = 0; i < NLA_IPC_BYTES; ++i) if (all[r * NLA_IPC_BYTES + i]) { zero = 0; break; }
        if (zero) bad = 1;
    }
    if (!ok || bad) FAIL(e, "%s", ok ? "another rank could not set up its window buffers" : "window buffers: allocation or export failed");
    for (int r = 0; r < e->world; ++r) {
        char *base;
        if (r == e->rank) base = (char *) e->d_shared;
        else {
            e->peer_block[r] = nla_ipc_open(all + r * NLA_IPC_BYTES);
            if (!e->peer_block[r]) { bad = 1; base = (char *) e->d_shared; }      /* (agreed below) */
            else base = (char *) e->peer_block[r];
        }
        ptx[r] = base; pfl[r] = base + txb; pst[r] = base + txb + flb;
    }
    /* all ranks mapped all blocks, or none goes on */
    if (!nla_comm_agree_ready(e->comm, !bad) || bad) FAIL(e, "%s", bad ? "mapping a peer's window buffers failed (hipIpcOpenMemHandle)" : "another rank could not map the window buffers");
    img = (char *) malloc(nla_crs_chain_sh_table_bytes());
    if (!img) FAIL(e, "out of memory");
    if (nla_crs_chain_sh_table(img, e->world, e->rank, e->c0, e->ldf, e->sh_chunks_total, e->sh_chunk0, ptx, pfl, pst, e->d_xbest, e->d_lbf, e->d_ubf) ||
        nla_memcpy_h2d(e->d_table, img, nla_crs_chain_sh_table_bytes(), e->main) ||
        nla_memset(e->d_lbf, 0, sizeof(double) * (size_t) e->ldf, e->main) || nla_memset(e->d_ubf, 0, sizeof(double) * (size_t) e->ldf, e->main) ||
        nla_memset(e->d_xbest, 0, sizeof(double) * (size_t) e->ldf, e->main) ||
        nla_memcpy_h2d(e->d_lbf, e->h_lb_full, sizeof(double) * (size_t) e->n, e->main) ||
        nla_memcpy_h2d(e->d_ubf, e->h_ub_full, sizeof(double) * (size_t) e->n, e->main) || nla_stream_sync(e->main)) { free(img); FAIL(e, "window table upload failed"); }
    free(img);
    return 0;
}

/* development aid (NLA_CRS_DEBUG_DIR=<dir>): what the init kernels saw and made — the stream words, the rows and their f — of the
 * LAST run, as <dir>/init_last.bin: header {n, ld, N, nwords, origin} (5 x i64), words (u32), X (N x ld f64), F (N f64).  A harness
 * that finds a run diverging from the oracle keeps the file and can tell wrong words from wrong rows from wrong values. */
static void dump_init(nla_crs_hip_engine *e, size_t nwords)
{
    char path[512];
    FILE *fp;
    const size_t nx = (size_t) e->N * (size_t) e->ld;
    int64_t hdr[5] = { e->n, e->ld, e->N, (int64_t) nwords, (int64_t) nla_mtstream_origin(e->mts) };
    uint32_t *w = (uint32_t *) malloc(sizeof(uint32_t) * (nwords ? nwords : 1));
    double *x = (double *) malloc(sizeof(double) * (nx + (size_t) e->N));
    if (w && x && !nla_memcpy_d2h(w, e->d_initwords, sizeof(uint32_t) * nwords, e->main) &&
        !nla_memcpy_d2h(x, e->d_X, sizeof(double) * nx, e->main) && !nla_memcpy_d2h(x + nx, e->d_F, sizeof(double) * (size_t) e->N, e->main) &&
        !nla_stream_sync(e->main)) {
        snprintf(path, sizeof path, "%s/init_last.bin", NLA_DBG_ENV("NLA_CRS_DEBUG_DIR"));
        if ((fp = fopen(path, "wb"))) {
            fwrite(hdr, sizeof hdr, 1, fp); fwrite(w, sizeof(uint32_t), nwords, fp); fwrite(x, sizeof(double), nx + (size_t) e->N, fp);
            fclose(fp);
        }
    }
    free(w); free(x);
}

/* ---- ops ------------------------------------------------------------------------------------ */
/* crs_init's row loop (crs.c:211-226).  Row i >= 1 owns stream words [2n(i-1), 2n i), so any block of
 * rows can be produced independently: with a communicator, rank r generates and evaluates rows
 * [1 + r*per, 1 + (r+1)*per) and the rows and their f are ALL-GATHERED in place ("CRS
 * init"); every rank then holds the whole population and runs the identical trial chain. */
/* a whole point from its column slices: every rank contributes the nc coordinates it holds of the row at `src`, all-gathered */
static int gather_point(nla_crs_hip_engine *e, const double *src, double *x)
{
    /* (a rank that fails here cannot tell the others in band — they are waiting in this very exchange: nla_comm_abort) */
#define GP(call, what) do { int rc_ = (call); if (rc_) { snprintf(e->err, sizeof e->err, "%s failed: %s", what, nla_dev_error_string(rc_)); nla_comm_abort(e->comm); return -1; } } while (0)
    GP(nla_memcpy_d2d(e->d_gsend, src, sizeof(double) * (size_t) e->nc, e->main), "gather_point: D2D");
    if (nla_comm_allgather_dev(e->comm, e->d_gsend, e->d_grecv, sizeof(double) * (size_t) e->colper, e->main)) {
        nla_comm_abort(e->comm);
        FAIL(e, "all-gather of a point's slices failed: %s", nlopt_amd_comm_error(e->comm));
    }
    GP(nla_memcpy_d2h(e->h_g, e->d_grecv, sizeof(double) * (size_t) e->colper * (size_t) e->world, e->main), "gather_point: D2H");
    GP(nla_stream_sync(e->main), "gather_point: synchronisation");
#undef GP
    for (int r = 0; r < e->world; ++r) {
        const int c0 = r * e->colper, cnt = e->n - c0 < e->colper ? e->n - c0 : e->colper;
        memcpy(x + c0, e->h_g + (size_t) r * (size_t) e->colper, sizeof(double) * (size_t) cnt);
    }
    return 0;
}

/* crs_init's row loop on a column-sharded population.  Every rank draws ITS columns of every row from the same stream words (row
 * i >= 1 owns words [2n(i-1), 2n i), coordinate j the pair at 2j).  The objective needs whole rows: the ROWS are dealt over the
 * ranks in blocks for that — rank r evaluates rows [1 + r per, 1 + (r+1) per) straight from the words (crs_init_rows_kernel without
 * the store: the same reduction as a single-GPU run, bit for bit) — and the f values are all-gathered in place (8 bytes per row).
 * Nothing of the population itself travels. */
static int op_init_population_sharded(nla_crs_hip_engine *e, const double *x0, double *F)
{
    const int n = e->n;
    const uint64_t wpr = 2ULL * (uint64_t) n;
    const int64_t per = (e->N - 1 + e->world - 1) / e->world;          /* rows per rank (evaluation) */
    const int64_t first = 1 + per * e->rank, last = first + per < e->N ? first + per : e->N;
    int64_t rows_per_chunk = (int64_t) (INIT_CHUNK_WORDS / wpr), r0;
    void *ev = nla_event_create(), *ev_ag0 = nla_event_create(), *ev_ag1 = nla_event_create();
    double *h_row = (double *) calloc((size_t) (e->ld > n ? e->ld : n), sizeof(double));
    double *d_lbf = NULL, *d_ubf = NULL, *d_x0 = NULL;                 /* whole-row bounds and starting guess (evaluation only) */
    const double *lbh = NULL, *ubh = NULL;
    int rc = -1, stage_ok = 0;
    if (e->world > ROWPAD) { snprintf(e->err, sizeof e->err, "more than %d ranks are not supported", ROWPAD); goto out; }   /* (the same on every rank) */
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    if (rows_per_chunk > e->N - 1) rows_per_chunk = e->N - 1 > 0 ? e->N - 1 : 1;
    e->d_initwords = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * (size_t) (rows_per_chunk * (int64_t) wpr));
    d_lbf = (double *) nla_dev_malloc(sizeof(double) * (size_t) n);
    d_ubf = (double *) nla_dev_malloc(sizeof(double) * (size_t) n);
    d_x0 = (double *) nla_dev_malloc(sizeof(double) * (size_t) n);
    {   /* all ranks go into the all-gather below, or none (comm.c, nla_comm_agree_ready) */
        const int mine = ev && ev_ag0 && ev_ag1 && h_row && e->d_initwords && d_lbf && d_ubf && d_x0;
        if (!nla_comm_agree_ready(e->comm, mine) || !mine) {
            snprintf(e->err, sizeof e->err, "%s", mine ? "another rank ran out of memory (init)" : "out of memory (init)");
            goto out;
        }
    }
    lbh = e->h_lb_full; ubh = e->h_ub_full;
    if (nla_memcpy_h2d(d_lbf, lbh, sizeof(double) * (size_t) n, e->main) || nla_memcpy_h2d(d_ubf, ubh, sizeof(double) * (size_t) n, e->main) ||
        nla_memcpy_h2d(d_x0, x0, sizeof(double) * (size_t) n, e->main)) { snprintf(e->err, sizeof e->err, "H2D (init) failed"); goto agree2; }
    /* row 0 = the caller's starting guess (crs.c:204): its slice (pad zero), and its value from the whole point */
    memcpy(h_row, x0 + e->c0, sizeof(double) * (size_t) e->nc);
    if (nla_memcpy_h2d(e->d_X, h_row, sizeof(double) * (size_t) e->ld, e->main) ||
        nla_k_eval(OBJK(e), n, n, d_x0, 1, e->d_F, e->main) || nla_stream_sync(e->main)) { snprintf(e->err, sizeof e->err, "row 0 failed"); goto agree2; }
    for (r0 = 1; r0 < e->N; r0 += rows_per_chunk) {
        const int64_t nr = e->N - r0 < rows_per_chunk ? e->N - r0 : rows_per_chunk;
        const int64_t lo = first > r0 ? first : r0, hi = last < r0 + nr ? last : r0 + nr;       /* this rank's rows inside the chunk */
        if (r0 > 1 && (nla_event_record(ev, e->main) || nla_stream_wait_event(e->rng, ev))) { snprintf(e->err, sizeof e->err, "event failed"); goto agree2; }
        if (nla_mtstream_fill(e->mts, wpr * (uint64_t) (r0 - 1), wpr * (uint64_t) nr, e->d_initwords)) { snprintf(e->err, sizeof e->err, "MT stream fill failed (init)"); goto agree2; }
        if (nla_event_record(ev, e->rng) || nla_stream_wait_event(e->main, ev)) { snprintf(e->err, sizeof e->err, "event failed"); goto agree2; }
        if (nla_k_crs_sh_init_rows(n, e->c0, e->nc, e->ld, e->d_lb, e->d_ub, e->d_initwords, r0, nr, e->d_X, e->main) ||
            (hi > lo && nla_k_crs_init_rows(OBJK(e), n, n, d_lbf, d_ubf, e->d_initwords + (size_t) (lo - r0) * (size_t) wpr, lo, hi - lo, NULL, e->d_F, e->main))) {
            snprintf(e->err, sizeof e->err, "init kernel launch failed"); goto agree2;
        }
    }
    stage_ok = 1;
agree2:
    /* every rank arrives here, whatever happened to it since the first agreement (H2D copies, fills, launches): all of them go
     * into the all-gather of the values, or none — a rank that failed alone would leave the others waiting in it */
    if (!nla_comm_agree_ready(e->comm, stage_ok) || !stage_ok) {
        if (stage_ok) snprintf(e->err, sizeof e->err, "another rank failed while it initialised its slice of the population");
        goto out;
    }
    nla_event_record(ev_ag0, e->main);
    if (nla_comm_allgather_dev(e->comm, e->d_F + first, e->d_F + 1, sizeof(double) * (size_t) per, e->main)) {
        snprintf(e->err, sizeof e->err, "all-gather of the initial values failed: %s", nlopt_amd_comm_error(e->comm)); nla_comm_abort(e->comm); goto out;
    }
    nla_event_record(ev_ag1, e->main);
    {   /* ... and all of them leave the initialisation together */
        const int mine = !(nla_memcpy_d2h(F, e->d_F, sizeof(double) * (size_t) e->N, e->main) || nla_stream_sync(e->main));
        if (!nla_comm_agree_ready(e->comm, mine) || !mine) {
            snprintf(e->err, sizeof e->err, "%s", mine ? "another rank could not read back the initial values" : "D2H F failed");
            goto out;
        }
    }
    if (e->stats) {
        e->stats->t_allgather_ms += (double) nla_event_elapsed_ms(ev_ag0, ev_ag1);
        e->stats->allgather_bytes += (uint64_t) e->world * (uint64_t) per * sizeof(double);
    }
    rc = 0;
out:
    nla_event_destroy(ev); nla_event_destroy(ev_ag0); nla_event_destroy(ev_ag1);
    free(h_row);
    nla_dev_free(d_lbf); nla_dev_free(d_ubf); nla_dev_free(d_x0);
    nla_dev_free(e->d_initwords); e->d_initwords = NULL; e->initwords_cap = 0;
    if (rc) return -1;
    return ensure_blocks(e, 0, 0);
}

static int op_init_population(void *ve, const double *x0, double *F)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const int n = e->n;
    const int world = nlopt_amd_comm_world(e->comm), rank = nlopt_amd_comm_rank(e->comm);
    const uint64_t wpr = 2ULL * (uint64_t) n;            /* words per row */
    const int64_t per = (e->N - 1 + world - 1) / world;  /* rows per rank */
    const int64_t first = 1 + per * rank;
    const int64_t last = first + per < e->N ? first + per : e->N;     /* this rank's rows: [first, last) */
    int64_t rows_per_chunk = (int64_t) (INIT_CHUNK_WORDS / wpr), r0;
    void *ev, *ev_ag0 = NULL, *ev_ag1 = NULL;
    if (e->sharded) return op_init_population_sharded(e, x0, F);
    if (world > ROWPAD) FAIL(e, "more than %d ranks are not supported", ROWPAD);          /* (the same on every rank) */
    ev = nla_event_create();
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    if (rows_per_chunk > per) rows_per_chunk = per;
    if (last > first) {
        size_t need = (size_t) (rows_per_chunk * (int64_t) wpr);
        if (need > e->initwords_cap) {
            nla_dev_free(e->d_initwords);
            e->d_initwords = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * need);
            e->initwords_cap = e->d_initwords ? need : 0;
        }
    }
    {   /* several ranks: all of them go into the all-gathers below, or none (comm.c, nla_comm_agree_ready) */
        const int mine = ev && (last <= first || e->d_initwords);
        if (!nla_comm_agree_ready(e->comm, mine) || !mine) {
            nla_event_destroy(ev);
            FAIL(e, "%s", mine ? "another rank ran out of memory (init)" : (ev ? "out of device memory (init words)" : "event create failed"));
        }
    }
    /* row 0 = the caller's starting guess (crs.c:204) */
    if (nla_memcpy_h2d(e->d_X, x0, sizeof(double) * (size_t) n, e->main)) { nla_event_destroy(ev); FAIL(e, "H2D x0 failed"); }
    if (e->obj >= 0 && nla_k_eval(OBJK(e), n, e->ld, e->d_X, 1, e->d_F, e->main)) { nla_event_destroy(ev); FAIL(e, "eval launch failed"); }
    if (e->obj == -2 && nla_userobj_eval_rows(e->user, n, e->ld, 1, e->d_X, e->d_F, NULL, e->sign, e->main)) { nla_event_destroy(ev); FAIL(e, "user objective launch failed"); }
    for (r0 = first; r0 < last; r0 += rows_per_chunk) {
        int64_t nr = last - r0 < rows_per_chunk ? last - r0 : rows_per_chunk;
        /* the words buffer is reused: the generator must not overwrite it before the previous
         * chunk's init kernel has consumed it */
        if (r0 > first) {
            if (nla_event_record(ev, e->main) || nla_stream_wait_event(e->rng, ev)) { nla_event_destroy(ev); FAIL(e, "event failed"); }
        }
        if (nla_mtstream_fill(e->mts, wpr * (uint64_t) (r0 - 1), wpr * (uint64_t) nr, e->d_initwords)) {
            nla_event_destroy(ev); FAIL(e, "MT stream fill failed (init)");
        }
        if (nla_event_record(ev, e->rng) || nla_stream_wait_event(e->main, ev)) { nla_event_destroy(ev); FAIL(e, "event failed"); }
        if (nla_k_crs_init_rows(e->obj >= 0 ? OBJK(e) : -1, n, e->ld, e->d_lb, e->d_ub, e->d_initwords, r0, nr, e->d_X, e->d_F, e->main) ||
            (e->obj == -2 && nla_userobj_eval_rows(e->user, n, e->ld, nr, e->d_X + (size_t) r0 * (size_t) e->ld, e->d_F + r0, NULL, e->sign, e->main))) {
            nla_event_destroy(ev); FAIL(e, "init kernel launch failed");
        }
    }
    if (world > 1) {
        ev_ag0 = nla_event_create(); ev_ag1 = nla_event_create();
        if (ev_ag0) nla_event_record(ev_ag0, e->main);
        if (nla_comm_allgather_dev(e->comm, e->d_X + (size_t) first * (size_t) e->ld, e->d_X + (size_t) e->ld,
                                   sizeof(double) * (size_t) per * (size_t) e->ld, e->main) ||
            (e->obj != -1 && nla_comm_allgather_dev(e->comm, e->d_F + first, e->d_F + 1, sizeof(double) * (size_t) per, e->main))) {
            nla_event_destroy(ev); nla_event_destroy(ev_ag0); nla_event_destroy(ev_ag1);
            FAIL(e, "all-gather of the initial population failed: %s", nlopt_amd_comm_error(e->comm));
        }
        if (ev_ag1) nla_event_record(ev_ag1, e->main);
    }
    if (e->obj != -1) {
        if (nla_memcpy_d2h(F, e->d_F, sizeof(double) * (size_t) e->N, e->main)) { nla_event_destroy(ev); FAIL(e, "D2H F failed"); }
    }
    {
        int rc = nla_stream_sync(e->main);
        nla_event_destroy(ev);
        if (!rc && ev_ag0 && ev_ag1 && e->stats) {
            e->stats->t_allgather_ms += (double) nla_event_elapsed_ms(ev_ag0, ev_ag1);
            e->stats->allgather_bytes += (uint64_t) world * (uint64_t) per * (sizeof(double) * (uint64_t) e->ld + (e->obj != -1 ? sizeof(double) : 0));
        }
        nla_event_destroy(ev_ag0); nla_event_destroy(ev_ag1);
        if (rc) FAIL(e, "init sync failed: %s", nla_dev_error_string(rc));
    }
    if (NLA_DBG_ENV("NLA_CRS_DEBUG_DIR") && world == 1 && rows_per_chunk >= per && !e->sharded) dump_init(e, (size_t) (wpr * (uint64_t) (e->N - 1)));
    nla_dev_free(e->d_initwords); e->d_initwords = NULL; e->initwords_cap = 0;
    /* start digesting the first batch of trial blocks while the host builds its ordered set */
    if (ensure_blocks(e, 0, 0)) return -1;
    return 0;
}

static int op_max_slots(void *ve, uint64_t first_block)
{
    /* the window and the block after it (mutation words) must lie in first_block's batch or the next */
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const uint64_t B = (uint64_t) e->B;
    uint64_t rem = (first_block / B + 2) * B - 1 - first_block;
    return (int) (rem < KCAP ? rem : KCAP);
}

/* upload W / t_in of the coming pass (K may be 0: commits only) together with the staged commits
 * in ONE copy, and write the commits to X before anything else runs on the main stream */
static int upload_and_commit(nla_crs_hip_engine *e, const int64_t *W, int nW, const int32_t *t_in, int K,
                             const int64_t **d_W, const int32_t **d_tin, const uint32_t *gen, const uint32_t **d_gen)
{
    const int nc = e->npending;
    size_t oW = 0, oR = oW + 8 * (size_t) nW, oT = oR + 8 * (size_t) nc, oS = oT + 4 * (size_t) K, oK = oS + 4 * (size_t) nc;
    const size_t oG = oK + 4 * (size_t) nc;
    const size_t total = oG + (gen ? 4 * (size_t) K : 0);
    if (total == 0) return 0;
    if (nW) memcpy(e->h_up + oW, W, 8 * (size_t) nW);
    if (nc) {
        memcpy(e->h_up + oR, e->pend_row, 8 * (size_t) nc);
        memcpy(e->h_up + oS, e->pend_slot, 4 * (size_t) nc);
        memcpy(e->h_up + oK, e->pend_kind, 4 * (size_t) nc);
    }
    if (K) memcpy(e->h_up + oT, t_in, 4 * (size_t) K);
    if (gen && K) memcpy(e->h_up + oG, gen, 4 * (size_t) K);
    CK(e, nla_memcpy_h2d(e->d_up, e->h_up, total, e->main));
    if (nc) {
        CK(e, nla_k_crs_commit(e->ncopy, e->ld, e->d_X, e->d_TX, e->d_TM, nc, (const int32_t *) (e->d_up + oS),
                               (const int32_t *) (e->d_up + oK), (const int64_t *) (e->d_up + oR), e->main));
        e->npending = 0;
    }
    if (d_W) *d_W = (const int64_t *) (e->d_up + oW);
    if (d_tin) *d_tin = (const int32_t *) (e->d_up + oT);
    if (d_gen) *d_gen = (const uint32_t *) (e->d_up + oG);
    return 0;
}

/* the pinned upload buffer is rewritten by the next pass: callers that do not synchronise
 * themselves must do so before the host touches it again (every pass ends with a sync) */
static int commit_sh(nla_crs_hip_engine *e, void *zero, size_t zero_bytes, int best_slot, int best_kind);
static int flush_commits(nla_crs_hip_engine *e)
{
    if (!e->npending) return 0;
    if (e->shchain) {
        if (commit_sh(e, NULL, 0, -1, 0)) return -1;
        CK(e, nla_stream_sync(e->main));
        return 0;
    }
    if (upload_and_commit(e, NULL, 0, NULL, 0, NULL, NULL, NULL, NULL)) return -1;
    CK(e, nla_stream_sync(e->main));
    return 0;
}

/* spin until the finish kernel's doorbell shows `seq` (1: it did not within 20 ms — the caller synchronises the stream instead) */
static int bell_wait(nla_crs_hip_engine *e, uint32_t seq)
{
    unsigned spins = 0;
    double t0 = 0;
    while (__atomic_load_n(e->h_bell, __ATOMIC_ACQUIRE) != seq) {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((++spins & 4095u) == 0) {
            const double t = nla_seconds();
            if (t0 == 0) t0 = t;
            else if (t - t0 > 0.02) return 1;
        }
    }
    return 0;
}

static int op_advance(void *ve, uint64_t first_block, int K, uint64_t fresh_from, int64_t i0, const int64_t *W, int nW,
                      nla_crs_slot_status *status)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const int n = e->n;
    const uint32_t ring = 2u * (uint32_t) e->B;
    int32_t t_in[KCAP];
    const int64_t *d_W = NULL;
    const int32_t *d_tin = NULL;
    if (K < 1 || K > KCAP || nW > KCAP || nW < 0) FAIL(e, "bad window K=%d nW=%d", K, nW);
    if (e->shchain) FAIL(e, "this engine was set up for device-resolved windows on a column-sharded population: no conservative passes");
    if (K > op_max_slots(ve, first_block)) FAIL(e, "window reaches past the prepared batches");
    if (ensure_blocks(e, first_block, first_block + (uint64_t) K)) { if (!e->err[0]) snprintf(e->err, sizeof e->err, "batch preparation failed"); return -1; }
    e->timed = (e->n >= 2048 || e->pass_log || (e->pass_no++ % TIME_EVERY) == 0) && e->stats;
#define EVREC(ev) do { if (e->timed) CK(e, nla_event_record((ev), e->main)); } while (0)
    for (int a = 0; a < K; ++a) {
        const uint64_t b = first_block + (uint64_t) a;
        t_in[a] = b >= fresh_from ? 0 : e->h_t[b & (KCAP - 1)];
    }
    if (e->sharded) {
        /* one rank of a column-sharded run: the same pass on the slice; the slices of the candidates that completed (trial point and
         * mutation) are all-gathered and every rank evaluates the assembled points — identical status records on every rank */
        const int ncol = (e->ld % 2 == 0 && e->nc % 2 != 0) ? e->nc + 1 : e->nc;      /* (an even count lets the gather take coordinate pairs; the pad column is zero) */
        /* A failure on this rank alone must not leave the others waiting in the all-gather: it is REMEMBERED, the rank still packs
         * (whatever its slots hold) and joins the exchange with the value 2 in its first flag word, and every rank — this one
         * included — sees it in record K and leaves the pass with the same error.  Only a rank that cannot even pack and send gives
         * up out of band (nla_comm_abort: the shm transport's barriers fail; the other transports have no such channel). */
        int lerr = 0;
#define SOFT(call) do { if (!lerr) { int rc_ = (call); if (rc_) { lerr = 1; snprintf(e->err, sizeof e->err, "%.160s failed: %s", #call, nla_dev_error_string(rc_)); } } } while (0)
#define HARD(call) do { int rc_ = (call); if (rc_) { if (!lerr) snprintf(e->err, sizeof e->err, "%.160s failed: %s", #call, nla_dev_error_string(rc_)); nla_comm_abort(e->comm); return -1; } } while (0)
        if (upload_and_commit(e, W, nW, t_in, K, &d_W, &d_tin, NULL, NULL)) { lerr = 1; d_tin = e->d_tout; }      /* (message set; any device array of K ints serves the pack) */
        if (nla_dbg_int("NLA_CRS_FAIL_RANK", -1) == e->rank && (int64_t) e->pass_no == (int64_t) nla_dbg_int("NLA_CRS_FAIL_PASS", -1)) {
            lerr = 1; snprintf(e->err, sizeof e->err, "injected failure (NLA_CRS_FAIL_RANK / NLA_CRS_FAIL_PASS)");
        }
        EVREC(e->ev0);
        SOFT(nla_k_crs_advance_cols(n, ncol, e->ld, e->d_X, i0, e->d_jn, e->d_pos, e->d_last, ring, first_block, K, d_W, nW,
                                    d_tin, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_TX, e->variant, e->main));
        EVREC(e->ev1);
        HARD(nla_k_crs_sh_mutate_pack(n, e->c0, e->nc, e->ld, e->colper, e->d_X, i0, e->d_TX, e->d_TM, e->d_words, ring, first_block, K,
                                      d_tin, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_csend, lerr ? 2 : e->stop_in[0], e->stop_in[1], e->main));
        if (nla_comm_allgather_dev(e->comm, e->d_csend, e->d_crecv, sizeof(double) * (2 * (size_t) K * (size_t) e->colper + 2), e->main)) {
            nla_comm_abort(e->comm);
            if (!lerr) snprintf(e->err, sizeof e->err, "all-gather of the candidates failed: %s", nlopt_amd_comm_error(e->comm));
            return -1;
        }
        HARD(nla_k_crs_sh_eval(OBJK(e), n, e->colper, first_block, K, d_tin, e->d_tout, KCAP - 1, e->d_crecv, e->world, e->d_fT, e->d_fM, e->d_status, e->main));
        if (e->stats) e->stats->allgather_bytes += (uint64_t) e->world * (2 * (uint64_t) K * (uint64_t) e->colper + 2) * sizeof(double);
        /* the K status records and, behind them, the ranks' agreed stop flags (and whether any rank failed) */
        HARD(nla_memcpy_d2h(e->h_status, e->d_status, sizeof(nla_crs_slot_status) * ((size_t) K + 1), e->main));
        if (e->idle_fn) e->idle_fn(e->idle_arg);
        HARD(nla_stream_sync(e->main));
#undef SOFT
#undef HARD
        if (e->h_status[K].t != 0) {
            if (!lerr) snprintf(e->err, sizeof e->err, "another rank failed in the middle of a sharded pass");
            return -1;
        }
        e->stop_out[0] = e->h_status[K].fT != 0.; e->stop_out[1] = e->h_status[K].fM != 0.;
        goto have_status;
    }
    if (K <= NLA_KARG_MAX && nW <= NLA_KARG_MAX && e->npending <= NLA_KARG_MAX && !e->force_upload && e->obj != -2) {
        /* small lists (the usual case): W, the resume points and the staged commits travel as kernel arguments — no copy in
         * front of the pass */
        /* the staged commits ride in the advance launch (hip/crs_kernels.hip, crs_fwd: copied by extra workgroups, reads of those rows
         * forwarded to the slots they come from) unless one of their source slots is a slot of this pass's own window — the ring
         * wrapped onto it — or there are too many: then the commit launch of its own in front, as before */
        int fuse = e->fuse_commit && e->npending > 0 && e->npending <= NLA_KC_MAX;
        for (int c = 0; fuse && c < e->npending; ++c)
            for (int a = 0; a < K; ++a)
                if ((int32_t) ((first_block + (uint64_t) a) & (KCAP - 1)) == e->pend_slot[c]) { fuse = 0; break; }
        if (e->npending && !fuse) {
            CK(e, nla_k_crs_commit_args(n, e->ld, e->d_X, e->d_TX, e->d_TM, e->npending, e->pend_slot, e->pend_kind, e->pend_row, e->main));
            e->npending = 0;
        }
        EVREC(e->ev0);
        if (fuse) {
            CK(e, nla_k_crs_advance_commit_args(n, e->ld, e->d_X, i0, e->d_jn, e->d_pos, e->d_last, ring, first_block, K, W, nW,
                                                t_in, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_TX, e->d_TM, e->npending, e->pend_slot,
                                                e->pend_kind, e->pend_row, e->variant, e->main));
            e->npending = 0;
        } else
        CK(e, nla_k_crs_advance_args(n, e->ld, e->d_X, i0, e->d_jn, e->d_pos, e->d_last, ring, first_block, K, W, nW,
                                     t_in, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_TX, e->variant, e->main));
        EVREC(e->ev1);
        if (e->direct_status && e->doorbell) {
            /* the finish kernel writes the K records straight into the pinned host buffer and rings: the pass is over for the host
             * when the bell shows this pass's number — no copy-back, no stream synchronisation (its wake-up is a third of a 30 us pass
             * at n = 512); a bell that does not come within 20 ms falls back to the synchronisation, which reports what happened */
            const uint32_t seq = ++e->bell_seq ? e->bell_seq : ++e->bell_seq;
            CK(e, nla_k_crs_finish_args_bell(OBJK(e), n, e->ld, e->d_X, i0, e->d_TX, e->d_TM, e->d_words, ring, first_block, K,
                                             t_in, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_fT, e->d_fM, e->h_status, e->d_bellcount,
                                             e->h_bell, seq, e->main));
            if (e->idle_fn) e->idle_fn(e->idle_arg);
            if (bell_wait(e, seq)) CK(e, nla_stream_sync(e->main));
            goto have_status;
        }
        CK(e, nla_k_crs_finish_args(OBJK(e), n, e->ld, e->d_X, i0, e->d_TX, e->d_TM, e->d_words, ring, first_block, K,
                                    t_in, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_fT, e->d_fM,
                                    e->direct_status ? e->h_status : e->d_status, e->main));
        if (e->direct_status) {
            /* the finish kernel wrote the K records straight into the pinned host buffer (visible at kernel completion):
             * no copy-back operation behind it either */
            if (e->idle_fn) e->idle_fn(e->idle_arg);
            CK(e, nla_stream_sync(e->main));
            goto have_status;
        }
        goto launched;
    }
    if (upload_and_commit(e, W, nW, t_in, K, &d_W, &d_tin, NULL, NULL)) return -1;
    EVREC(e->ev0);
    CK(e, nla_k_crs_advance(n, e->ld, e->d_X, i0, e->d_jn, e->d_pos, e->d_last, ring, first_block, K, d_W, nW,
                            d_tin, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_TX, e->variant, e->main));
    EVREC(e->ev1);
    CK(e, nla_k_crs_finish(OBJK(e), n, e->ld, e->d_X, i0, e->d_TX, e->d_TM, e->d_words, ring, first_block, K,
                           d_tin, e->d_tout, KCAP - 1, e->d_lb, e->d_ub, e->d_fT, e->d_fM, e->d_status, e->main));
launched:
    if (e->obj == -2) {
        /* the objective is the user's kernel: f of every window slot's trial point and mutation in two launches over the slot
         * rows (rows of unfinished slots hold partial sums: their values are never looked at), then the rings come back */
        for (int a = 0; a < K; ++a) e->h_list[a] = -(int32_t) (((first_block + (uint64_t) a) & (KCAP - 1)) + 1);
        CK(e, nla_memcpy_h2d(e->d_list, e->h_list, sizeof(int32_t) * (size_t) K, e->main));
        CK(e, nla_userobj_evalgrad_list(e->user, n, e->ld, K, e->d_list, e->d_TX, e->d_fT, NULL, e->sign, e->main));
        CK(e, nla_userobj_evalgrad_list(e->user, n, e->ld, K, e->d_list, e->d_TM, e->d_fM, NULL, e->sign, e->main));
        CK(e, nla_memcpy_d2h(e->h_fTM, e->d_fT, sizeof(double) * 2 * KCAP, e->main));
    }
    CK(e, nla_memcpy_d2h(e->h_status, e->d_status, sizeof(nla_crs_slot_status) * (size_t) K, e->main));
    if (e->idle_fn) e->idle_fn(e->idle_arg);
    CK(e, nla_stream_sync(e->main));
    if (e->obj == -2)
        for (int a = 0; a < K; ++a) {
            const size_t q = (size_t) ((first_block + (uint64_t) a) & (KCAP - 1));
            e->h_status[a].fT = e->h_fTM[q]; e->h_status[a].fM = e->h_fTM[KCAP + q];
        }
have_status:
    memcpy(status, e->h_status, sizeof(nla_crs_slot_status) * (size_t) K);

    for (int a = 0; a < K; ++a) e->h_t[(first_block + (uint64_t) a) & (KCAP - 1)] = status[a].t;
    if (e->timed) {
        float ms = nla_event_elapsed_ms(e->ev0, e->ev1);
        if (ms >= 0) e->stats->t_gather_ms += ms;
        e->stats->gather_launches += 1;
    }
#undef EVREC
    if (e->pass_log) {             /* K, nW, slots already complete / fresh / stopped short, rows summed, kernel ms */
        long rows = 0;
        int done_in = 0, fresh = 0, stopped = 0;
        for (int a = 0; a < K; ++a) {
            rows += status[a].t - t_in[a];
            done_in += t_in[a] == n;
            fresh += t_in[a] == 0;
            stopped += status[a].t < n;
        }
        fprintf(e->pass_log, "%d,%d,%d,%d,%d,%d,%ld,%.4f\n", n, K, nW, done_in, fresh, stopped, rows, (double) nla_event_elapsed_ms(e->ev0, e->ev1));
    }
    return 0;
}

/* staged; written to X by the next pass's single upload (or by a flush when something reads X) */
static int op_commit(void *ve, int ncommit, const uint64_t *block, const int32_t *kind, const int64_t *row)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    if (ncommit <= 0) return 0;
    if (e->npending && flush_commits(e)) return -1;
    if (ncommit > KCAP) FAIL(e, "too many commits");
    for (int c = 0; c < ncommit; ++c) {
        e->pend_slot[c] = (int32_t) (block[c] & (KCAP - 1));
        e->pend_kind[c] = kind[c];
        e->pend_row[c] = row[c];
    }
    e->npending = ncommit;
    return 0;
}

static int op_read_slot(void *ve, uint64_t block, int kind, double *x)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const double *src = (kind == 1 ? e->d_TX : e->d_TM) + (size_t) (block & (KCAP - 1)) * (size_t) e->ld;
    if (e->shchain) {                                   /* whole points on every rank: no exchange */
        const double *whole = (kind == 1 ? e->d_TX : e->d_TM) + (size_t) (block & (KCAP - 1)) * (size_t) e->ldf;
        CK(e, nla_memcpy_d2h(x, whole, sizeof(double) * (size_t) e->n, e->main));
        CK(e, nla_stream_sync(e->main));
        return 0;
    }
    if (flush_commits(e)) return -1;
    if (e->sharded) return gather_point(e, src, x);
    CK(e, nla_memcpy_d2h(x, src, sizeof(double) * (size_t) e->n, e->main));
    CK(e, nla_stream_sync(e->main));
    return 0;
}

static int op_read_row(void *ve, int64_t row, double *x)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    if (flush_commits(e)) return -1;
    if (e->sharded) return gather_point(e, e->d_X + (size_t) row * (size_t) e->ld, x);
    CK(e, nla_memcpy_d2h(x, e->d_X + (size_t) row * (size_t) e->ld, sizeof(double) * (size_t) e->n, e->main));
    CK(e, nla_stream_sync(e->main));
    return 0;
}

static int op_mutate_slot(void *ve, uint64_t block, int64_t i0)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const uint32_t ring = 2u * (uint32_t) e->B;
    if (ensure_blocks(e, block, block + 1) || flush_commits(e)) return -1;
    CK(e, nla_k_crs_mutate(e->n, e->d_X + (size_t) i0 * (size_t) e->ld, e->d_TX + (size_t) (block & (KCAP - 1)) * (size_t) e->ld,
                           e->d_words + (size_t) ((block + 1) % ring) * 2 * (size_t) e->n, e->d_lb, e->d_ub, e->main));
    return 0;
}

/* window mode: the staged commits (whole points -> rows of the slice), in launches of at most NLA_KARG_MAX; the first one also clears
 * `zero` and refreshes the whole best row from a slot */
static int commit_sh(nla_crs_hip_engine *e, void *zero, size_t zero_bytes, int best_slot, int best_kind)
{
    int done = 0, first = 1;
    while (first || done < e->npending) {
        const int cnt = e->npending - done < NLA_KARG_MAX ? e->npending - done : NLA_KARG_MAX;
        CK(e, nla_k_crs_commit_sh(e->nc, e->ld, e->ldf, e->c0, e->d_X, e->d_TX, e->d_TM, cnt, e->pend_slot + done, e->pend_kind + done, e->pend_row + done,
                                  first ? zero : NULL, first ? zero_bytes : 0, e->n, first ? best_slot : -1, best_kind, e->d_xbest, e->main));
        done += cnt; first = 0;
    }
    e->npending = 0;
    return 0;
}

/* a whole window of a column-sharded population, resolved on the device: no collective call — the slices travel inside the launch */
static int op_chain_sh(nla_crs_hip_engine *e, uint64_t first_block, int K, int64_t i0, double f_best, const int64_t *W, const double *Wf, int nW,
                       nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap)
{
    const int n = e->n;
    const uint32_t ring = 2u * (uint32_t) e->B;
    const int on_host = nW <= NLA_KARG_MAX;
    const int64_t *d_W = W;
    const double *d_Wf = Wf;
    const size_t zero_bytes = nla_crs_chain_ctrl_bytes(K, nW) - sizeof(uint32_t);
    void *zero = (char *) e->d_ctrl + sizeof(uint32_t);
    int best_slot = -1, best_kind = 0;
    /* the whole best row: every trial starts from it (the slice in X serves that), every mutation is formed around it (that needs all of
     * it).  A new best point is an accepted trial point or mutation of the window before — whole in this rank's TX / TM, named by the
     * staged commits; only the first window's best (a row of the initial population) has to be put together from the ranks' slices */
    if (i0 != e->xbest_row) {
        for (int c = e->npending - 1; c >= 0; --c)
            if (e->pend_row[c] == i0) { best_slot = e->pend_slot[c]; best_kind = e->pend_kind[c]; break; }
        if (best_slot < 0) {
            if (flush_commits(e)) return -1;
            CK(e, nla_memcpy_d2d(e->d_gsend, e->d_X + (size_t) i0 * (size_t) e->ld, sizeof(double) * (size_t) e->nc, e->main));
            if (nla_comm_allgather_dev(e->comm, e->d_gsend, e->d_grecv, sizeof(double) * (size_t) e->colper, e->main)) {
                nla_comm_abort(e->comm);
                FAIL(e, "all-gather of the best row's slices failed: %s", nlopt_amd_comm_error(e->comm));
            }
            CK(e, nla_memcpy_d2d(e->d_xbest, e->d_grecv, sizeof(double) * (size_t) n, e->main));     /* (rank r's colper columns sit at r * colper = c0 of rank r) */
        }
        e->xbest_row = i0;
    }
    if (!on_host) {
        memcpy(e->h_up, W, 8 * (size_t) nW); memcpy(e->h_up + 8 * (size_t) nW, Wf, 8 * (size_t) nW);
        CK(e, nla_memcpy_h2d(e->d_up, e->h_up, 16 * (size_t) nW, e->main));
        d_W = (const int64_t *) e->d_up; d_Wf = (const double *) (e->d_up + 8 * (size_t) nW);
    }
    if (commit_sh(e, zero, zero_bytes, best_slot, best_kind)) return -1;
    e->timed = e->stats != NULL;
    if (e->timed) CK(e, nla_event_record(e->ev0, e->main));
    ++e->sh_seq;
    {
        const int rc = nla_k_crs_chain_sh(OBJK(e), n, e->ncolp, e->ld, e->ldf, e->d_X, i0, f_best, e->d_jn, e->d_pos, e->d_last, e->d_words, ring, first_block, K,
                                          d_W, d_Wf, nW, on_host, KCAP - 1, e->d_lb, e->d_ub, e->d_TX, e->d_TM, e->d_ctrl, e->ticket_base, e->h_status,
                                          e->h_fwcnt, e->h_fwrec, fwcap, 1, e->d_table, e->sh_seq, (uint32_t) ((e->stop_in[0] ? 1 : 0) | (e->stop_in[1] ? 2 : 0)), e->sh_grid_cap, e->main);
        if (rc) FAIL(e, "nla_k_crs_chain_sh failed: %s", nla_dev_error_string(rc));
    }
    e->ticket_base += nla_crs_chain_sh_tickets(n, e->ncolp, K, e->sh_grid_cap);
    if (e->timed) CK(e, nla_event_record(e->ev1, e->main));
    if (e->idle_fn) e->idle_fn(e->idle_arg);
    CK(e, nla_stream_sync(e->main));
    if (e->h_status[K].t != 0) FAIL(e, "a rank's share of a trial point did not arrive within 1.5 s (column-sharded window %u)", e->sh_seq);
    e->stop_out[0] = e->h_status[K].fT != 0.; e->stop_out[1] = e->h_status[K].fM != 0.;
    memcpy(status, e->h_status, sizeof(nla_crs_slot_status) * (size_t) K);
    memcpy(fwcnt, e->h_fwcnt, sizeof(uint32_t) * (size_t) K);
    for (int a = 0; a < K; ++a) {
        const uint32_t c = e->h_fwcnt[a] < (uint32_t) fwcap ? e->h_fwcnt[a] : (uint32_t) fwcap;
        if (c) memcpy(fwrec + (size_t) a * (size_t) fwcap, e->h_fwrec + (size_t) a * (size_t) fwcap, sizeof(uint32_t) * (size_t) c);
    }
    for (int a = 0; a < K; ++a) e->h_t[(first_block + (uint64_t) a) & (KCAP - 1)] = n;
    if (e->timed) {
        float ms = nla_event_elapsed_ms(e->ev0, e->ev1);
        if (ms >= 0) e->stats->t_gather_ms += ms;
        e->stats->gather_launches += 1;
        /* what this rank pushed to the others inside the launch: its columns of every slot, to world - 1 ranks */
        e->stats->allgather_bytes += (uint64_t) K * (uint64_t) (e->world - 1) * (uint64_t) e->ncolp * sizeof(double);
    }
    return 0;
}

/* a whole window in one launch with the chain resolved on the device (hip/crs_chain.hip) */
static int op_chain(void *ve, uint64_t first_block, int K, int64_t i0, double f_best, const int64_t *W, const double *Wf, int nW,
                    nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const int n = e->n;
    const uint32_t ring = 2u * (uint32_t) e->B;
    const int on_host = nW <= NLA_KARG_MAX && !e->force_upload;
    const int64_t *d_W = W;
    const double *d_Wf = Wf;
    if (e->obj < 0 || !e->d_ctrl) FAIL(e, "no device-resolved windows for a host objective");
    if (K < 1 || K > CHAIN_KMAX || nW < 0 || nW > CHAIN_KMAX || fwcap != CHAIN_FWCAP) FAIL(e, "bad window K=%d nW=%d", K, nW);
    if (K > op_max_slots(ve, first_block)) FAIL(e, "window reaches past the prepared batches");
    if (ensure_blocks(e, first_block, first_block + (uint64_t) K)) { if (!e->err[0]) snprintf(e->err, sizeof e->err, "batch preparation failed"); return -1; }
    if (e->shchain) return op_chain_sh(e, first_block, K, i0, f_best, W, Wf, nW, status, fwcnt, fwrec, fwcap);
    /* what goes in front of the window on the stream (round 5, "lean windows": a 256-slot window used to be copy / commit / copy / fill /
     * fill / chain — six operations with 4-7 us between any two, profiles/r05_n512_timeline.txt): ONE copy with everything the device
     * needs (W, its f values, the commit lists — none at all when they fit the kernel arguments), the commit kernel of the previous
     * window's accepted points, which also clears the control block, then the window */
    {
        const size_t zero_bytes = nla_crs_chain_ctrl_bytes(K, nW) - sizeof(uint32_t);
        void *zero = (char *) e->d_ctrl + sizeof(uint32_t);
        int zeroed = 0;
        if (e->npending && e->npending <= NLA_KARG_MAX && !e->force_upload) {
            CK(e, nla_k_crs_commit_zero(n, e->ld, e->d_X, e->d_TX, e->d_TM, e->npending, e->pend_slot, e->pend_kind, e->pend_row, 1, zero, zero_bytes, e->main));
            e->npending = 0; zeroed = 1;
        }
        if (e->npending || !on_host) {
            const int nc = e->npending, nWu = on_host ? 0 : nW;
            const size_t oW = 0, oF = oW + 8 * (size_t) nWu, oR = oF + 8 * (size_t) nWu, oS = oR + 8 * (size_t) nc, oK = oS + 4 * (size_t) nc;
            const size_t total = oK + 4 * (size_t) nc;
            if (nWu) { memcpy(e->h_up + oW, W, 8 * (size_t) nWu); memcpy(e->h_up + oF, Wf, 8 * (size_t) nWu); }
            if (nc) {
                memcpy(e->h_up + oR, e->pend_row, 8 * (size_t) nc);
                memcpy(e->h_up + oS, e->pend_slot, 4 * (size_t) nc);
                memcpy(e->h_up + oK, e->pend_kind, 4 * (size_t) nc);
            }
            CK(e, nla_memcpy_h2d(e->d_up, e->h_up, total, e->main));
            if (nc) {
                CK(e, nla_k_crs_commit_zero(e->ncopy, e->ld, e->d_X, e->d_TX, e->d_TM, nc, (const int32_t *) (e->d_up + oS), (const int32_t *) (e->d_up + oK),
                                            (const int64_t *) (e->d_up + oR), 0, zero, zero_bytes, e->main));
                e->npending = 0; zeroed = 1;
            }
            if (nWu) { d_W = (const int64_t *) (e->d_up + oW); d_Wf = (const double *) (e->d_up + oF); }
        }
        /* the event pair around the launch (the roofline figure of bench.py) on every window from n = 2048 on, on one window in
         * TIME_EVERY_WINDOW below: two barrier packets and an elapsed-time query per ~200 us window are a few per cent there */
        e->timed = (n >= 2048 || e->pass_log || (e->pass_no++ % TIME_EVERY_WINDOW) == 0) && e->stats;
        if (e->timed) CK(e, nla_event_record(e->ev0, e->main));
        {
            const int rc = nla_k_crs_chain_lean(OBJK(e), n, e->ld, e->d_X, i0, f_best, e->d_jn, e->d_pos, e->d_last, e->d_words, ring, first_block, K, d_W,
                                                d_Wf, nW, on_host, KCAP - 1, e->d_lb, e->d_ub, e->d_TX, e->d_TM, e->d_ctrl, e->ticket_base, e->h_status,
                                                e->h_fwcnt, e->h_fwrec, fwcap, zeroed, e->main);
            if (rc) FAIL(e, "nla_k_crs_chain failed: %s", nla_dev_error_string(rc));
        }
    }
    e->ticket_base += nla_crs_chain_tickets(n, e->ld, K);
    if (e->timed) CK(e, nla_event_record(e->ev1, e->main));
    if (e->idle_fn) e->idle_fn(e->idle_arg);      /* the window is with the device: the driver's upkeep of its ordered set runs beside it */
    CK(e, nla_stream_sync(e->main));              /* status and records were written into pinned host memory by the kernel */
    memcpy(status, e->h_status, sizeof(nla_crs_slot_status) * (size_t) K);
    memcpy(fwcnt, e->h_fwcnt, sizeof(uint32_t) * (size_t) K);
    /* only the records a slot wrote (most slots of a window read none or one of the window's worst rows): the buffer is pinned memory the
     * device has just written — every line of it is a miss for the host — and the driver reads fwrec[a][0 .. min(fwcnt[a], fwcap)) only */
    for (int a = 0; a < K; ++a) {
        const uint32_t c = e->h_fwcnt[a] < (uint32_t) fwcap ? e->h_fwcnt[a] : (uint32_t) fwcap;
        if (c) memcpy(fwrec + (size_t) a * (size_t) fwcap, e->h_fwrec + (size_t) a * (size_t) fwcap, sizeof(uint32_t) * (size_t) c);
    }
    for (int a = 0; a < K; ++a) e->h_t[(first_block + (uint64_t) a) & (KCAP - 1)] = n;
    if (e->timed) {
        float ms = nla_event_elapsed_ms(e->ev0, e->ev1);
        if (ms >= 0) e->stats->t_gather_ms += ms;
        e->stats->gather_launches += 1;
        if (e->pass_log) fprintf(e->pass_log, "%d,%d,%d,%d,%d,%d,%ld,%.4f\n", n, K, nW, 0, K, 0, (long) K * n, (double) ms);
    }
    return 0;
}

static const char *op_last_error(void *ve) { return ((nla_crs_hip_engine *) ve)->err; }
static void op_stop_flags_in(void *ve, int forced, int timed) { nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve; e->stop_in[0] = forced != 0; e->stop_in[1] = timed != 0; }
static void op_stop_flags_out(void *ve, int *forced, int *timed) { nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve; *forced = e->stop_out[0]; *timed = e->stop_out[1]; }

static void op_set_idle(void *ve, void (*fn)(void *), void *arg) { nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve; e->idle_fn = fn; e->idle_arg = arg; }

static int op_reset_slot(void *ve, uint64_t block)
{
    ((nla_crs_hip_engine *) ve)->h_t[block & (KCAP - 1)] = 0;
    return 0;
}

const nla_crs_engine_ops nla_crs_hip_ops = {
    op_init_population, op_max_slots, op_advance, op_chain, op_reset_slot, op_commit, op_read_slot, op_read_row, op_mutate_slot, op_last_error,
    op_stop_flags_in, op_stop_flags_out, op_set_idle
};

static nlopt_result crs_open_common(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub, const double *x,
                                    nla_stopping *stop, int population, nla_crs_problem *pb, nla_crs_hip_engine **eout)
{
    int64_t N = population ? (int64_t) population : 10 * ((int64_t) n + 1);     /* crs.c:172-179 */
    nla_userobj *user = NULL;
    double sign = 1.;
    *eout = NULL;
    if (N < n + 1) {                                                            /* crs.c:180-184 */
        nla_stop_msg(stop, "population %d should be >= dimension + 1 = %d", (int) N, n + 1);
        return NLOPT_INVALID_ARGS;
    }
    memset(pb, 0, sizeof *pb);
    pb->forward = n >= NLA_CRS_FORWARD_MIN_N;
    pb->n = n; pb->N = N; pb->lb = lb; pb->ub = ub; pb->f = f; pb->f_data = f_data; pb->stop = stop;
    {
        nla_evaluator ev;
        nla_evaluator_resolve(&ev, opt, f, f_data);
        pb->obj = ev.kind == NLA_EVAL_DEVICE ? ev.obj : (ev.kind == NLA_EVAL_USER ? -2 : -1);
        user = ev.user; sign = ev.sign;
    }
    if (opt) {
        pb->trace = opt->trace; pb->trace_cap = opt->trace_cap; pb->trace_len = &opt->trace_len;
        pb->stats = &opt->stats;
        pb->max_spec = (int) nlopt_get_param(opt, "amd_max_spec", 0);
        pb->window_factor = nlopt_get_param(opt, "amd_window_factor", 0);
        /* device-resolved windows (hip/crs_chain.hip, the chain advanced by the resolver wavefront) at every dimension.  Measured on the
         * MI355X, N = 1e5 (profiles/r05_staged_ab.txt; conservative passes -> windows -> windows of 256 slots): n = 64 1.26 -> 1.44 -> 1.56 M
         * evals/s, n = 128 1.07 -> 1.32 -> 1.39 M, n = 256 0.75 -> 1.03 -> 1.08 M, n = 512 (round 4) 445 -> 606 k */
        pb->forward = nlopt_get_param(opt, "amd_forward", n >= NLA_CRS_FORWARD_MIN_N ? 1 : 0) != 0;
        pb->forward = nla_dbg_int("NLA_CRS_FORWARD", pb->forward);                 /* A/B switch for the bench */
        if (nlopt_get_param(opt, "amd_host_eval", 0) != 0) pb->obj = -1;   /* force the host-callback path */
    }
    if (nla_dev_count() <= 0) {
        nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)");
        nla_comm_agree_ready(opt ? opt->comm : NULL, 0);
        return NLOPT_FAILURE;
    }
    {
        /* several ranks and a compiled-in objective: the population is sharded by coordinate and the window runs as conservative
         * passes on every rank's slice ("amd_shard" = 0: every rank keeps the whole population — replicas of the one chain) */
        nlopt_amd_comm *comm = opt ? opt->comm : NULL;
        const int shard = comm && nla_crs_can_shard(n, nlopt_amd_comm_world(comm)) && pb->obj >= 0 &&
                          (!opt || nlopt_get_param(opt, "amd_shard", 1) != 0);
        /* ... or, where the coordinates can be dealt in whole 128-byte lines over at most 8 ranks, as device-resolved windows whose slices
         * cross between the ranks' kernels through peer-mapped memory ("amd_shard_windows" = 0: the conservative passes;
         * "amd_cu_share" = k: k ranks share one device — each rank's windows run on its k-th of the compute units) */
        const int windows = shard && pb->forward && pb->max_spec != 1 /* (a window of one slot: the driver runs conservative passes) */ &&
                            nla_crs_can_shard_windows(n, nlopt_amd_comm_world(comm)) &&
                            (!opt || nlopt_get_param(opt, "amd_shard_windows", 1) != 0);
        const int cu_parts = opt ? (int) nlopt_get_param(opt, "amd_cu_share", 0) : 0;
        if (shard) { pb->comm = comm; if (!windows) pb->forward = 0; }
        *eout = nla_crs_hip_engine_create(n, N, lb, ub, pb->obj, pb->forward, comm, windows ? 2 : shard, windows || !comm ? cu_parts : 0, pb->stats, NULL);
        if (windows) {
            /* all ranks run windows, or all fall back to the conservative passes (a rank that could not map a peer's buffers, an
             * allocation that failed): the same decision everywhere */
            const int all = nla_comm_agree_ready(comm, *eout != NULL);
            if (!all) {
                if (*eout) nla_crs_hip_engine_destroy(*eout, 0);
                pb->forward = 0;
                *eout = nla_crs_hip_engine_create(n, N, lb, ub, pb->obj, 0, comm, 1, 0, pb->stats, NULL);
            }
        }
        if (*eout) { (*eout)->fuse_commit = 1; (*eout)->doorbell = 1; }      /* (rounds 4-5 had A/B switches here: profiles/r04_crs_fuse_commit_ab.txt, r04_crs_doorbell_ab.txt) */
    }
    /* several ranks: all of them go on, or none (a rank that could not set up would leave the others in the first all-gather) */
    if (opt && nlopt_amd_comm_world(opt->comm) > 1) {
        const int all = nla_comm_agree_same(opt->comm, *eout != NULL, nla_problem_fingerprint(NLOPT_GN_CRS2_LM, n, (int) N, pb->obj, lb, ub, x, stop) + nla_params_fingerprint(opt));
        if (all <= 0 && *eout) {
            nla_crs_hip_engine_destroy(*eout, 0); *eout = NULL;
            if (all < 0) { nla_stop_msg(stop, NLA_MSG_RANKS_DIFFER); return NLOPT_INVALID_ARGS; }
            nla_stop_msg(stop, "nlopt_amd: another rank could not set up its device engine");
            return NLOPT_FAILURE;
        }
    }
    if (!*eout) {
        nla_stop_msg(stop, "nlopt_amd: could not create the device engine (out of device memory?)");
        return NLOPT_OUT_OF_MEMORY;
    }
    (*eout)->user = user; (*eout)->sign = sign;
    (*eout)->variant = 0;                          /* automatic tiling of the conservative passes' gather (nla_k_crs_advance) */
    return NLOPT_SUCCESS;
}

/* reference-shaped entry point: crs_minimize(n, f, f_data, lb, ub, x, minf, stop, population, lds=0)
 * (src/algs/crs/crs.h:34-40; called from the dispatcher as at src/api/optimize.c:744-747) */
nlopt_result nla_crs_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub,
                              double *x, double *minf, nla_stopping *stop, int population)
{
    nla_crs_problem pb;
    nla_crs_hip_engine *e;
    nlopt_result ret;
    uint64_t words = 0;
    ret = crs_open_common(opt, n, f, f_data, lb, ub, x, stop, population, &pb, &e);
    if (ret != NLOPT_SUCCESS) return ret;
    ret = nla_crs_run(&nla_crs_hip_ops, e, &pb, x, minf, &words);
    if (pb.stats) pb.stats->mt_words = words;
    nla_crs_hip_engine_destroy(e, words);
    return ret;
}

/* ---- stepwise sessions (nlopt_amd.h): the same run, paused between speculation rounds ---------- */
struct nlopt_amd_crs_session {
    nlopt_opt opt;
    nla_stopping stop;
    nla_crs_problem pb;
    nla_crs_hip_engine *e;
    nla_crs_session *S;
    nlopt_result ret;
};

nlopt_result nla_setup_run(nlopt_opt opt, double *x, double *minf, nla_stopping *stop);

nlopt_amd_crs_session *nlopt_amd_crs_open(nlopt_opt opt, double *x, double *minf, nlopt_result *ret_out)
{
    nlopt_amd_crs_session *h;
    nlopt_result ret;
    int pop;
    nla_unset_errmsg(opt);
    if (!opt || opt->algorithm != NLOPT_GN_CRS2_LM || opt->maximize) { if (ret_out) *ret_out = NLOPT_INVALID_ARGS; return NULL; }
    h = (nlopt_amd_crs_session *) calloc(1, sizeof *h);
    if (!h) { if (ret_out) *ret_out = NLOPT_OUT_OF_MEMORY; return NULL; }
    h->opt = opt;
    nlopt_set_force_stop(opt, 0);
    ret = nla_setup_run(opt, x, minf, &h->stop);
    if (ret == NLOPT_SUCCESS) {
        pop = opt->stochastic_population > 0 ? (int) opt->stochastic_population
                                             : (nla_stochastic_population > 0 ? nla_stochastic_population : 0);
        ret = crs_open_common(opt, (int) opt->n, opt->f, opt->f_data, opt->lb, opt->ub, x, &h->stop, pop, &h->pb, &h->e);
    }
    if (ret != NLOPT_SUCCESS) { if (ret_out) *ret_out = ret; free(h); return NULL; }
    h->S = nla_crs_begin(&nla_crs_hip_ops, h->e, &h->pb, x, minf, &ret);
    h->ret = ret;
    if (ret_out) *ret_out = ret;
    if (!h->S) { nla_crs_hip_engine_destroy(h->e, 0); free(h); return NULL; }
    return h;
}

nlopt_result nlopt_amd_crs_step(nlopt_amd_crs_session *h, long eval_budget)
{
    if (!h || !h->S) return NLOPT_INVALID_ARGS;
    if (h->ret == NLOPT_SUCCESS) h->ret = nla_crs_advance(h->S, (int64_t) eval_budget);
    return h->ret;
}

nlopt_result nlopt_amd_crs_close(nlopt_amd_crs_session *h)
{
    nlopt_result ret;
    uint64_t words = 0;
    if (!h) return NLOPT_INVALID_ARGS;
    ret = nla_crs_end(h->S, &words);
    if (h->pb.stats) h->pb.stats->mt_words = words;
    nla_crs_hip_engine_destroy(h->e, words);
    free(h);
    return ret;
}
