/* crs_engine.c — the HIP engine behind the CRS2_LM driver: device memory, streams and kernel
 * sequencing.  Plain C over the C-ABI launchers of include/nlopt_amd.h (no HIP types here).
 *
 * Memory (all HBM, sized for one MI355X; N=1e5, n=4096 is 3.3 GB, N=1e6 33 GB of 288 GB):
 *   X        N x ld fp64 population                 (the reference's ps matrix without the f column)
 *   batch[2] stream blocks pre-digested ahead of use: words (B+1 blocks x 2n u32),
 *            jn/last (B i32), pos (B x n i32)        — double-buffered, filled on the rng stream
 *   TX, TM   Kcap x ld fp64 speculative trial / mutation points, fT/fM/minhz their results
 * Streams: `main` runs gather -> post -> D2H -> commit each round; `rng` runs the MT generator,
 * the GF(2) jumps and the Vitter kernel for the *next* batch of blocks concurrently (VALU-bound
 * work hiding under the HBM-bound gather); an event per batch orders main after rng.
 */
#include "nla_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define INIT_CHUNK_WORDS (1ULL << 28)      /* 1 GiB of stream words per init pass */
#define KCAP 1024

typedef struct {
    int64_t index;                 /* batch number (blocks [index*B, (index+1)*B)), -1 = empty */
    uint32_t *d_words;             /* (B+1) * 2n */
    int32_t *d_jn, *d_pos, *d_last;
    void *ev_ready;
    int waited;                    /* main stream already ordered after ev_ready */
} crs_batch;

struct nla_crs_hip_engine {
    int n, ld, obj;
    int64_t N;
    int B;                         /* blocks per batch */
    void *main, *rng;
    nla_mtstream *mts;
    double *d_lb, *d_ub, *d_X, *d_F;
    uint32_t *d_initwords; size_t initwords_cap;
    crs_batch bat[2];
    double *d_TX, *d_TM, *d_fT, *d_fM;
    int32_t *d_minhz, *d_cslot, *d_ckind;
    int64_t *d_W, *d_crow;
    /* pinned staging */
    double *h_f;                   /* fT[KCAP] | fM[KCAP] */
    int32_t *h_minhz, *h_cslot, *h_ckind;
    int64_t *h_W, *h_crow;
    void *ev0, *ev1;
    nlopt_amd_stats *stats;
    int lastK;
    char err[256];
};

#define FAIL(e, ...) do { snprintf((e)->err, sizeof (e)->err, __VA_ARGS__); return -1; } while (0)
#define CK(e, call) do { int rc_ = (call); if (rc_) FAIL(e, "%s failed: %s", #call, nla_dev_error_string(rc_)); } while (0)

static uint64_t trial_word0(const nla_crs_hip_engine *e) { return 2ULL * (uint64_t) e->n * (uint64_t) (e->N - 1); }

static int prepare_batch(nla_crs_hip_engine *e, crs_batch *b, int64_t index)
{
    const uint64_t w0 = trial_word0(e) + 2ULL * (uint64_t) e->n * (uint64_t) index * (uint64_t) e->B;
    b->index = index;
    b->waited = 0;
    if (nla_mtstream_fill(e->mts, w0, 2ULL * (uint64_t) e->n * (uint64_t) (e->B + 1), b->d_words))
        FAIL(e, "MT stream fill failed for batch %lld", (long long) index);
    CK(e, nla_k_crs_vitter(e->n, e->N, b->d_words, e->B, b->d_jn, b->d_pos, b->d_last, e->rng));
    CK(e, nla_event_record(b->ev_ready, e->rng));
    return 0;
}

static crs_batch *batch_for(nla_crs_hip_engine *e, uint64_t block)
{
    const int64_t index = (int64_t) (block / (uint64_t) e->B);
    crs_batch *cur = &e->bat[index & 1], *nxt = &e->bat[(index + 1) & 1];
    if (cur->index != index && prepare_batch(e, cur, index)) return NULL;
    /* prefetch the following batch: everything that used the other buffer has been synchronised */
    if (nxt->index != index + 1 && prepare_batch(e, nxt, index + 1)) return NULL;
    if (!cur->waited) {
        if (nla_stream_wait_event(e->main, cur->ev_ready)) return NULL;
        cur->waited = 1;
    }
    return cur;
}

void nla_crs_hip_engine_destroy(nla_crs_hip_engine *e, uint64_t words_used)
{
    if (!e) return;
    if (e->main) nla_stream_sync(e->main);
    if (e->rng) nla_stream_sync(e->rng);
    if (e->mts) { nla_mtstream_finish(e->mts, words_used); nla_mtstream_destroy(e->mts); }
    for (int i = 0; i < 2; ++i) {
        nla_dev_free(e->bat[i].d_words); nla_dev_free(e->bat[i].d_jn); nla_dev_free(e->bat[i].d_pos);
        nla_dev_free(e->bat[i].d_last); nla_event_destroy(e->bat[i].ev_ready);
    }
    nla_dev_free(e->d_lb); nla_dev_free(e->d_ub); nla_dev_free(e->d_X); nla_dev_free(e->d_F);
    nla_dev_free(e->d_initwords);
    nla_dev_free(e->d_TX); nla_dev_free(e->d_TM); nla_dev_free(e->d_fT);   /* d_fM aliases d_fT + KCAP */
    nla_dev_free(e->d_minhz); nla_dev_free(e->d_cslot); nla_dev_free(e->d_ckind); nla_dev_free(e->d_W); nla_dev_free(e->d_crow);
    nla_host_free(e->h_f); nla_host_free(e->h_minhz); nla_host_free(e->h_cslot); nla_host_free(e->h_ckind);
    nla_host_free(e->h_W); nla_host_free(e->h_crow);
    nla_event_destroy(e->ev0); nla_event_destroy(e->ev1);
    nla_stream_destroy(e->main); nla_stream_destroy(e->rng);
    free(e);
}

nla_crs_hip_engine *nla_crs_hip_engine_create(int n, int64_t N, const double *lb, const double *ub, int obj,
                                              nlopt_amd_stats *stats, char **errmsg)
{
    nla_crs_hip_engine *e;
    size_t B;
    (void) errmsg;
    if (nla_dev_count() <= 0) return NULL;
    e = (nla_crs_hip_engine *) calloc(1, sizeof *e);
    if (!e) return NULL;
    e->n = n; e->N = N; e->obj = obj; e->stats = stats;
    e->ld = (n + 1) & ~1;
    e->bat[0].index = e->bat[1].index = -1;
    B = (size_t) ((1ULL << 25) / (2ULL * (uint64_t) n));
    if (B > 65536) B = 65536;
    if (B < 1024) B = 1024;
    e->B = (int) B;
    e->main = nla_stream_create();
    e->rng = nla_stream_create();
    if (!e->main || !e->rng) goto fail;
    e->mts = nla_mtstream_create(e->rng);
    if (!e->mts) goto fail;
    e->d_lb = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld);
    e->d_ub = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld);
    e->d_X = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld * (size_t) N);
    e->d_F = (double *) nla_dev_malloc(sizeof(double) * (size_t) N);
    for (int i = 0; i < 2; ++i) {
        e->bat[i].d_words = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * 2 * (size_t) n * (B + 1));
        e->bat[i].d_jn = (int32_t *) nla_dev_malloc(sizeof(int32_t) * B);
        e->bat[i].d_last = (int32_t *) nla_dev_malloc(sizeof(int32_t) * B);
        e->bat[i].d_pos = (int32_t *) nla_dev_malloc(sizeof(int32_t) * B * (size_t) n);
        e->bat[i].ev_ready = nla_event_create();
        if (!e->bat[i].d_words || !e->bat[i].d_jn || !e->bat[i].d_last || !e->bat[i].d_pos || !e->bat[i].ev_ready) goto fail;
    }
    e->d_TX = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld * KCAP);
    e->d_TM = (double *) nla_dev_malloc(sizeof(double) * (size_t) e->ld * KCAP);
    e->d_fT = (double *) nla_dev_malloc(sizeof(double) * 2 * KCAP);
    e->d_fM = e->d_fT ? e->d_fT + KCAP : NULL;
    e->d_minhz = (int32_t *) nla_dev_malloc(sizeof(int32_t) * KCAP);
    e->d_cslot = (int32_t *) nla_dev_malloc(sizeof(int32_t) * KCAP);
    e->d_ckind = (int32_t *) nla_dev_malloc(sizeof(int32_t) * KCAP);
    e->d_W = (int64_t *) nla_dev_malloc(sizeof(int64_t) * KCAP);
    e->d_crow = (int64_t *) nla_dev_malloc(sizeof(int64_t) * KCAP);
    e->h_f = (double *) nla_host_malloc(sizeof(double) * 2 * KCAP);
    e->h_minhz = (int32_t *) nla_host_malloc(sizeof(int32_t) * KCAP);
    e->h_cslot = (int32_t *) nla_host_malloc(sizeof(int32_t) * KCAP);
    e->h_ckind = (int32_t *) nla_host_malloc(sizeof(int32_t) * KCAP);
    e->h_W = (int64_t *) nla_host_malloc(sizeof(int64_t) * KCAP);
    e->h_crow = (int64_t *) nla_host_malloc(sizeof(int64_t) * KCAP);
    e->ev0 = nla_event_create();
    e->ev1 = nla_event_create();
    if (!e->d_lb || !e->d_ub || !e->d_X || !e->d_F || !e->d_TX || !e->d_TM || !e->d_fT || !e->d_minhz || !e->d_cslot ||
        !e->d_ckind || !e->d_W || !e->d_crow || !e->h_f || !e->h_minhz || !e->h_cslot || !e->h_ckind || !e->h_W ||
        !e->h_crow || !e->ev0 || !e->ev1) goto fail;
    if (nla_memcpy_h2d(e->d_lb, lb, sizeof(double) * (size_t) n, e->main) ||
        nla_memcpy_h2d(e->d_ub, ub, sizeof(double) * (size_t) n, e->main) || nla_stream_sync(e->main)) goto fail;
    return e;
fail:
    nla_crs_hip_engine_destroy(e, 0);
    return NULL;
}

/* ---- ops ------------------------------------------------------------------------------------ */
static int op_init_population(void *ve, const double *x0, double *F)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const int n = e->n;
    const uint64_t wpr = 2ULL * (uint64_t) n;            /* words per row */
    int64_t rows_per_chunk = (int64_t) (INIT_CHUNK_WORDS / wpr), r0;
    void *ev = nla_event_create();
    if (!ev) FAIL(e, "event create failed");
    if (rows_per_chunk < 1) rows_per_chunk = 1;
    if (rows_per_chunk > e->N - 1) rows_per_chunk = e->N - 1;
    if (e->N > 1) {
        size_t need = (size_t) (rows_per_chunk * (int64_t) wpr);
        if (need > e->initwords_cap) {
            nla_dev_free(e->d_initwords);
            e->d_initwords = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * need);
            e->initwords_cap = e->d_initwords ? need : 0;
            if (!e->d_initwords) { nla_event_destroy(ev); FAIL(e, "out of device memory (init words)"); }
        }
    }
    /* row 0 = the caller's starting guess (crs.c:204) */
    if (nla_memcpy_h2d(e->d_X, x0, sizeof(double) * (size_t) n, e->main)) { nla_event_destroy(ev); FAIL(e, "H2D x0 failed"); }
    if (e->obj >= 0 && nla_k_eval(e->obj, n, e->ld, e->d_X, 1, e->d_F, e->main)) { nla_event_destroy(ev); FAIL(e, "eval launch failed"); }
    for (r0 = 1; r0 < e->N; r0 += rows_per_chunk) {
        int64_t nr = e->N - r0 < rows_per_chunk ? e->N - r0 : rows_per_chunk;
        /* the words buffer is reused: the generator must not overwrite it before the previous
         * chunk's init kernel has consumed it */
        if (r0 > 1) {
            if (nla_event_record(ev, e->main) || nla_stream_wait_event(e->rng, ev)) { nla_event_destroy(ev); FAIL(e, "event failed"); }
        }
        if (nla_mtstream_fill(e->mts, wpr * (uint64_t) (r0 - 1), wpr * (uint64_t) nr, e->d_initwords)) {
            nla_event_destroy(ev); FAIL(e, "MT stream fill failed (init)");
        }
        if (nla_event_record(ev, e->rng) || nla_stream_wait_event(e->main, ev)) { nla_event_destroy(ev); FAIL(e, "event failed"); }
        if (nla_k_crs_init_rows(e->obj, n, e->ld, e->d_lb, e->d_ub, e->d_initwords, r0, nr, e->d_X, e->d_F, e->main)) {
            nla_event_destroy(ev); FAIL(e, "init kernel launch failed");
        }
    }
    if (e->obj >= 0) {
        if (nla_memcpy_d2h(F, e->d_F, sizeof(double) * (size_t) e->N, e->main)) { nla_event_destroy(ev); FAIL(e, "D2H F failed"); }
    }
    {
        int rc = nla_stream_sync(e->main);
        nla_event_destroy(ev);
        if (rc) FAIL(e, "init sync failed: %s", nla_dev_error_string(rc));
    }
    nla_dev_free(e->d_initwords); e->d_initwords = NULL; e->initwords_cap = 0;
    /* start digesting the first batch of trial blocks while the host builds its ordered set */
    if (!batch_for(e, 0)) return -1;
    return 0;
}

static int op_max_slots(void *ve, uint64_t first_block)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const uint64_t B = (uint64_t) e->B;
    uint64_t rem = B - first_block % B;
    return (int) (rem < KCAP ? rem : KCAP);
}

static int op_speculate(void *ve, uint64_t first_block, int K, int64_t i0, const int64_t *W, int nW,
                        double *fT, double *fM, int32_t *minhz)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const int n = e->n;
    crs_batch *b = batch_for(e, first_block);
    size_t off;
    if (!b) { if (!e->err[0]) snprintf(e->err, sizeof e->err, "batch preparation failed"); return -1; }
    if (K < 1 || K > KCAP || nW > KCAP) FAIL(e, "bad speculation size K=%d nW=%d", K, nW);
    off = (size_t) (first_block - (uint64_t) b->index * (uint64_t) e->B);
    if (off + (size_t) K > (size_t) e->B) FAIL(e, "speculation crosses a batch boundary");
    memcpy(e->h_W, W, sizeof(int64_t) * (size_t) nW);
    CK(e, nla_memcpy_h2d(e->d_W, e->h_W, sizeof(int64_t) * (size_t) nW, e->main));
    CK(e, nla_event_record(e->ev0, e->main));
    CK(e, nla_k_crs_gather(n, e->ld, e->d_X, i0, b->d_jn + off, b->d_pos + off * (size_t) n, b->d_last + off, K,
                           e->d_lb, e->d_ub, e->d_TX, e->main));
    CK(e, nla_event_record(e->ev1, e->main));
    CK(e, nla_k_crs_post(e->obj, n, e->ld, e->d_X, i0, e->d_TX, e->d_TM, b->d_words + (off + 1) * 2 * (size_t) n, K,
                         e->d_W, nW, b->d_pos + off * (size_t) n, b->d_last + off, e->d_lb, e->d_ub,
                         e->d_fT, e->d_fM, e->d_minhz, e->main));
    if (e->obj >= 0) {
        CK(e, nla_memcpy_d2h(e->h_f, e->d_fT, sizeof(double) * (size_t) K, e->main));
        CK(e, nla_memcpy_d2h(e->h_f + KCAP, e->d_fM, sizeof(double) * (size_t) K, e->main));
    }
    CK(e, nla_memcpy_d2h(e->h_minhz, e->d_minhz, sizeof(int32_t) * (size_t) K, e->main));
    CK(e, nla_stream_sync(e->main));
    if (e->obj >= 0) {
        memcpy(fT, e->h_f, sizeof(double) * (size_t) K);
        memcpy(fM, e->h_f + KCAP, sizeof(double) * (size_t) K);
    }
    memcpy(minhz, e->h_minhz, sizeof(int32_t) * (size_t) K);
    e->lastK = K;
    if (e->stats) {
        float ms = nla_event_elapsed_ms(e->ev0, e->ev1);
        if (ms >= 0) e->stats->t_gather_ms += ms;
        e->stats->gather_launches += 1;
        e->stats->gather_bytes += (uint64_t) K * 8ULL * (uint64_t) n * (uint64_t) (n + 1);
    }
    return 0;
}

static int op_commit(void *ve, int ncommit, const int32_t *slot, const int32_t *kind, const int64_t *row)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    if (ncommit <= 0) return 0;
    if (ncommit > KCAP) FAIL(e, "too many commits");
    memcpy(e->h_cslot, slot, sizeof(int32_t) * (size_t) ncommit);
    memcpy(e->h_ckind, kind, sizeof(int32_t) * (size_t) ncommit);
    memcpy(e->h_crow, row, sizeof(int64_t) * (size_t) ncommit);
    CK(e, nla_memcpy_h2d(e->d_cslot, e->h_cslot, sizeof(int32_t) * (size_t) ncommit, e->main));
    CK(e, nla_memcpy_h2d(e->d_ckind, e->h_ckind, sizeof(int32_t) * (size_t) ncommit, e->main));
    CK(e, nla_memcpy_h2d(e->d_crow, e->h_crow, sizeof(int64_t) * (size_t) ncommit, e->main));
    CK(e, nla_k_crs_commit(e->n, e->ld, e->d_X, e->d_TX, e->d_TM, ncommit, e->d_cslot, e->d_ckind, e->d_crow, e->main));
    return 0;
}

static int op_read_slot(void *ve, int slot, int kind, double *x)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    const double *src = (kind == 1 ? e->d_TX : e->d_TM) + (size_t) slot * (size_t) e->ld;
    CK(e, nla_memcpy_d2h(x, src, sizeof(double) * (size_t) e->n, e->main));
    CK(e, nla_stream_sync(e->main));
    return 0;
}

static int op_read_row(void *ve, int64_t row, double *x)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    CK(e, nla_memcpy_d2h(x, e->d_X + (size_t) row * (size_t) e->ld, sizeof(double) * (size_t) e->n, e->main));
    CK(e, nla_stream_sync(e->main));
    return 0;
}

static int op_mutate_slot(void *ve, int slot, uint64_t block, int64_t i0)
{
    nla_crs_hip_engine *e = (nla_crs_hip_engine *) ve;
    /* `block` is at most one past the batch of the slot's own block: the batch holds B+1 blocks of words */
    crs_batch *b = batch_for(e, block > 0 ? block - 1 : 0);
    size_t off;
    if (!b) return -1;
    off = (size_t) (block - (uint64_t) b->index * (uint64_t) e->B);
    CK(e, nla_k_crs_mutate(e->n, e->d_X + (size_t) i0 * (size_t) e->ld, e->d_TX + (size_t) slot * (size_t) e->ld,
                           b->d_words + off * 2 * (size_t) e->n, e->d_lb, e->d_ub, e->main));
    return 0;
}

static const char *op_last_error(void *ve) { return ((nla_crs_hip_engine *) ve)->err; }

const nla_crs_engine_ops nla_crs_hip_ops = {
    op_init_population, op_max_slots, op_speculate, op_commit, op_read_slot, op_read_row, op_mutate_slot, op_last_error
};

static nlopt_result crs_open_common(nlopt_opt opt, int n, nlopt_func f, void *f_data, const double *lb, const double *ub,
                                    nla_stopping *stop, int population, nla_crs_problem *pb, nla_crs_hip_engine **eout)
{
    int64_t N = population ? (int64_t) population : 10 * ((int64_t) n + 1);     /* crs.c:172-179 */
    *eout = NULL;
    if (N < n + 1) {                                                            /* crs.c:180-184 */
        nla_stop_msg(stop, "population %d should be >= dimension + 1 = %d", (int) N, n + 1);
        return NLOPT_INVALID_ARGS;
    }
    memset(pb, 0, sizeof *pb);
    pb->n = n; pb->N = N; pb->lb = lb; pb->ub = ub; pb->f = f; pb->f_data = f_data; pb->stop = stop;
    pb->obj = nlopt_amd_objective_id(f);
    if (opt) {
        pb->trace = opt->trace; pb->trace_cap = opt->trace_cap; pb->trace_len = &opt->trace_len;
        pb->stats = &opt->stats;
        pb->max_spec = (int) nlopt_get_param(opt, "amd_max_spec", 0);
        if (nlopt_get_param(opt, "amd_host_eval", 0) != 0) pb->obj = -1;   /* force the host-callback path */
    }
    if (nla_dev_count() <= 0) {
        nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)");
        return NLOPT_FAILURE;
    }
    *eout = nla_crs_hip_engine_create(n, N, lb, ub, pb->obj, pb->stats, NULL);
    if (!*eout) {
        nla_stop_msg(stop, "nlopt_amd: could not create the device engine (out of device memory?)");
        return NLOPT_OUT_OF_MEMORY;
    }
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
    ret = crs_open_common(opt, n, f, f_data, lb, ub, stop, population, &pb, &e);
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
        ret = crs_open_common(opt, (int) opt->n, opt->f, opt->f_data, opt->lb, opt->ub, &h->stop, pop, &h->pb, &h->e);
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
