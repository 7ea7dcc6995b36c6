#define _GNU_SOURCE            /* qsort_r */
/* isres_driver.c — NLOPT_GN_ISRES behind the reference's entry point
 *   isres_minimize(n, f, f_data, m, fc, p, h, lb, ub, x, minf, stop, population)   (isres.h:34-41),
 * host side: the generation loop, the best-point/stop bookkeeping the reference runs after EVERY
 * candidate (isres.c:168-198, replayed in candidate order over the device's f/penalty arrays), the
 * stream accounting, and the kernel sequencing through the C-ABI launchers of include/nlopt_amd.h.
 *
 * One generation on the device (isres_kernels.hip):
 *   eval      f + penalties of all pop candidates                       (isres.c:134-166)
 *   rank      all feasible: stable sort by f (counting ranks)            (:204)
 *             else: MT words -> "u < PF" bits -> systolic stochastic ranking; the reference's
 *             early exit after a sweep without exchange (:227) is honoured by checking the
 *             per-sweep flags and, if it ever fires, re-running with exactly that many sweeps
 *   evolve    MT words -> accepted normal deviates in order -> the serial mutation / variation
 *             chain (:234-280)
 * The MT19937 stream position after the run equals the reference's: 2 pop n words for the initial
 * population, 2 per ranking step actually taken, 4 per Box-Muller attempt up to the last deviate
 * consumed.
 *
 * Device objective + device constraints (pointer identity, nlopt_amd.h part 1) run entirely on the
 * GPU.  Multi-GPU (nlopt_amd_set_comm): the candidates are block-partitioned over the
 * ranks for the evaluation and (f, penalty, inequality penalty, feasibility) are ALL-GATHERED — the one
 * exchange a generation needs; selection and evolution are one serial chain through the stream position
 * and run replicated, so every rank holds the whole population and no rows travel.
 * Any other callback takes the host-callback path: X is copied back and f / constraints are
 * called on the caller's thread in the reference's order (:137-165); ranking and evolution still
 * run on the device.  There is no CPU fallback for the device work.
 */
#include "nla_internal.h"
#include "nla_switches.h"
#include "objfuncs.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define WORD_CHUNK_MAX (1ULL << 30)        /* stream words generated per pass: at most 4 GiB of the 288 (~1700 generator wavefronts per launch) */

typedef struct {
    int n, ld, m, p, obj, dev_eval;
    nla_evaluator ev;              /* how the objective is evaluated (compiled-in kernel / user kernel / host callback) */
    int64_t pop, survivors, units, rowwords;
    nlopt_amd_comm *comm;
    uint64_t wchunk;                /* words per generator pass: what the largest phase needs, up to WORD_CHUNK_MAX */
    int64_t per, popcap;            /* candidates per rank, per * world >= pop (all-gather wants equal blocks) */
    int bits_two_pass;              /* A/B switch: ranking words through a buffer + isres_bits_kernel instead of the fused kernel */
    void *st, *ev0, *ev1;
    void *rs;                       /* the stream the generator works on: st itself, or (overlap) a second stream — see "overlap" below */
    int gated;                      /* 1 (one rank, generator on its own stream): the ranking pipeline starts while its bits are still produced */
    int *d_gate;                    /* units + 1 counters (+ the generator's ticket, + the pipeline's "gave up waiting" word): block c of the ranking bits (sweeps 64 c .. 64 c + 63) is complete when d_gate[c] has
                                     * reached its target (hip/mt_kernels.hip: nla_k_mt_rankbits_gated; zeroed on the generator's stream before every launch) */
    void *ev_gate;                  /* recorded behind that zeroing: the pipeline's launch on the main stream waits for it */
    int gen_waves_per_cu;           /* wavefronts of the gated generator a CU holds at a time (0 = as many as fit) */
    int evolve_serial;              /* "amd_isres_evolve_serial" != 0: the one-workgroup evolve kernel (the parallel one's reference in the tests) */
    int overlap;                    /* default; "amd_isres_overlap" = 0 / NLA_ISRES_OVERLAP=0 turn it off: generator work beside the latency-bound kernels */
    int spec_valid; uint64_t spec_word0; int64_t spec_attempts, spec_zcount;   /* deviates generated ahead of the evolve phase (overlap) */
    int bits_pre; uint64_t bits_pre_g0;   /* the gated generator of this generation's ranking bits is already enqueued (in front of the host's bookkeeping loop) */
    uint64_t spec_made, spec_used;  /* ... how often, and how often the evolve phase could take them */
    nla_mtstream *mts;
    uint64_t words_used;
    double *d_lb, *d_ub, *d_X, *d_S, *d_F, *d_PEN, *d_GPEN, *d_scratch, *d_z;
    int32_t *d_FEAS, *d_irank, *d_counts, *d_inv;
    double *d_rho; void *d_ws;           /* parallel evolve (hip/isres_evolve2.hip): redraw statistics, block workspace */
    double *d_mu;                        /* ... and the redraws a child of each parent (by rank position) is expected to make: survivors doubles */
    int parallel_evolve;
    uint64_t ev_rounds, ev_fallbacks, ev_real;
    int *d_progress, *d_ticket;
    uint8_t *d_swapped;
    uint64_t *d_streams, *d_bits;
    int64_t *d_zatt, *d_ztotal, *d_state;
    uint32_t *d_words;
    nla_dev_constraint *d_con;
    int64_t zcap;
    double *h_F, *h_PEN, *h_GPEN, *h_X;
    int32_t *h_FEAS;
    uint8_t *h_swapped;
    int *h_progress;
    char err[200];
} isres_dev;

#define DFAIL(d, ...) do { snprintf((d)->err, sizeof (d)->err, __VA_ARGS__); return -1; } while (0)
#define DCK(d, call) do { int rc_ = (call); if (rc_) DFAIL(d, "%.90s failed: %.60s", #call, nla_dev_error_string(rc_)); } while (0)

static void dev_free_all(isres_dev *d)
{
    if (d->st) nla_stream_sync(d->st);
    if (d->rs) nla_stream_sync(d->rs);
    if (d->mts) { nla_mtstream_finish(d->mts, d->words_used); nla_mtstream_destroy(d->mts); }
    nla_dev_free(d->d_lb); nla_dev_free(d->d_ub); nla_dev_free(d->d_X); nla_dev_free(d->d_S); nla_dev_free(d->d_F);
    nla_dev_free(d->d_PEN); nla_dev_free(d->d_GPEN); nla_dev_free(d->d_scratch); nla_dev_free(d->d_z); nla_dev_free(d->d_FEAS);
    nla_dev_free(d->d_irank); nla_dev_free(d->d_counts); nla_dev_free(d->d_progress); nla_dev_free(d->d_ticket);
    nla_dev_free(d->d_swapped); nla_dev_free(d->d_streams); nla_dev_free(d->d_bits); nla_dev_free(d->d_zatt); nla_dev_free(d->d_gate);
    nla_dev_free(d->d_ztotal); nla_dev_free(d->d_state); nla_dev_free(d->d_words); nla_dev_free(d->d_con);
    nla_dev_free(d->d_inv); nla_dev_free(d->d_rho); nla_dev_free(d->d_ws); nla_dev_free(d->d_mu);
    nla_host_free(d->h_F); nla_host_free(d->h_PEN); nla_host_free(d->h_GPEN); nla_host_free(d->h_X); nla_host_free(d->h_FEAS);
    nla_host_free(d->h_swapped); nla_host_free(d->h_progress);
    nla_event_destroy(d->ev0); nla_event_destroy(d->ev1); nla_event_destroy(d->ev_gate);
    if (d->rs && d->rs != d->st) nla_stream_destroy(d->rs);
    if (d->st) nla_stream_destroy(d->st);
}

static int dev_alloc(isres_dev *d, const double *lb, const double *ub, const nla_dev_constraint *con)
{
    const size_t pop = (size_t) d->pop, ld = (size_t) d->ld, n = (size_t) d->n;
    int ok = 1;
    d->per = (d->pop + nlopt_amd_comm_world(d->comm) - 1) / nlopt_amd_comm_world(d->comm);
    d->popcap = d->per * nlopt_amd_comm_world(d->comm);
    d->units = (d->pop + 63) / 64;
    d->rowwords = (d->pop - 1 + 63) / 64;
    if (d->rowwords < 1) d->rowwords = 1;
    d->zcap = 4 * d->pop * (1 + 2 * (int64_t) d->n) + 4096;
    {
        const uint64_t init_w = 2ULL * (uint64_t) d->pop * (uint64_t) d->n, rank_w = 2ULL * (uint64_t) d->pop * (uint64_t) d->pop,
                       evo_w = 8ULL * (uint64_t) d->pop * (1 + 2 * (uint64_t) d->n);
        uint64_t need = init_w > rank_w ? init_w : rank_w;
        if (evo_w > need) need = evo_w;
        d->wchunk = 1ULL << 22;
        while (d->wchunk < need && d->wchunk < WORD_CHUNK_MAX) d->wchunk <<= 1;
    }
    d->st = nla_stream_create();
    { const char *e = NLA_DBG_ENV("NLA_ISRES_BITS_TWO_PASS"); d->bits_two_pass = e && atoi(e) > 0; }
    d->ev0 = nla_event_create(); d->ev1 = nla_event_create();      /* device time of the ranking kernel for the stats */
    if (!d->st || !d->ev0 || !d->ev1) return -1;
    /* OVERLAP (measured on MI355X, config 3: 72.9 -> 65.3 ms per generation; profiles/r03_isres_overlap_ab.txt): the generator — segment-state jumps, ranking bits, the evolve phase's deviates —
     * works on a stream of its own, beside the kernels of the generation that are bound by latency, not by throughput:
     *   rank counting (196 workgroups)            ||  the ranking's jumps + words -> bits
     *   the ranking pipeline (782 lone wavefronts) ||  the deviates of the evolve phase, generated AHEAD at the stream position the
     *                                                  ranking will leave if it takes all pop sweeps (it does unless a sweep swaps
     *                                                  nothing, isres.c:227; otherwise they are thrown away and generated again)
     *   the evolve rounds (serial look-up chains)  ||  the segment states the NEXT ranking's words start from
     * Every hand-over between the two streams is a host synchronisation of the producing stream (the driver synchronises at these
     * points anyway): nothing is ordered by events, and with overlap off rs IS st — the one-stream code of rounds 1-2. */
    d->rs = !d->overlap ? d->st : (nla_dbg_int("NLA_ISRES_RS_BACKGROUND", 0) > 0 ? nla_stream_create_background() : nla_stream_create());
    if (!d->rs) return -1;
    d->mts = nla_mtstream_create(d->rs);
    if (!d->mts) return -1;
    /* (room for 16 generations' stream from the start: 2 (pop - 1) pop words of ranking uniforms + about 4 pop (1 + 2 n) of deviates each) */
    if (nla_mtstream_expect(d->mts, 16ULL * (2ULL * (uint64_t) (pop > 1 ? pop - 1 : 1) * (uint64_t) pop + 4ULL * (uint64_t) pop * (uint64_t) (1 + 2 * n)))) return -1;
#define A(ptr, T, count) do { d->ptr = (T *) nla_dev_malloc(sizeof(T) * (size_t) (count)); if (!d->ptr) ok = 0; } while (0)
    A(d_lb, double, ld); A(d_ub, double, ld); A(d_X, double, pop * ld); A(d_S, double, pop * ld);
    A(d_F, double, d->popcap); A(d_PEN, double, d->popcap); A(d_GPEN, double, d->popcap); A(d_FEAS, int32_t, d->popcap);
    A(d_irank, int32_t, pop); A(d_scratch, double, 3 * ld);
    /* the stochastic ranking's pipeline state and uniform bits (pop^2 / 8 bytes each) only where a penalty can be nonzero: without
     * constraints every generation sorts by f (isres.c:203-204), and the population may then be as large as the rows fit */
    A(d_streams, uint64_t, (size_t) ((d->m + d->p > 0 ? d->units : 0) + 1) * pop); A(d_progress, int, d->units + 1); A(d_ticket, int, 1);
    A(d_swapped, uint8_t, pop); A(d_bits, uint64_t, d->m + d->p > 0 ? (size_t) d->popcap * (size_t) d->rowwords : 1);      /* (equal blocks of `per` rows for the all-gather) */
    A(d_words, uint32_t, d->wchunk); A(d_z, double, d->zcap); A(d_zatt, int64_t, d->zcap);
    A(d_counts, int32_t, d->wchunk / 4 / 1024 + 16); A(d_ztotal, int64_t, 1); A(d_state, int64_t, 16);
    A(d_con, nla_dev_constraint, d->m + d->p + 1);
    A(d_gate, int, d->units + 3);
    d->ev_gate = nla_event_create();
    if (!d->ev_gate) ok = 0;
    d->parallel_evolve = nla_isres_evolve2_supported(d->n) && !d->evolve_serial && !NLA_DBG_ENV("NLA_ISRES_EVOLVE_SERIAL");
    if (d->parallel_evolve) {
        A(d_inv, int32_t, pop); A(d_rho, double, 4); A(d_mu, double, d->survivors > 0 ? d->survivors : 1);
        d->d_ws = nla_dev_malloc(nla_isres_evolve2_ws_bytes(d->n));
        if (!d->d_ws || (d->d_rho && nla_memset(d->d_rho, 0, sizeof(double) * 4, d->st))) ok = 0;
    }
#undef A
    d->h_F = (double *) nla_host_malloc(sizeof(double) * pop);
    d->h_PEN = (double *) nla_host_malloc(sizeof(double) * pop);
    d->h_GPEN = (double *) nla_host_malloc(sizeof(double) * pop);
    d->h_FEAS = (int32_t *) nla_host_malloc(sizeof(int32_t) * pop);
    d->h_swapped = (uint8_t *) nla_host_malloc(pop);
    d->h_progress = (int *) nla_host_malloc(sizeof(int) * (size_t) (d->units + 1));
    d->h_X = (double *) nla_host_malloc(sizeof(double) * (d->dev_eval ? n : pop * ld));
    if (!ok || !d->h_F || !d->h_PEN || !d->h_GPEN || !d->h_FEAS || !d->h_swapped || !d->h_progress || !d->h_X) return -1;
    if (nla_memcpy_h2d(d->d_lb, lb, sizeof(double) * n, d->st) || nla_memcpy_h2d(d->d_ub, ub, sizeof(double) * n, d->st)) return -1;
    if (d->m + d->p > 0 && d->dev_eval && nla_memcpy_h2d(d->d_con, con, sizeof(nla_dev_constraint) * (size_t) (d->m + d->p), d->st)) return -1;
    d->h_progress[0] = (int) d->pop;
    for (int64_t u = 1; u <= d->units; ++u) d->h_progress[u] = 0;
    return nla_stream_sync(d->st) ? -1 : 0;
}

/* initial population (isres.c:122-128): 2 pop n stream words, in passes of d->wchunk */
static int dev_init_population(isres_dev *d, const double *x0)
{
    const uint64_t wpi = 2ULL * (uint64_t) d->n;
    int64_t per = (int64_t) (d->wchunk / wpi), k0;
    if (per < 1) DFAIL(d, "dimension too large for the word buffer");
    DCK(d, nla_memcpy_h2d(d->d_scratch, x0, sizeof(double) * (size_t) d->n, d->rs));
    for (k0 = 0; k0 < d->pop; k0 += per) {
        const int64_t cnt = d->pop - k0 < per ? d->pop - k0 : per;
        if (nla_mtstream_fill(d->mts, d->words_used + wpi * (uint64_t) k0, wpi * (uint64_t) cnt, d->d_words)) DFAIL(d, "MT stream fill failed");
        DCK(d, nla_k_isres_init(d->n, d->ld, d->d_lb, d->d_ub, d->d_words, k0, cnt, d->d_scratch, d->d_X, d->d_S, d->rs));
        DCK(d, nla_stream_sync(d->rs));
    }
    d->words_used += wpi * (uint64_t) d->pop;
    return 0;
}

/* selection (isres.c:202-229); *sweeps_out = ranking sweeps actually taken (0 on the sort path) */
static int dev_more_deviates(isres_dev *d, uint64_t phase_word0, int64_t *attempts_done, int64_t nattempts, int64_t *zcount);

/* can the ranking's bits be generated by the gated launch (in-order gates, the pipeline started beside it)? */
static int dev_rank_is_gated(const isres_dev *d)
{
    return d->gated && d->rs != d->st && nlopt_amd_comm_world(d->comm) == 1 && !d->bits_two_pass && d->d_gate != NULL;
}
/* ... that launch, on the generator's stream: from where the stream stands (words_used).  Called by dev_rank, or — so that the generator, which
 * bounds the ranking phase, does not wait for the host's pass over the generation's values (0.3 ms at pop = 5e4) — in front of that pass */
static int dev_rank_bits_gated(isres_dev *d)
{
    const int64_t pop = d->pop, popm1 = pop - 1;
    DCK(d, nla_memset(d->d_bits, 0, sizeof(uint64_t) * (size_t) pop * (size_t) d->rowwords, d->rs));
    DCK(d, nla_memset(d->d_gate, 0, sizeof(int) * (size_t) (d->units + 3), d->rs));
    DCK(d, nla_event_record(d->ev_gate, d->rs));
    d->bits_pre_g0 = nla_mtstream_origin(d->mts) + d->words_used;
    if (nla_mtstream_rankbits_gated(d->mts, d->words_used, d->words_used, 2ULL * (uint64_t) popm1 * (uint64_t) pop, popm1, d->rowwords, d->d_bits,
                                    d->d_gate, d->d_gate + d->units + 1, d->gen_waves_per_cu))
        DFAIL(d, "MT stream ranking bits failed");
    return 0;
}

static int dev_rank(isres_dev *d, int all_feasible, int64_t *sweeps_out, double *t_rng, nlopt_amd_stats *st)
{
    const int64_t pop = d->pop, popm1 = pop - 1;
    int64_t nsweeps = pop, rows_per, r0, i;
    double t0;
    /* GATED (round 5: in-order gates inside ONE generator launch; round 4 had four launches with a flag kernel behind each): the bits of
     * the pop sweeps are produced on the generator's stream by wavefronts that claim the stream's segments front first and count, per
     * block of 64 sweeps, how many of them are through with it (hip/mt_kernels.hip, nla_k_mt_rankbits_gated); the ranking pipeline is
     * launched at once on the main stream and its unit u — sweeps 64 u .. 64 u + 63 — waits until block u is complete
     * (hip/isres_stochrank.h).  The pipeline's unit u starts ~15 us after unit u - 1, which is time enough for the ten segments of a
     * block: the generation of the bits disappears behind the pipeline.  One rank with the generator on its own stream only (several
     * ranks all-gather the complete bits first). */
    int gated = dev_rank_is_gated(d);
    uint64_t gate_g0 = 0;
    int gate_err = 0;
    *sweeps_out = 0;
    DCK(d, nla_k_isres_rank_count(pop, d->d_F, d->d_PEN, d->d_streams, d->d_irank, d->st));
    if (all_feasible || popm1 <= 0) { d->bits_pre = 0; return 0; }      /* irank = stable sort by f (or the single individual) */
    /* the uniforms of all pop sweeps, reduced to bits, generated in whole-sweep passes */
    t0 = nla_seconds();
    rows_per = (int64_t) (d->wchunk / (2ULL * (uint64_t) popm1));
    if (rows_per < 1) DFAIL(d, "population too large for the word buffer");
    {
        /* several ranks: the sweeps' rows of bits are dealt in blocks over the ranks — each generates the stream words of ITS sweeps
         * (5e9 words per generation at pop = 5e4 in all: the costliest replicated phase) and reduces them to bits; the rows are then
         * all-gathered in place (pop^2 / 8 bytes: 313 MB at pop = 5e4).  The segment states behind the words (mt_jump_kernel) are
         * still built by every rank up to its last segment */
        const int world = nlopt_amd_comm_world(d->comm), rank = nlopt_amd_comm_rank(d->comm);
        const int64_t first = world > 1 ? (d->per * rank < pop ? d->per * rank : pop) : 0;
        const int64_t last = world > 1 ? (first + d->per < pop ? first + d->per : pop) : pop;
        if (gated) {
            if (!d->bits_pre && dev_rank_bits_gated(d)) return -1;
            gate_g0 = d->bits_pre_g0;
            d->bits_pre = 0;
        } else if (!d->bits_two_pass) {
            /* words -> bits in one kernel, all of this rank's sweeps in one launch (hip/mt_kernels.hip, mt_rankbits_kernel):
             * the words are never written to memory */
            if (last > first) {
                DCK(d, nla_memset(d->d_bits + (size_t) first * (size_t) d->rowwords, 0, sizeof(uint64_t) * (size_t) (last - first) * (size_t) d->rowwords, d->rs));
                if (nla_mtstream_rankbits(d->mts, d->words_used, d->words_used + 2ULL * (uint64_t) popm1 * (uint64_t) first,
                                          2ULL * (uint64_t) popm1 * (uint64_t) (last - first), popm1, d->rowwords, d->d_bits))
                    DFAIL(d, "MT stream ranking bits failed");
            }
        } else
        for (r0 = first; r0 < last; r0 += rows_per) {               /* A/B (NLA_ISRES_BITS_TWO_PASS=1): words into a buffer in passes, then isres_bits_kernel */
            const int64_t nr = last - r0 < rows_per ? last - r0 : rows_per;
            if (nla_mtstream_fill(d->mts, d->words_used + 2ULL * (uint64_t) popm1 * (uint64_t) r0, 2ULL * (uint64_t) popm1 * (uint64_t) nr, d->d_words))
                DFAIL(d, "MT stream fill failed");
            DCK(d, nla_k_isres_bits(d->d_words, r0, (int) nr, pop, d->d_bits, d->rs));
        }
        if (world > 1 && nla_comm_allgather_dev(d->comm, d->d_bits + (size_t) (d->per * rank) * (size_t) d->rowwords, d->d_bits,
                                                sizeof(uint64_t) * (size_t) d->per * (size_t) d->rowwords, d->rs))
            DFAIL(d, "all-gather of the ranking bits failed: %s", nlopt_amd_comm_error(d->comm));
    }
    if (!gated) {
        DCK(d, nla_stream_sync(d->rs));                /* the bits are there ... */
        if (d->rs != d->st) DCK(d, nla_stream_sync(d->st));      /* ... and so are the packed elements (rank counting ran beside them) */
    }
    *t_rng += nla_seconds() - t0;
    d->spec_valid = 0;
    for (;;) {
        DCK(d, nla_memcpy_h2d(d->d_progress, d->h_progress, sizeof(int) * (size_t) (d->units + 1), d->st));
        DCK(d, nla_memset(d->d_ticket, 0, sizeof(int), d->st));
        if (d->ev0) nla_event_record(d->ev0, d->st);
        if (gated && nsweeps == pop) DCK(d, nla_stream_wait_event(d->st, d->ev_gate));      /* the counters are zero before a unit looks at them */
        DCK(d, nla_k_isres_stochrank_gated(pop, nsweeps, d->d_streams, d->d_progress, d->d_bits, d->d_ticket, d->d_swapped, d->d_irank,
                                           gated && nsweeps == pop ? d->d_gate : NULL, gate_g0, pop, d->st));
        if (d->ev1) nla_event_record(d->ev1, d->st);
        DCK(d, nla_memcpy_d2h(d->h_swapped, d->d_swapped, (size_t) nsweeps, d->st));
        if (d->overlap && nsweeps == pop && !d->spec_valid) {
            /* while the pipeline runs: the evolve phase's deviates from where the stream will stand after pop sweeps */
            const int64_t expect = d->pop * (1 + 2 * (int64_t) d->n);
            const double t1 = nla_seconds();
            d->spec_word0 = d->words_used + 2ULL * (uint64_t) popm1 * (uint64_t) pop;
            d->spec_attempts = 0; d->spec_zcount = 0;
            DCK(d, nla_memset(d->d_ztotal, 0, sizeof(int64_t), d->rs));
            if (dev_more_deviates(d, d->spec_word0, &d->spec_attempts, (int64_t) (1.35 * (double) expect / 0.785) + 4096, &d->spec_zcount)) return -1;
            d->spec_valid = 1; ++d->spec_made;
            /* ... and behind them the segment states the NEXT ranking's words start from (that ranking begins behind these deviates).  Round 3 put
             * this beside the evolve rounds; but mt_jump_kernel's workgroups hold LDS on every compute unit while it runs, and the rounds' chain
             * kernel — ONE workgroup with 147 of a compute unit's 160 KB of LDS — then finds none to run on: one 45 us launch per generation
             * took 2.8-3.0 ms (every trace since round 4; understood in round 6, profiles/r06_isres_ahead.txt).  Here the main stream launches
             * nothing until the pipeline has ended; what is left of the jump then delays the first round by that much at most. */
            if (nla_dbg_int("NLA_ISRES_JUMP_IN_RANK", 1) && pop > 1 &&
                nla_mtstream_reserve(d->mts, d->spec_word0 + 4ULL * (uint64_t) d->spec_attempts + 2ULL * (uint64_t) popm1 * (uint64_t) pop))
                DFAIL(d, "MT stream reserve failed");
            *t_rng += nla_seconds() - t1;
        }
        if (gated && nsweeps == pop) DCK(d, nla_memcpy_d2h(&gate_err, d->d_gate + d->units + 2, sizeof gate_err, d->st));
        DCK(d, nla_stream_sync(d->st));
        if (gated && gate_err) {
            /* a unit of the pipeline gave up waiting for its block of bits (hip/isres_stochrank.h: 4 s): the generator's launch is late
             * beyond reason or failed.  Its stream's error, if any, comes out of the synchronisation; otherwise the bits are complete
             * after it and the ranking is done again on them, without gates */
            DCK(d, nla_stream_sync(d->rs));
            gated = 0; gate_err = 0;
            if (st) ++st->isres_gate_timeouts;
            continue;
        }
        if (st && d->ev0 && d->ev1) {
            const float ms = nla_event_elapsed_ms(d->ev0, d->ev1);
            if (ms >= 0) st->t_stochrank_ms += ms;
            ++st->stochrank_launches;
            st->stochrank_ticks += (uint64_t) pop + 2ULL * (uint64_t) nsweeps + (uint64_t) (nla_isres_stochrank_handoff() - 1) * (uint64_t) ((nsweeps + 63) / 64);     /* (hip/isres_stochrank.h) */
        }
        for (i = 0; i < nsweeps; ++i) if (!d->h_swapped[i]) break;      /* `if (!swapped) break;` isres.c:227 */
        if (i >= nsweeps - 1) break;               /* no early exit, or it was the last sweep anyway */
        nsweeps = i + 1;                           /* the reference stopped after sweep i: redo exactly that */
    }
    *sweeps_out = nsweeps;
    d->words_used += 2ULL * (uint64_t) popm1 * (uint64_t) nsweeps;
    if (d->spec_valid && d->spec_word0 != d->words_used) d->spec_valid = 0;     /* the ranking stopped early: those were the wrong words */
    return 0;
}

/* ---- a generation with NaN objective / penalty values: the selection on the HOST, literally as the reference does it ---------------
 * The device ranking works on dense integer ranks of the values (a total order); a NaN has none.  The reference's comparisons with
 * a NaN are all false: its sort comparator (isres.c:48-54) then calls the NaN equal to everything — not an ordering, so the result
 * is whatever glibc's qsort_r (which nlopt_qsort_r calls on Linux, util/qsort_r.c:164-170) makes of those answers — and its
 * stochastic-ranking sweeps (isres.c:207-228) never move an element past a NaN.  Both are reproduced here by running exactly that
 * code: the same qsort_r of the same libc with the same comparator, the same sweeps over uniforms drawn sequentially from the run's
 * stream.  Slow (pop^2 steps on one core) and rare; everything else of the generation stays on the device. */
static int nan_key_compare(const void *a_, const void *b_, void *keys_)                 /* isres.c:48-54 */
{
    const double *keys = (const double *) keys_;
    const int32_t *a = (const int32_t *) a_, *b = (const int32_t *) b_;
    return keys[*a] < keys[*b] ? -1 : (keys[*a] > keys[*b] ? +1 : 0);
}
static int host_rank_with_nan(isres_dev *d, int all_feasible, int64_t *sweeps_out)
{
    const int64_t pop = d->pop;
    const double *fval = d->h_F, *penalty = d->h_PEN;
    int32_t *irank = (int32_t *) malloc(sizeof(int32_t) * (size_t) pop);
    int64_t i, j, sweeps = 0;
    if (!irank) DFAIL(d, "out of memory (host ranking)");
    for (i = 0; i < pop; ++i) irank[i] = (int32_t) i;
    if (all_feasible) qsort_r(irank, (size_t) pop, sizeof(int32_t), nan_key_compare, (void *) fval);
    else {
        uint32_t mt[NLA_MT_N];
        int pos;
        nla_mtstream_host_state(d->mts, d->words_used, mt, &pos);
        for (i = 0; i < pop; ++i) {                                                      /* isres.c:207-228 */
            int swapped = 0;
            for (j = 0; j < pop - 1; ++j) {
                uint32_t w[2];
                double u;
                for (int q = 0; q < 2; ++q) {
                    if (pos >= NLA_MT_N) { nla_mt_regen(mt); pos = 0; }
                    w[q] = nla_mt_temper(mt[pos++]);
                }
                u = 0. + (1. - 0.) * (((w[0] >> 5) * 67108864.0 + (w[1] >> 6)) * (1.0 / 9007199254740992.0));
                if (u < 0.45 || (penalty[irank[j]] == 0 && penalty[irank[j + 1]] == 0)) {
                    if (fval[irank[j]] > fval[irank[j + 1]]) { const int32_t t = irank[j]; irank[j] = irank[j + 1]; irank[j + 1] = t; swapped = 1; }
                } else if (penalty[irank[j]] > penalty[irank[j + 1]]) { const int32_t t = irank[j]; irank[j] = irank[j + 1]; irank[j + 1] = t; swapped = 1; }
            }
            ++sweeps;
            if (!swapped) break;
        }
        d->words_used += 2ULL * (uint64_t) (pop - 1) * (uint64_t) sweeps;
    }
    if (nla_memcpy_h2d(d->d_irank, irank, sizeof(int32_t) * (size_t) pop, d->st) || nla_stream_sync(d->st)) { free(irank); DFAIL(d, "upload of the ranking failed"); }
    free(irank);
    *sweeps_out = sweeps;
    return 0;
}

/* append the accepted deviates of `nattempts` further attempts of the evolve phase */
static int dev_more_deviates(isres_dev *d, uint64_t phase_word0, int64_t *attempts_done, int64_t nattempts, int64_t *zcount)
{
    while (nattempts > 0) {
        int64_t a = nattempts < (int64_t) (d->wchunk / 4) ? nattempts : (int64_t) (d->wchunk / 4), zt;
        if (*zcount + a > d->zcap) a = d->zcap - *zcount;
        if (a <= 0) DFAIL(d, "deviate buffer exhausted");
        if (nla_mtstream_fill(d->mts, phase_word0 + 4ULL * (uint64_t) *attempts_done, 4ULL * (uint64_t) a, d->d_words)) DFAIL(d, "MT stream fill failed");
        DCK(d, nla_k_isres_nrand(d->d_words, a, *attempts_done, d->d_counts, d->d_ztotal, *zcount, d->d_z, d->d_zatt, d->rs));
        DCK(d, nla_memcpy_d2h(&zt, d->d_ztotal, sizeof zt, d->rs));
        DCK(d, nla_stream_sync(d->rs));
        *zcount = zt;
        *attempts_done += a;
        nattempts -= a;
    }
    return 0;
}

/* mutation + variation (isres.c:234-280) */
static int dev_evolve(isres_dev *d, double taup, double tau, double *t_rng)
{
    const uint64_t word0 = d->words_used;
    const int64_t expect = d->pop * (1 + 2 * (int64_t) d->n);
    int64_t attempts_done = 0, zcount = 0, state[16] = { 0 }, last_att;
    double t0 = nla_seconds();
    int phase, reserved = 0;
    if (d->spec_valid && d->spec_word0 == word0) {             /* generated beside the ranking pipeline (overlap) */
        attempts_done = d->spec_attempts; zcount = d->spec_zcount; ++d->spec_used;
    } else {
        DCK(d, nla_memset(d->d_ztotal, 0, sizeof(int64_t), d->rs));
        if (dev_more_deviates(d, word0, &attempts_done, (int64_t) (1.35 * (double) expect / 0.785) + 4096, &zcount)) return -1;
    }
    d->spec_valid = 0;
    *t_rng += nla_seconds() - t0;
    if (d->parallel_evolve) {
        DCK(d, nla_k_isres_inverse(d->pop, d->d_irank, d->d_inv, d->st));
        /* what the mutation phase's rounds predict the children's starts with (the parents' rows do not change during that phase) */
        DCK(d, nla_k_isres_evolve_parent_mu(d->n, d->ld, d->survivors, d->d_lb, d->d_ub, d->d_irank, d->d_X, d->d_S, d->d_mu, d->st));
    }
    for (phase = 0; phase < 2; ++phase) {
        const int64_t kend = phase == 0 ? d->pop : d->survivors;
        state[0] = phase == 0 ? d->survivors : 0;
        state[2] = 0; state[9] = 0; state[10] = 0; state[11] = 0; state[14] = 0;
        DCK(d, nla_memcpy_h2d(d->d_state, state, sizeof state, d->st));
        if (d->parallel_evolve && phase == 1)                  /* memcpy(x0, xs, n) before the variation loop (isres.c:253) */
            DCK(d, nla_memcpy_d2d(d->d_scratch, d->d_X, sizeof(double) * (size_t) d->n, d->st));
        DCK(d, nla_stream_sync(d->st));
        for (;;) {
            if (d->parallel_evolve) {
                /* a round resolves up to a block of individuals (fewer when the predicted windows are left or a variation individual depends on an
                 * earlier one of its block: 680 of a mutation block's 1024, 188 of a variation block's 256 per round at config 3): enqueue what the phase should need, then look.  Every
                 * look is a stream synchronisation, a copy of the state and a cold start of the next batch — some 50 us, and round 4's
                 * batches of at most 24 rounds made ~28 of them per generation; a round enqueued in vain costs its launches (~10 us) */
                const int64_t left = kend - state[0];
                int rounds = (int) (left / (phase == 0 ? 680 : 170)) + 2;
                if (rounds > 72) rounds = 72;
                DCK(d, nla_k_isres_evolve_rounds(d->n, d->ld, phase, d->pop, d->survivors, zcount, taup, tau, d->d_lb, d->d_ub, d->d_z,
                                                 d->d_irank, d->d_inv, d->d_X, d->d_S, d->d_scratch, d->d_state, d->d_rho, d->d_ws, d->d_mu, rounds, d->st));
                d->ev_rounds += (uint64_t) rounds;
                if (d->overlap && !reserved) {
                    /* beside the rounds: the segment states behind the NEXT ranking's words (those of this phase's deviates generated
                     * so far + pop sweeps: a little more than it will need) */
                    reserved = 1;
                    if (d->pop > 1 && nla_mtstream_reserve(d->mts, word0 + 4ULL * (uint64_t) attempts_done + 2ULL * (uint64_t) (d->pop - 1) * (uint64_t) d->pop))
                        DFAIL(d, "MT stream reserve failed");
                }
            } else
                DCK(d, nla_k_isres_evolve(d->n, d->ld, phase, d->pop, d->survivors, zcount, taup, tau, d->d_lb, d->d_ub, d->d_z, d->d_irank,
                                          d->d_X, d->d_S, d->d_scratch, d->d_state, d->st));
            DCK(d, nla_memcpy_d2h(state, d->d_state, sizeof state, d->st));
            DCK(d, nla_stream_sync(d->st));
            if (state[2]) {                                    /* the deviates generated so far ran out */
                t0 = nla_seconds();
                if (dev_more_deviates(d, word0, &attempts_done, (int64_t) (0.5 * (double) expect / 0.785) + 4096, &zcount)) return -1;
                *t_rng += nla_seconds() - t0;
                state[2] = 0;
                DCK(d, nla_memcpy_h2d(d->d_state, state, sizeof state, d->st));
                DCK(d, nla_stream_sync(d->st));
                continue;
            }
            if (d->parallel_evolve && state[10]) {             /* one individual the look-up could not resolve: the serial kernel takes it */
                state[10] = 0; state[14] = state[0] + 1;
                DCK(d, nla_memcpy_h2d(d->d_state, state, sizeof state, d->st));
                DCK(d, nla_k_isres_evolve(d->n, d->ld, phase, d->pop, d->survivors, zcount, taup, tau, d->d_lb, d->d_ub, d->d_z, d->d_irank,
                                          d->d_X, d->d_S, d->d_scratch, d->d_state, d->st));
                DCK(d, nla_memcpy_d2h(state, d->d_state, sizeof state, d->st));
                DCK(d, nla_stream_sync(d->st));
                ++d->ev_fallbacks;
                if (!state[2]) { state[14] = 0; DCK(d, nla_memcpy_h2d(d->d_state, state, sizeof state, d->st)); DCK(d, nla_stream_sync(d->st)); }
                else { state[14] = 0; state[10] = 1; continue; }     /* ran out inside the serial step: refill above, then retry it */
            }
            if (!d->parallel_evolve || state[0] >= kend) break;
        }
        d->ev_real += (uint64_t) state[11];                    /* rounds of this phase that resolved something */
#ifdef NLA_EV2_REASONS
        fprintf(stderr, "evolve phase %d: rounds %lld, stopped by window %lld, by end/dependency %lld, by a candidate that left its deviates %lld, full blocks %lld, resolved %lld\n", phase, (long long) state[11], (long long) state[3], (long long) state[4], (long long) state[5], (long long) state[6], (long long) state[7]);
        fprintf(stderr, "   rounds whose mutated-coordinate sum differs from the direct sum: %lld, total difference %lld\n", (long long) state[13], (long long) state[15]);
        state[3] = state[4] = state[5] = state[6] = state[7] = state[13] = state[15] = 0;
#endif
    }
    if (NLA_DBG_ENV("NLA_ISRES_DEBUG")) fprintf(stderr, "evolve2: rounds enqueued %llu, serial fallbacks %llu; overlap %d: deviates generated ahead %llu times, used %llu times\n", (unsigned long long) d->ev_rounds, (unsigned long long) d->ev_fallbacks, d->overlap, (unsigned long long) d->spec_made, (unsigned long long) d->spec_used);
    if (NLA_DBG_ENV("NLA_ISRES_DEBUG")) fprintf(stderr, "evolve: fixpoint rounds %lld for %lld individuals, deviates %lld; cycles stage %lld eval %lld scan %lld fin %lld all %lld\n", (long long) state[3], (long long) d->pop, (long long) state[1], (long long) state[4], (long long) state[5], (long long) state[6], (long long) state[7], (long long) state[8]);
    if (state[1] <= 0) DFAIL(d, "evolve consumed no deviates");
    DCK(d, nla_memcpy_d2h(&last_att, d->d_zatt + (state[1] - 1), sizeof last_att, d->st));
    DCK(d, nla_stream_sync(d->st));
    d->words_used += 4ULL * (uint64_t) (last_att + 1);
    return 0;
}

static int con_eval_host(const nla_constraint *c, unsigned n, const double *x, double *res)
{
    if (c->f) res[0] = c->f(n, x, NULL, c->f_data);
    else c->mf(c->m, res, n, x, NULL, c->f_data);
    return 0;
}

/* a constraint the evaluation kernel can compute: the registered block-sum constraint with valid data */
static int device_constraint(const nla_constraint *cc)
{
    const unsigned *qQ = (const unsigned *) cc->f_data;
    return cc->f && cc->m == 1 && nlopt_amd_constraint_id(cc->f) == NLA_CON_BLOCKSUM && qQ && qQ[1] != 0 && qQ[0] < qQ[1];
}
/* will a run with these constraints evaluate everything on the device (given a device objective)?  The dispatcher asks before it
 * decides who negates a maximised objective (api_optimize.c). */
int nla_isres_constraints_on_device(unsigned m, const nla_constraint *fc, unsigned p, const nla_constraint *h)
{
    unsigned c;
    for (c = 0; c < m; ++c) if (!device_constraint(fc + c)) return 0;
    for (c = 0; c < p; ++c) if (!device_constraint(h + c)) return 0;
    return 1;
}

nlopt_result nla_isres_minimize(nlopt_opt opt, int n, nlopt_func f, void *f_data, int m, nla_constraint *fc, int p, nla_constraint *h,
                                const double *lb, const double *ub, double *x, double *minf, nla_stopping *stop, int population)
{
    const double PHI = 1.0, SURVIVOR = 1.0 / 7.0;                 /* isres.c:71-73 */
    isres_dev D;
    nlopt_result ret = NLOPT_SUCCESS;
    nla_dev_constraint *con = NULL;
    nlopt_amd_stats *st = opt ? &opt->stats : NULL;
    double minf_penalty = HUGE_VAL, minf_gpenalty = HUGE_VAL, taup, tau, *results = NULL;
    unsigned maxdim = 1;
    int j, c, dev_eval, agreed_force = 0;
    const int need_x = stop->xtol_rel > 0 || stop->xtol_abs != NULL;
    int64_t k;

    *minf = HUGE_VAL;
    if (!population) population = 20 * (n + 1);                                        /* :88 */
    if (population < 1) { nla_stop_msg(stop, "population %d is too small", population); return NLOPT_INVALID_ARGS; }
    taup = PHI / sqrt(2 * n);
    tau = PHI / sqrt(2 * sqrt(n));
    for (j = 0; j < n; ++j)
        if (nla_isinf(lb[j]) || nla_isinf(ub[j])) { nla_stop_msg(stop, "isres requires a finite search region"); return NLOPT_INVALID_ARGS; }
    if (nla_dev_count() <= 0) {
        nla_stop_msg(stop, "nlopt_amd: no HIP device visible (this library has no CPU fallback)");
        nla_comm_agree_ready(opt ? opt->comm : NULL, 0);
        return NLOPT_FAILURE;
    }
    /* the stochastic ranking packs (individual, rank of f, rank of penalty) into 64 bits, 20 bits each (hip/isres_kernels.hip), and its
     * uniforms' bits take pop^2 / 8 bytes (137 GB at 2^20); without constraints no generation ranks stochastically (all penalties
     * are 0: isres.c:203-204 sorts by f) and the population is only bounded by what the rows need in memory */
    if (population > (1 << 20) && m + p > 0) {
        nla_stop_msg(stop, "nlopt_amd: ISRES with nonlinear constraints supports populations up to 2^20 (the stochastic ranking's state grows as population^2)");
        return NLOPT_INVALID_ARGS;
    }

    /* can everything be evaluated on the device? */
    memset(&D, 0, sizeof D);
    nla_evaluator_resolve(&D.ev, opt, f, f_data);
    D.obj = D.ev.kind == NLA_EVAL_DEVICE ? D.ev.obj : -1;
    dev_eval = D.ev.kind != NLA_EVAL_HOST && !(opt && nlopt_get_param(opt, "amd_host_eval", 0) != 0);
    con = (nla_dev_constraint *) calloc((size_t) (m + p + 1), sizeof *con);
    if (!con) { nla_stop_msg(stop, "nlopt_amd: out of memory"); nla_comm_agree_ready(opt ? opt->comm : NULL, 0); return NLOPT_OUT_OF_MEMORY; }
    for (c = 0; c < m + p; ++c) {
        const nla_constraint *cc = c < m ? fc + c : h + (c - m);
        if (cc->m > maxdim) maxdim = cc->m;
        if (device_constraint(cc)) {
            const unsigned *qQ = (const unsigned *) cc->f_data;
            con[c].type = NLA_CON_BLOCKSUM; con[c].q = qQ[0]; con[c].Q = qQ[1]; con[c].tol = cc->tol[0];
        } else dev_eval = 0;
    }
    results = (double *) malloc(sizeof(double) * maxdim);
    if (!results) { nla_stop_msg(stop, "nlopt_amd: out of memory"); free(con); nla_comm_agree_ready(opt ? opt->comm : NULL, 0); return NLOPT_OUT_OF_MEMORY; }

    D.n = n; D.ld = (n + 1) & ~1; D.m = m; D.p = p; D.pop = population; D.dev_eval = dev_eval;
    D.comm = opt ? opt->comm : NULL;
    D.evolve_serial = opt ? nlopt_get_param(opt, "amd_isres_evolve_serial", 0) != 0 : 0;
    D.gated = 1;                     /* (A/B switches in round 5, profiles/r05_isres_steps.txt: gated 49.5 ms per generation, not gated 55.3) */
    D.gen_waves_per_cu = 8;
    D.overlap = opt ? nlopt_get_param(opt, "amd_isres_overlap", 1) != 0 : 1;            /* 0: the one-stream generation */
    D.overlap = nla_dbg_int("NLA_ISRES_OVERLAP", D.overlap) > 0;     /* A/B switch for the bench */
    D.survivors = (int64_t) ceil(population * SURVIVOR);                               /* :93 */
    {   /* several ranks: all of them go on, or none (a rank that could not set up would leave the others in the first all-gather) */
        const int mine = dev_alloc(&D, lb, ub, con) == 0;
        const int all = nlopt_amd_comm_world(D.comm) > 1
            ? nla_comm_agree_same(D.comm, mine, nla_problem_fingerprint(NLOPT_GN_ISRES, n, population, D.obj + 100 * (m + p), lb, ub, x, stop) + nla_params_fingerprint(opt)) : mine;
        if (all <= 0 || !mine) {
            if (!mine) nla_stop_msg(stop, "nlopt_amd: could not create the ISRES device state (out of device memory?)");
            else if (all < 0) nla_stop_msg(stop, NLA_MSG_RANKS_DIFFER);
            else nla_stop_msg(stop, "nlopt_amd: another rank could not set up its ISRES device state");
            dev_free_all(&D); free(con); free(results);
            return !mine ? NLOPT_OUT_OF_MEMORY : (all < 0 ? NLOPT_INVALID_ARGS : NLOPT_FAILURE);
        }
    }
#define DEVFAIL() do { nla_stop_msg(stop, "device engine: %s", D.err); ret = NLOPT_FAILURE; goto done; } while (0)
    if (dev_init_population(&D, x)) DEVFAIL();

    for (;;) {                                       /* each loop body = one generation (isres.c:130) */
        int all_feasible = 1;
        int64_t kbest = -1, sweeps = 0;
        double t0, t_rng = 0;
        nla_stopping agreed_view;
        const nla_stopping *sp;                      /* what the clock / force_stop tests of this generation look at (all ranks the same) */
        if (opt && opt->progress) opt->progress(opt->progress_data, st ? (long) st->generations : 0, (long) *stop->nevals_p);
        t0 = nla_seconds();
        if (dev_eval) {
            /* this rank's block of candidates, then the all-gather (in place: block r sits at r * per) */
            const int64_t first = D.per * nlopt_amd_comm_rank(D.comm);
            const int64_t mine = first >= D.pop ? 0 : (D.pop - first < D.per ? D.pop - first : D.per);
            if (nla_k_isres_eval((D.obj >= 0 && D.ev.sign < 0) ? (D.obj | NLA_OBJ_NEGATE) : D.obj, n, D.ld, D.d_X + (size_t) first * (size_t) D.ld, mine, m, p, D.d_con, D.d_F + first, D.d_PEN + first,
                                 D.d_GPEN + first, D.d_FEAS + first, D.st) ||
                (D.ev.kind == NLA_EVAL_USER && nla_userobj_eval_rows(D.ev.user, n, D.ld, mine, D.d_X + (size_t) first * (size_t) D.ld, D.d_F + first,
                                                                      NULL, D.ev.sign, D.st)) ||
                nla_comm_allgather_dev(D.comm, D.d_F + first, D.d_F, sizeof(double) * (size_t) D.per, D.st) ||
                nla_comm_allgather_dev(D.comm, D.d_PEN + first, D.d_PEN, sizeof(double) * (size_t) D.per, D.st) ||
                nla_comm_allgather_dev(D.comm, D.d_GPEN + first, D.d_GPEN, sizeof(double) * (size_t) D.per, D.st) ||
                nla_comm_allgather_dev(D.comm, D.d_FEAS + first, D.d_FEAS, sizeof(int32_t) * (size_t) D.per, D.st) ||
                nla_memcpy_d2h(D.h_F, D.d_F, sizeof(double) * (size_t) D.pop, D.st) ||
                nla_memcpy_d2h(D.h_PEN, D.d_PEN, sizeof(double) * (size_t) D.pop, D.st) ||
                nla_memcpy_d2h(D.h_GPEN, D.d_GPEN, sizeof(double) * (size_t) D.pop, D.st) ||
                nla_memcpy_d2h(D.h_FEAS, D.d_FEAS, sizeof(int32_t) * (size_t) D.pop, D.st) || nla_stream_sync(D.st)) {
                snprintf(D.err, sizeof D.err, "evaluation pass failed");
                DEVFAIL();
            }
        } else if (nla_memcpy_d2h(D.h_X, D.d_X, sizeof(double) * (size_t) D.pop * (size_t) D.ld, D.st) || nla_stream_sync(D.st)) {
            snprintf(D.err, sizeof D.err, "population read-back failed");
            DEVFAIL();
        }
        if (st) st->t_eval_s += nla_seconds() - t0;

        /* the generator of the ranking's bits goes out in front of the bookkeeping below when the generation will be ranked on the device with
         * bits (some penalty positive, no NaN: the same two tests the ranking itself makes further down) */
        D.bits_pre = 0;
        if (dev_eval && D.pop > 1 && D.m + D.p > 0 && dev_rank_is_gated(&D) && !NLA_DBG_ENV("NLA_ISRES_NO_BITS_PRE")) {
            int anypen = 0, nan = 0;
            for (k = 0; k < D.pop; ++k) { anypen |= D.h_PEN[k] > 0; nan |= (D.h_F[k] != D.h_F[k]) || (D.h_PEN[k] != D.h_PEN[k]); }
            if (anypen && !nan) { if (dev_rank_bits_gated(&D)) DEVFAIL(); D.bits_pre = 1; }
        }
        /* the reference's per-candidate bookkeeping, in candidate order (isres.c:134-199) */
        /* several ranks: the clock and the force_stop flag are decided once per generation, by all ranks together (comm.c) */
        sp = nla_comm_agree_stop(D.comm, stop, &agreed_view, &agreed_force);
        if (!sp) { snprintf(D.err, sizeof D.err, "stop agreement failed: %s", nlopt_amd_comm_error(D.comm)); DEVFAIL(); }
        for (k = 0; k < D.pop; ++k) {
            int feasible = 1;
            double gpenalty, fk, pk;
            const double *xk = NULL;
            ++*stop->nevals_p;
            if (dev_eval) { fk = D.h_F[k]; pk = D.h_PEN[k]; gpenalty = D.h_GPEN[k]; feasible = D.h_FEAS[k]; }
            else {
                unsigned ires;
                xk = D.h_X + (size_t) k * (size_t) D.ld;
                fk = f((unsigned) n, xk, NULL, f_data);
                if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
                pk = 0;
                for (c = 0; c < m; ++c) {
                    con_eval_host(fc + c, (unsigned) n, xk, results);
                    if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
                    for (ires = 0; ires < fc[c].m; ++ires) {
                        double gval = results[ires];
                        if (gval > fc[c].tol[ires]) feasible = 0;
                        if (gval < 0) gval = 0;
                        pk += gval * gval;
                    }
                }
                gpenalty = pk;
                for (c = 0; c < p; ++c) {
                    con_eval_host(h + c, (unsigned) n, xk, results);
                    if (nla_stop_forced(stop)) { ret = NLOPT_FORCED_STOP; goto done; }
                    for (ires = 0; ires < h[c].m; ++ires) {
                        double hval = results[ires];
                        if (fabs(hval) > h[c].tol[ires]) feasible = 0;
                        pk += hval * hval;
                    }
                }
                D.h_F[k] = fk; D.h_PEN[k] = pk;
            }
            if (st) ++st->evals_trial;
            if (opt && opt->trace) {
                if (opt->trace_len < opt->trace_cap) {
                    nlopt_amd_trace_rec *r = opt->trace + opt->trace_len;
                    r->f = fk; r->row = k; r->kind = 3; r->accepted = feasible;
                }
                ++opt->trace_len;
            }
            if (pk > 0) all_feasible = 0;
            if ((pk <= minf_penalty || feasible) && (fk <= *minf || minf_gpenalty > 0)
                && ((feasible ? 0 : pk) != minf_penalty || fk != *minf)) {                       /* :174-177 */
                if (dev_eval && need_x) {          /* nlopt_stop_x / later comparisons need this candidate: fetch it (rare) */
                    if (nla_memcpy_d2h(D.h_X, D.d_X + (size_t) k * (size_t) D.ld, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) {
                        snprintf(D.err, sizeof D.err, "candidate read-back failed");
                        DEVFAIL();
                    }
                    xk = D.h_X;
                }
                if (fk < stop->minf_max && feasible) ret = NLOPT_MINF_MAX_REACHED;
                else if (!nla_isinf(*minf)) {
                    if (nla_stop_f(stop, fk, *minf) && nla_stop_f(stop, feasible ? 0 : pk, minf_penalty)) ret = NLOPT_FTOL_REACHED;
                    else if (need_x && nla_stop_x(stop, xk, x)) ret = NLOPT_XTOL_REACHED;
                }
                if (xk) { memcpy(x, xk, sizeof(double) * (size_t) n); kbest = -1; }   /* memcpy(x, xs+k*n), isres.c:188 */
                else kbest = k;                                                     /* ... deferred to one read-back per generation */
                *minf = fk;
                minf_penalty = feasible ? 0 : pk;
                minf_gpenalty = feasible ? 0 : gpenalty;
                if (ret != NLOPT_SUCCESS) break;
            }
            if (nla_stop_forced(sp)) ret = NLOPT_FORCED_STOP;
            else if (nla_stop_evals(stop)) ret = NLOPT_MAXEVAL_REACHED;
            else if (nla_stop_time(sp)) ret = NLOPT_MAXTIME_REACHED;
            if (ret != NLOPT_SUCCESS) break;
        }
        if (kbest >= 0) {                          /* the last accepted best of this generation -> x */
            if (nla_memcpy_d2h(x, D.d_X + (size_t) kbest * (size_t) D.ld, sizeof(double) * (size_t) n, D.st) || nla_stream_sync(D.st)) {
                snprintf(D.err, sizeof D.err, "best-point read-back failed");
                DEVFAIL();
            }
        }
        if (ret != NLOPT_SUCCESS) goto done;
        if (!dev_eval) {                           /* the ranking kernels read f / penalty from the device */
            if (nla_memcpy_h2d(D.d_F, D.h_F, sizeof(double) * (size_t) D.pop, D.st) ||
                nla_memcpy_h2d(D.d_PEN, D.h_PEN, sizeof(double) * (size_t) D.pop, D.st)) { snprintf(D.err, sizeof D.err, "upload failed"); DEVFAIL(); }
        }

        t0 = nla_seconds();
        {
            int has_nan = 0;
            if (!NLA_DBG_ENV("NLA_ISRES_NO_NAN_HOST"))           /* (A/B switch for the tests: without the host selection a NaN generation differs) */
            for (k = 0; k < D.pop && !has_nan; ++k) has_nan = (D.h_F[k] != D.h_F[k]) || (D.h_PEN[k] != D.h_PEN[k]);
            if (has_nan ? host_rank_with_nan(&D, all_feasible, &sweeps) : dev_rank(&D, all_feasible, &sweeps, &t_rng, st)) DEVFAIL();
        }
        if (st) { st->t_rank_s += nla_seconds() - t0; st->rank_sweeps += (uint64_t) sweeps; }
        t0 = nla_seconds();
        if (dev_evolve(&D, taup, tau, &t_rng)) DEVFAIL();
        if (st) { st->t_evolve_s += nla_seconds() - t0; st->t_rng_s += t_rng; ++st->generations; st->mt_words = D.words_used; st->evolve_rounds_enqueued = D.ev_rounds; st->evolve_rounds = D.ev_real; }
    }
done:
    if (st) st->mt_words = D.words_used;
    dev_free_all(&D);
    free(con);
    free(results);
    return ret;
}
