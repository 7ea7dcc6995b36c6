/* mtstream.c — the calling thread's MT19937 generator, continued on the device.
 *
 * The reference draws all randomness from one thread-local generator in a fixed order
 * (src/util/mt19937ar.c:76-77).  A run on the device must consume
 * *that* stream at *those* offsets.  nla_mtstream_create() snapshots the host generator; global
 * word index g then means word g%624 of the (g/624)-th regeneration of the snapshot, and the run's
 * first word is g = origin.  Segment p (NLA_MT_SEG_REGENS regenerations) starts from a block array
 * obtained by GF(2) jump-ahead; arrays for segments 0..P-1 are built in ceil(log2 P) doubling
 * rounds (array[c+i] = jump_{c segments}(array[i])).  nla_mtstream_finish() puts the host
 * generator where the reference's would be after the run, so later draws continue the stream.
 * All device work is enqueued on the stream given at creation.
 */
#include "nla_internal.h"
#include <stdlib.h>
#include <string.h>

struct nla_mtstream {
    void *stream;
    int seg_regens;             /* regenerations per segment: NLA_MT_SEG_REGENS, or the caller's shorter power of two (nla_mtstream_create_seg) */
    uint64_t seg_words;         /* 624 * seg_regens */
    uint32_t base[NLA_MT_N];
    int base_consumed;
    uint32_t *d_states;         /* nstates x 624 */
    int nstates, cap_states, round_base;   /* round_base: power of two, states [0,round_base) are the round's sources */
    uint64_t *d_polys[NLA_MT_MAXPOW2];
    int err;
};

static int log2_int(unsigned v) { int k = 0; while ((1u << k) < v) ++k; return k; }

/* A stream whose segments are seg_regens regenerations long (a power of two, 1 .. NLA_MT_SEG_REGENS): the same words at the same
 * offsets — only how many wavefronts produce them changes (and how many jump-ahead states are built: one per segment).  Only _fill
 * serves such a stream (the fused ranking-bits kernel is ISRES's, which wants the long segments). */
nla_mtstream *nla_mtstream_create_seg(void *stream, int seg_regens)
{
    nla_mtstream *s;
    if (seg_regens < 1 || seg_regens > NLA_MT_SEG_REGENS || (seg_regens & (seg_regens - 1))) return NULL;
    s = (nla_mtstream *) calloc(1, sizeof *s);
    if (!s) return NULL;
    s->stream = stream;
    s->seg_regens = seg_regens;
    s->seg_words = (uint64_t) NLA_MT_N * (uint64_t) seg_regens;
    nla_mt_export(s->base, &s->base_consumed);
    s->cap_states = 64;
    s->d_states = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * NLA_MT_N * (size_t) s->cap_states);
    if (!s->d_states) { free(s); return NULL; }
    if (nla_memcpy_h2d(s->d_states, s->base, sizeof s->base, stream) || nla_stream_sync(stream)) {
        nla_dev_free(s->d_states); free(s); return NULL;
    }
    s->nstates = 1;
    s->round_base = 1;
    return s;
}

nla_mtstream *nla_mtstream_create(void *stream) { return nla_mtstream_create_seg(stream, NLA_MT_SEG_REGENS); }

int nla_mtstream_expect(nla_mtstream *s, uint64_t words)
{
    uint64_t want = words / s->seg_words + 2;
    int ncap = s->cap_states;
    uint32_t *nd;
    if (want > (1u << 17)) want = 1u << 17;
    if (want <= (uint64_t) s->cap_states) return 0;
    while ((uint64_t) ncap < want) ncap *= 2;
    nd = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * NLA_MT_N * (size_t) ncap);
    if (!nd) return -1;
    if (nla_memcpy_d2d(nd, s->d_states, sizeof(uint32_t) * NLA_MT_N * (size_t) s->nstates, s->stream) || nla_stream_sync(s->stream)) { nla_dev_free(nd); return -1; }
    nla_dev_free(s->d_states);
    s->d_states = nd;
    s->cap_states = ncap;
    return 0;
}

void nla_mtstream_destroy(nla_mtstream *s)
{
    if (!s) return;
    nla_stream_sync(s->stream);
    for (int k = 0; k < NLA_MT_MAXPOW2; ++k) nla_dev_free(s->d_polys[k]);
    nla_dev_free(s->d_states);
    free(s);
}

uint64_t nla_mtstream_origin(const nla_mtstream *s) { return (uint64_t) s->base_consumed; }

static const uint64_t *dev_poly(nla_mtstream *s, int k)
{
    if (k < 0 || k >= NLA_MT_MAXPOW2) return NULL;
    if (!s->d_polys[k]) {
        const uint64_t *h = nla_mt_jump_poly_pow2(k);
        uint64_t *d = (uint64_t *) nla_dev_malloc(sizeof(uint64_t) * NLA_MT_POLYWORDS);
        if (!h || !d) return NULL;
        /* h is a process-lifetime table: safe source for an async copy */
        if (nla_memcpy_h2d(d, h, sizeof(uint64_t) * NLA_MT_POLYWORDS, s->stream)) { nla_dev_free(d); return NULL; }
        s->d_polys[k] = d;
    }
    return s->d_polys[k];
}

static int ensure_states(nla_mtstream *s, uint64_t seg_needed)
{
    const int seg_log = log2_int((unsigned) s->seg_regens);
    if (seg_needed >= (1ULL << 30)) return -1;
    while ((uint64_t) s->nstates <= seg_needed) {
        int done, cnt, want;
        const uint64_t *poly;
        if (s->nstates == 2 * s->round_base) s->round_base = s->nstates;   /* round complete */
        done = s->nstates - s->round_base;          /* states of this round already built */
        want = (int) (seg_needed + 1 - (uint64_t) s->nstates);
        cnt = s->round_base - done;
        if (cnt > want) cnt = want;
        if (s->nstates + cnt > s->cap_states) {
            int ncap = s->cap_states;
            uint32_t *nd;
            while (ncap < s->nstates + cnt) ncap *= 2;
            nd = (uint32_t *) nla_dev_malloc(sizeof(uint32_t) * NLA_MT_N * (size_t) ncap);
            if (!nd) return -1;
            if (nla_memcpy_d2d(nd, s->d_states, sizeof(uint32_t) * NLA_MT_N * (size_t) s->nstates, s->stream) ||
                nla_stream_sync(s->stream)) { nla_dev_free(nd); return -1; }
            nla_dev_free(s->d_states);
            s->d_states = nd;
            s->cap_states = ncap;
        }
        /* array[round_base + i] = array[i] jumped by round_base segments */
        poly = dev_poly(s, seg_log + log2_int((unsigned) s->round_base));
        if (!poly) return -1;
        if (nla_k_mt_jump(poly, s->d_states + (size_t) done * NLA_MT_N,
                          s->d_states + (size_t) s->nstates * NLA_MT_N, cnt, s->stream)) return -1;
        s->nstates += cnt;
    }
    return 0;
}

int nla_mtstream_fill(nla_mtstream *s, uint64_t rel_first, uint64_t count, uint32_t *d_out)
{
    uint64_t g_first, seg0, seg1;
    if (!count) return 0;
    g_first = (uint64_t) s->base_consumed + rel_first;
    seg0 = g_first / s->seg_words;
    seg1 = (g_first + count - 1) / s->seg_words;
    if (ensure_states(s, seg1)) return -1;
    if (s->seg_regens != NLA_MT_SEG_REGENS)
        return nla_k_mt_generate_seg(s->d_states + (size_t) seg0 * NLA_MT_N, seg0, (int) (seg1 - seg0 + 1), g_first, count, d_out, s->seg_regens, s->stream);
    return nla_k_mt_generate(s->d_states + (size_t) seg0 * NLA_MT_N, seg0, (int) (seg1 - seg0 + 1),
                             g_first, count, d_out, s->stream);
}

/* build the segment states up to stream word rel_last (relative to the run's first word) now — asynchronously on the generator's
 * stream — so that a later fill / rankbits up to there finds them ready (ISRES overlap mode: beside the evolve rounds) */
int nla_mtstream_reserve(nla_mtstream *s, uint64_t rel_last)
{
    return ensure_states(s, ((uint64_t) s->base_consumed + rel_last) / s->seg_words);
}

/* the ranking bits of stream words [rel_first, rel_first + count) (relative to the run's first word, as nla_mtstream_fill),
 * rel_rank0 = the ranking's first word: see nla_k_mt_rankbits */
int nla_mtstream_rankbits_gated(nla_mtstream *s, uint64_t rel_rank0, uint64_t rel_first, uint64_t count, int64_t popm1, int64_t rowwords,
                                uint64_t *d_bits, int *d_gate, int *d_ticket, int waves_per_cu)
{
    uint64_t g_first, seg0, seg1;
    if (!count) return 0;
    if (s->seg_regens != NLA_MT_SEG_REGENS) return -1;          /* (the fused kernel knows the default segment length only) */
    g_first = (uint64_t) s->base_consumed + rel_first;
    seg0 = g_first / NLA_MT_SEG_WORDS;
    seg1 = (g_first + count - 1) / NLA_MT_SEG_WORDS;
    if (ensure_states(s, seg1)) return -1;
    return nla_k_mt_rankbits_gated(s->d_states + (size_t) seg0 * NLA_MT_N, seg0, (int) (seg1 - seg0 + 1), (uint64_t) s->base_consumed + rel_rank0,
                                   g_first, count, popm1, rowwords, d_bits, d_gate, d_ticket, waves_per_cu, s->stream);
}

int nla_mtstream_rankbits(nla_mtstream *s, uint64_t rel_rank0, uint64_t rel_first, uint64_t count, int64_t popm1, int64_t rowwords, uint64_t *d_bits)
{
    uint64_t g_first, seg0, seg1;
    if (!count) return 0;
    if (s->seg_regens != NLA_MT_SEG_REGENS) return -1;          /* (the fused kernel knows the default segment length only) */
    g_first = (uint64_t) s->base_consumed + rel_first;
    seg0 = g_first / NLA_MT_SEG_WORDS;
    seg1 = (g_first + count - 1) / NLA_MT_SEG_WORDS;
    if (ensure_states(s, seg1)) return -1;
    return nla_k_mt_rankbits(s->d_states + (size_t) seg0 * NLA_MT_N, seg0, (int) (seg1 - seg0 + 1), (uint64_t) s->base_consumed + rel_rank0,
                             g_first, count, popm1, rowwords, d_bits, s->stream);
}

/* the generator as the host would hold it `rel_word` words into the run: block array + index of the next word in it (for the rare
 * host fall-backs that must draw sequentially from the run's stream, e.g. ISRES's ranking of a generation with NaN values) */
void nla_mtstream_host_state(nla_mtstream *s, uint64_t rel_word, uint32_t mt[NLA_MT_N], int *pos)
{
    const uint64_t g = (uint64_t) s->base_consumed + rel_word;
    nla_mt_advance_blocks_host(s->base, g / NLA_MT_N, mt);
    *pos = (int) (g % NLA_MT_N);
}

int nla_mtstream_finish(nla_mtstream *s, uint64_t consumed)
{
    const uint64_t g = (uint64_t) s->base_consumed + consumed;
    const uint64_t blk = g / NLA_MT_N;
    const uint64_t seg = blk / (uint64_t) s->seg_regens;
    uint32_t mt[NLA_MT_N];
    uint64_t r;
    if (seg < (uint64_t) s->nstates) {
        if (nla_memcpy_d2h(mt, s->d_states + (size_t) seg * NLA_MT_N, sizeof mt, s->stream) ||
            nla_stream_sync(s->stream)) return -1;
        for (r = blk % (uint64_t) s->seg_regens; r > 0; --r) nla_mt_regen(mt);
    } else {
        nla_mt_advance_blocks_host(s->base, blk, mt);
    }
    nla_mt_import(mt, (int) (g % NLA_MT_N));
    return 0;
}
