/* port_kernels.c — CPU ORACLE (test infrastructure): plain-C statements of what each HIP kernel
 * must compute, used (a) by the -m gpu parity tests as the per-kernel checker and (b) by
 * port_emu_engine.c to exercise the product's host-side speculate/commit logic on machines
 * without a GPU.  Each follows the reference lines it cites; none is linked into the product. */
#include "port_oracle.h"
#include "objfuncs.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* tempered stream words of a generator state, in order (mt19937ar.c:97-131) */
void orc_k_words(uint64_t count, uint32_t *out)
{
    for (uint64_t i = 0; i < count; ++i) out[i] = orc_genrand_int32();
}

static double res53w(uint32_t w0, uint32_t w1)                 /* mt19937ar.c:194-198 */
{
    uint32_t a = w0 >> 5, b = w1 >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* crs.c:216-218: row r coordinate j = lb[j] + (ub[j]-lb[j]) * res53(words of (r,j)) */
void orc_k_init_rows(int n, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t nrows, double *X)
{
    for (int64_t r = 0; r < nrows; ++r)
        for (int j = 0; j < n; ++j) {
            const uint32_t *w = words + ((size_t) r * (size_t) n + (size_t) j) * 2;
            X[(size_t) r * (size_t) ld + j] = lb[j] + (ub[j] - lb[j]) * res53w(w[0], w[1]);
        }
}

void orc_k_eval(int obj, int n, int ld, const double *P, int64_t count, double *F)
{
    const double sign = (obj >= 0 && (obj & 0x100)) ? -1. : 1.;          /* NLA_OBJ_NEGATE (include/nlopt_amd.h): the launcher's caller minimises -f */
    if (obj >= 0) obj &= 0xff;
    for (int64_t c = 0; c < count; ++c) F[c] = sign * nla_obj_eval_seq(obj, (unsigned) n, P + (size_t) c * (size_t) ld, NULL);
}

/* crs.c:72,89-109 on one 2n-word block, positions in reduced space (best row removed) */
void orc_k_vitter(int n, int64_t N, const uint32_t *words, int nblocks, int32_t *jn, int32_t *pos, int32_t *last)
{
    for (int b = 0; b < nblocks; ++b) {
        const uint32_t *w = words + (size_t) b * 2 * (size_t) n;
        int32_t *out = pos + (size_t) b * (size_t) n;
        int Nleft = (int) (N - 1), nleft = n, Nfree = Nleft - nleft, i = 0, t = 0, wi = 1;
        jn[b] = (int32_t) (w[0] % (uint32_t) n);
        while (nleft > 1) {
            double q = ((double) Nfree) / Nleft;
            double v = res53w(w[wi], w[wi + 1]);
            wi += 2;
            while (q > v) { ++i; --Nfree; --Nleft; q = (q * Nfree) / Nleft; }
            out[t++] = i;
            ++i; --Nleft; --nleft;
        }
        out[n - 1] = i;
        last[b] = (int32_t) (w[2 * n - 1] % (uint32_t) Nleft);
    }
}

/* actual row of pick t of a block, given the best row i0 (the `i += i == i0` skipping, crs.c:92,97,106,109) */
static int64_t pick_row(int n, const int32_t *pos, int32_t last, int64_t i0, int t)
{
    if (t < n - 1) { int64_t r = pos[t]; return r + (r >= i0); }
    {
        int64_t rb = pos[n - 1], a = rb + (rb >= i0) + last;
        a += (a == i0);
        return a;
    }
}

/* crs.c:69,101-120 for K slots */
void orc_k_gather(int n, int ld, const double *X, int64_t i0, const int32_t *jn, const int32_t *pos, const int32_t *last,
                  int K, const double *lb, const double *ub, double *TX)
{
    for (int s = 0; s < K; ++s) {
        double *x = TX + (size_t) s * (size_t) ld;
        const int32_t *p = pos + (size_t) s * (size_t) n;
        memcpy(x, X + (size_t) i0 * (size_t) ld, sizeof(double) * (size_t) n);
        for (int t = 0; t < n; ++t) {
            const double *xi = X + (size_t) pick_row(n, p, last[s], i0, t) * (size_t) ld;
            if (t == jn[s]) for (int k = 0; k < n; ++k) x[k] -= xi[k] * (0.5 * n);
            else            for (int k = 0; k < n; ++k) x[k] += xi[k];
        }
        for (int k = 0; k < n; ++k) {
            x[k] *= 2.0 / n;
            if (x[k] > ub[k]) x[k] = ub[k];
            else if (x[k] < lb[k]) x[k] = lb[k];
        }
    }
}

/* crs.c:140-145: out = clamp(best(1+w) - w p), w from the block's n urands */
void orc_k_mutate(int n, const double *best, const double *p, const uint32_t *words, const double *lb, const double *ub,
                  double *out)
{
    for (int i = 0; i < n; ++i) {
        double w = 0. + (1. - 0.) * res53w(words[2 * i], words[2 * i + 1]);
        double v = best[i] * (1 + w) - w * p[i];
        if (v > ub[i]) v = ub[i];
        else if (v < lb[i]) v = lb[i];
        out[i] = v;
    }
}

/* resumable gather-sum for one slot (crs.c:69,101-120 in pieces): picks [t0, return value) are
 * added to acc (n doubles; initialised from the best row when t0 == 0), stopping before the first
 * pick >= t0 whose row is among W[0..nun); when the sum completes (returns n) acc is scaled by
 * 2/n and clamped, i.e. becomes the trial point x. */
int orc_k_advance_slot_cols(int n, int ncol, int ld, const double *X, int64_t i0, int32_t jn, const int32_t *pos, int32_t last,
                            const int64_t *W, int nun, int t0, const double *lb, const double *ub, double *acc)
{
    /* n picks are summed; ncol coordinates of every row are held here (ncol == n: whole rows; a column slice otherwise) */
    int e = n, t;
    for (t = t0; t < n && e == n; ++t) {
        const int64_t r = pick_row(n, pos, last, i0, t);
        for (int j = 0; j < nun; ++j) if (W[j] == r) { e = t; break; }
    }
    if (e == t0) return e;
    if (t0 == 0) memcpy(acc, X + (size_t) i0 * (size_t) ld, sizeof(double) * (size_t) ncol);
    for (t = t0; t < e; ++t) {
        const double *xi = X + (size_t) pick_row(n, pos, last, i0, t) * (size_t) ld;
        if (t == jn) for (int k = 0; k < ncol; ++k) acc[k] -= xi[k] * (0.5 * n);
        else         for (int k = 0; k < ncol; ++k) acc[k] += xi[k];
    }
    if (e == n)
        for (int k = 0; k < ncol; ++k) {
            acc[k] *= 2.0 / n;
            if (acc[k] > ub[k]) acc[k] = ub[k];
            else if (acc[k] < lb[k]) acc[k] = lb[k];
        }
    return e;
}
int orc_k_advance_slot(int n, int ld, const double *X, int64_t i0, int32_t jn, const int32_t *pos, int32_t last,
                       const int64_t *W, int nun, int t0, const double *lb, const double *ub, double *acc)
{
    return orc_k_advance_slot_cols(n, n, ld, X, i0, jn, pos, last, W, nun, t0, lb, ub, acc);
}


/* nla_k_crs_chain (hip/crs_chain.hip) stated sequentially: the window's slots front to back; each slot's gather-sum reads a pick
 * of row W[j], j < a, as the chain resolved so far says the row stands (written by an earlier block of the window -> that block's
 * trial point / mutation, else the row itself), then the slot is evaluated (trial, mutation with the next block's words) and the
 * chain advanced over every block that can now be decided — crs_trial's decisions (crs.c:125-156) on the list of worst rows W
 * with their values Wf, a new value that lands among them being tracked.  The device does the same with the slots in flight
 * concurrently; a slot there waits until the fate of the row it needs is known, so its reads are these.  Records as the kernel
 * writes them: j | producer slot << 8 | kind << 16. */
typedef struct { double fT, fM; int32_t t, pad; } orc_slot_status;
/* the same on a COLUMN SLICE (hip/crs_chain.hip, SH instance): X holds columns [c0, c0 + ncols) of every row (stride
 * ld), TX / TM hold WHOLE points (stride ldf); the slot's slice is formed here, `exchange` (NULL: single process) hands it to the other
 * ranks and returns when theirs are in this rank's TX; evaluation, mutation (around the whole best row `xbest`, clamped by the whole
 * bounds lbf / ubf) and the chain are the single-process statements on the whole point.  lb / ub: the slice's bounds. */
typedef int (*orc_chain_exchange_fn)(void *ctx, int a, int q);
int orc_k_crs_chain_cols(int obj, int n, int ncols, int c0, int ld, int ldf, const double *X, int64_t i0, double f_best, const double *xbest,
                         const int32_t *jn_ring, const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                         uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int slot_mask, const double *lb, const double *ub,
                         const double *lbf, const double *ubf, double *TX, double *TM, orc_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec,
                         int fwcap, const orc_slot_status *decide_with, uint32_t *dbg, orc_chain_exchange_fn exchange, void *ctx);
void orc_k_crs_chain(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring, const int32_t *pos_ring,
                     const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                     const int64_t *W, const double *Wf, int nW, int slot_mask, const double *lb, const double *ub, double *TX, double *TM,
                     orc_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, const orc_slot_status *decide_with, uint32_t *dbg)
{
    (void) orc_k_crs_chain_cols(obj, n, n, 0, ld, ld, X, i0, f_best, X + (size_t) i0 * (size_t) ld, jn_ring, pos_ring, last_ring, words_ring, ring_blocks,
                                first_block, K, W, Wf, nW, slot_mask, lb, ub, lb, ub, TX, TM, status, fwcnt, fwrec, fwcap, decide_with, dbg, NULL, NULL);
}
int orc_k_crs_chain_cols(int obj, int n, int ncols, int c0, int ld, int ldf, const double *X, int64_t i0, double f_best, const double *xbest,
                         const int32_t *jn_ring, const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                         uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int slot_mask, const double *lb, const double *ub,
                         const double *lbf, const double *ubf, double *TX, double *TM, orc_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec,
                         int fwcap, const orc_slot_status *decide_with, uint32_t *dbg, orc_chain_exchange_fn exchange, void *ctx)
{
    /* decide_with != NULL: the chain's decisions are taken on THESE f values (a device run's, which differ from this file's in the
     * last bits: enough to flip a comparison between two nearly equal values) — everything else is computed here as always */
    enum { XCAP = 32 };
    uint32_t next = 0, wp = 0, nextra = 0, *rowstate = (uint32_t *) calloc((size_t) (nW > 0 ? nW : 1), sizeof(uint32_t));
    int halt = 0;
    double xf[XCAP]; int64_t xrow[XCAP];
    for (int a = 0; a < K; ++a) {
        const uint64_t block = first_block + (uint64_t) a;
        const uint32_t rb = (uint32_t) (block % ring_blocks);
        const int q = (int) (block & (uint64_t) slot_mask), nun = a < nW ? a : nW;
        const int32_t *pos = pos_ring + (size_t) rb * (size_t) n;
        double *whole = TX + (size_t) q * (size_t) ldf, *acc = whole + c0, *m = TM + (size_t) q * (size_t) ldf;
        uint32_t nrec = 0;
        memcpy(acc, X + (size_t) i0 * (size_t) ld, sizeof(double) * (size_t) ncols);
        for (int t = 0; t < n; ++t) {
            const int64_t r = pick_row(n, pos, last_ring[rb], i0, t);
            const double *xi = X + (size_t) r * (size_t) ld;
            for (int j = 0; j < nun; ++j)
                if (W[j] == r && r != i0) {
                    const uint32_t rs = rowstate[j];
                    uint32_t pj = 0, kind = 0;
                    if (rs && (int) (rs >> 3) < a) {
                        pj = rs >> 3; kind = (rs >> 1) & 3u;
                        xi = (kind == 1 ? TX : TM) + (size_t) ((first_block + pj) & (uint64_t) slot_mask) * (size_t) ldf + c0;
                    }
                    if ((int) nrec < fwcap) fwrec[(size_t) a * (size_t) fwcap + nrec] = (uint32_t) j | (pj << 8) | (kind << 16);
                    ++nrec;
                    break;
                }
            if (t == jn_ring[rb]) for (int k = 0; k < ncols; ++k) acc[k] -= xi[k] * (0.5 * n);
            else                  for (int k = 0; k < ncols; ++k) acc[k] += xi[k];
        }
        for (int k = 0; k < ncols; ++k) {
            acc[k] *= 2.0 / n;
            if (acc[k] > ub[k]) acc[k] = ub[k];
            else if (acc[k] < lb[k]) acc[k] = lb[k];
        }
        fwcnt[a] = nrec;
        if (exchange && exchange(ctx, a, q)) { free(rowstate); return -1; }
        orc_k_eval(obj, n, ldf, whole, 1, &status[a].fT);
        orc_k_mutate(n, xbest, whole, words_ring + (size_t) ((block + 1) % ring_blocks) * 2 * (size_t) n, lbf, ubf, m);
        orc_k_eval(obj, n, ldf, m, 1, &status[a].fM);
        status[a].t = n; status[a].pad = 0;
        /* the chain: every block up to this one can be decided now */
        while (!halt && next < (uint32_t) K && next <= (uint32_t) a) {
            const uint32_t j = next;
            double fw = -HUGE_VAL, fnew = 0;
            int64_t rw = -1;
            int xi = -1, kind = 0;
            if (wp < (uint32_t) nW) { fw = Wf[wp]; rw = W[wp]; }
            for (uint32_t e = 0; e < nextra; ++e)
                if (rw < 0 || xf[e] > fw || (xf[e] == fw && xrow[e] > rw)) { fw = xf[e]; rw = xrow[e]; xi = (int) e; }
            if (rw < 0) { halt = 1; break; }
            const orc_slot_status *dj = decide_with ? decide_with + j : status + j;
            if (dj->fT < fw) { kind = 1; fnew = dj->fT; }
            else if (dj->fM < fw) { kind = 2; fnew = dj->fM; }
            if (kind) {
                if (xi >= 0) { xf[xi] = xf[nextra - 1]; xrow[xi] = xrow[nextra - 1]; --nextra; }
                else { rowstate[wp] = 1u | ((uint32_t) kind << 1) | (j << 3); ++wp; }
                if (nW > 0 && (fnew > Wf[nW - 1] || (fnew == Wf[nW - 1] && rw > W[nW - 1]))) {
                    if (nextra == XCAP) halt = 1;
                    else { xf[nextra] = fnew; xrow[nextra] = rw; ++nextra; }
                }
                if (fnew < f_best || (fnew == f_best && rw < i0)) halt = 1;
            }
            next = j + (kind == 1 ? 1u : 2u);
        }
    }
    if (dbg) {                                  /* final chain state for comparisons: next, halt, wp, nextra, then rowstate[nW] */
        dbg[0] = next; dbg[1] = (uint32_t) halt; dbg[2] = wp; dbg[3] = nextra;
        for (int j = 0; j < nW; ++j) dbg[8 + j] = rowstate[j];
    }
    free(rowstate);
    return 0;
}
