/* emu_device.c — TEST INFRASTRUCTURE: a CPU stand-in for the product's *device layer*, so that the product's host drivers
 * (isres_driver.c, mlsl_driver.c, lbfgs_driver.c, mma_driver.c, mtstream.c, comm.c, the API shell) can be run without a GPU —
 * in particular by the world_size-2 gloo tests of the sharded ISRES / MLSL paths (tests/test_multiproc.py).
 *
 * It implements the C-ABI of include/nlopt_amd.h that hip/devrt.hip and the kernel launchers export: "device" memory is host
 * memory, streams and events are tokens, every launcher does its kernel's job with plain loops in the REFERENCE's operation
 * order (objectives through objfuncs.h's sequential formulas, local searches through the oracle ports) — so a driver run over
 * this layer must reproduce the oracle evaluation by evaluation, bit for bit.  The multi-start evolve of ISRES is reported as
 * unsupported so the driver takes its serial-kernel path.  (CRS2_LM's driver additionally has its own engine-level emulation,
 * port_emu_engine.c.)
 *
 * Linked with the product's C sources into oracle/libnlopt_amd_emu.so (make emudev).  The product library never sees this file;
 * only tests load the emulated library (by path).
 */
#include "../nlopt_amd/csrc/nla_internal.h"
#include "../nlopt_amd/csrc/objfuncs.h"
#include "port_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#define EMU_ERR 999

/* ---- runtime (hip/devrt.hip) --------------------------------------------------------------------------------------------- */
int nla_dev_count(void) { return 1; }
int nla_dev_set(int dev) { return dev == 0 ? 0 : EMU_ERR; }
/* fault injection for the drivers' error paths: orc_emu_fail_alloc_at(k) makes the k-th allocation from now on (device or pinned
 * host memory) fail once; 0 = never.  orc_emu_allocs() = allocations since that call. */
static long emu_alloc_count = 0, emu_alloc_fail_at = 0, emu_live = 0;
void orc_emu_fail_alloc_at(long k) { emu_alloc_count = 0; emu_alloc_fail_at = k; }
long orc_emu_allocs(void) { return emu_alloc_count; }
long orc_emu_live(void) { return emu_live; }         /* device / pinned buffers, streams and events not yet released */
static void *emu_alloc(size_t bytes)
{
    void *p;
    if (++emu_alloc_count == emu_alloc_fail_at) return NULL;
    p = malloc(bytes ? bytes : 1);
    if (p) ++emu_live;
    return p;
}
static void emu_release(void *p) { if (p) { --emu_live; free(p); } }
/* the same for kernel launches: the k-th nla_k_* call from now on returns an error instead of doing its job */
static long emu_launch_count = 0, emu_launch_fail_at = 0;
void orc_emu_fail_launch_at(long k) { emu_launch_count = 0; emu_launch_fail_at = k; }
long orc_emu_launches(void) { return emu_launch_count; }
static int emu_launch_fails(void) { return ++emu_launch_count == emu_launch_fail_at; }
#define EMU_LAUNCH() do { if (emu_launch_fails()) return EMU_ERR; } while (0)
void *nla_dev_malloc(size_t bytes) { return emu_alloc(bytes); }
void nla_dev_free(void *p) { emu_release(p); }
/* "uncached device memory" is what the workgroups of one launch — and, for a column-sharded CRS2_LM job, the launches of different
 * RANKS — hand data to each other through: here an anonymous shared-memory file (memfd) mapped shared, so that another process can map
 * the same pages (nla_ipc_export / nla_ipc_open below: the peer opens /proc/<pid>/fd/<fd>; nothing is left behind when a process dies) */
#define EMU_UC_MAX 64
static struct { void *p; size_t bytes; int fd; } emu_uc[EMU_UC_MAX];
void *nla_dev_malloc_uncached(size_t bytes)
{
    int k, fd;
    void *p;
    if (++emu_alloc_count == emu_alloc_fail_at) return NULL;
    if (!bytes) bytes = 1;
    for (k = 0; k < EMU_UC_MAX && emu_uc[k].p; ++k) { }
    if (k == EMU_UC_MAX) return NULL;
    fd = (int) syscall(SYS_memfd_create, "nla_emu_uc", 0u);
    if (fd < 0) return NULL;
    if (ftruncate(fd, (off_t) bytes)) { close(fd); return NULL; }
    p = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) { close(fd); return NULL; }
    emu_uc[k].p = p; emu_uc[k].bytes = bytes; emu_uc[k].fd = fd;
    ++emu_live;
    return p;
}
void nla_dev_free_uncached(void *p)
{
    if (!p) return;
    for (int k = 0; k < EMU_UC_MAX; ++k)
        if (emu_uc[k].p == p) { munmap(p, emu_uc[k].bytes); close(emu_uc[k].fd); emu_uc[k].p = NULL; --emu_live; return; }
}
void nla_debug_uncached_stats(long out[4]) { out[0] = out[1] = out[2] = out[3] = 0; }
void *nla_host_malloc(size_t bytes) { return emu_alloc(bytes); }
void nla_host_free(void *p) { emu_release(p); }
int nla_host_register(void *p, size_t bytes) { (void) p; (void) bytes; return 0; }      /* "device" memory is host memory here */
void nla_host_unregister(void *p) { (void) p; }
int nla_memcpy_h2d(void *dst, const void *src, size_t bytes, void *st) { (void) st; if (bytes) memmove(dst, src, bytes); return 0; }
int nla_memcpy_d2h(void *dst, const void *src, size_t bytes, void *st) { (void) st; if (bytes) memmove(dst, src, bytes); return 0; }
int nla_memcpy_d2d(void *dst, const void *src, size_t bytes, void *st) { (void) st; if (bytes) memmove(dst, src, bytes); return 0; }
int nla_memset(void *dst, int value, size_t bytes, void *st) { (void) st; if (bytes) memset(dst, value, bytes); return 0; }
void *nla_stream_create(void) { return emu_alloc(1); }
void *nla_stream_create_background(void) { return emu_alloc(1); }
void *nla_stream_create_cu_share(int part, int parts) { (void) part; (void) parts; return emu_alloc(1); }
/* peer-mapped memory between emulated devices: the blob names the exporting process and its memfd */
typedef struct { uint64_t magic; int64_t pid; int32_t fd; int32_t pad; uint64_t bytes; } emu_ipc_blob;
static struct { void *p; size_t bytes; } emu_ipc_open[EMU_UC_MAX];
int nla_ipc_export(const void *p, void *blob96)
{
    emu_ipc_blob b;
    if (getenv("NLA_EMU_NO_IPC")) return EMU_ERR;                 /* (tests: the set-up falls back to the conservative passes, on every rank) */
    for (int k = 0; k < EMU_UC_MAX; ++k)
        if (emu_uc[k].p == p) {
            memset(&b, 0, sizeof b);
            b.magic = 0x656d75697063ull; b.pid = (int64_t) getpid(); b.fd = emu_uc[k].fd; b.bytes = emu_uc[k].bytes;
            memset(blob96, 0, NLA_IPC_BYTES);
            memcpy(blob96, &b, sizeof b);
            return 0;
        }
    return EMU_ERR;
}
void *nla_ipc_open(const void *blob96)
{
    emu_ipc_blob b;
    char path[64];
    int fd, k;
    void *p;
    memcpy(&b, blob96, sizeof b);
    if (b.magic != 0x656d75697063ull) return NULL;
    for (k = 0; k < EMU_UC_MAX && emu_ipc_open[k].p; ++k) { }
    if (k == EMU_UC_MAX) return NULL;
    snprintf(path, sizeof path, "/proc/%lld/fd/%d", (long long) b.pid, (int) b.fd);
    fd = open(path, O_RDWR);
    if (fd < 0) return NULL;
    p = mmap(NULL, (size_t) b.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return NULL;
    emu_ipc_open[k].p = p; emu_ipc_open[k].bytes = (size_t) b.bytes;
    ++emu_live;
    return p;
}
void nla_ipc_close(void *p)
{
    if (!p) return;
    for (int k = 0; k < EMU_UC_MAX; ++k)
        if (emu_ipc_open[k].p == p) { munmap(p, emu_ipc_open[k].bytes); emu_ipc_open[k].p = NULL; --emu_live; return; }
}
/* the column-sharded window (hip/crs_chain.hip, SH instance), one emulated device per PROCESS: this rank's launcher walks the window's
 * slots front to back (port_kernels.c, orc_k_crs_chain_cols), stores its columns of a slot into every rank's TX, raises its chunk flags
 * everywhere and waits — really waits: the other ranks are other processes doing the same — until the slot's flags of all ranks show this
 * launch, then evaluates and advances the chain as the single-process launcher does.  Same flag / stop-word protocol as the kernel. */
#define EMU_SH_MAXW 8
typedef struct { double fT, fM; int32_t t, pad; } orc_slot_status;
typedef struct { int world, rank, c0, ldf, chunks_total, chunk0; double *peerTX[EMU_SH_MAXW]; uint32_t *peerflags[EMU_SH_MAXW], *peerstop[EMU_SH_MAXW];
                 const double *xbest, *lbf, *ubf; } emu_shard;
int nla_crs_chain_sh_chunks(int n, int ncols) { (void) n; return (ncols + 63) / 64; }
size_t nla_crs_chain_sh_table_bytes(void) { return sizeof(emu_shard); }
size_t nla_crs_chain_sh_stop_bytes(void) { return sizeof(uint32_t) * 2 * EMU_SH_MAXW; }
int nla_crs_chain_sh_table(void *host_image, int world, int rank, int c0, int ldf, int chunks_total, int chunk0, void *const *peerTX,
                           void *const *peerflags, void *const *peerstop, const double *xbest, const double *lbf, const double *ubf)
{
    emu_shard t;
    if (world < 2 || world > EMU_SH_MAXW || rank < 0 || rank >= world) return EMU_ERR;
    memset(&t, 0, sizeof t);
    t.world = world; t.rank = rank; t.c0 = c0; t.ldf = ldf; t.chunks_total = chunks_total; t.chunk0 = chunk0;
    for (int r = 0; r < world; ++r) { t.peerTX[r] = (double *) peerTX[r]; t.peerflags[r] = (uint32_t *) peerflags[r]; t.peerstop[r] = (uint32_t *) peerstop[r]; }
    t.xbest = xbest; t.lbf = lbf; t.ubf = ubf;
    memcpy(host_image, &t, sizeof t);
    return 0;
}
uint32_t nla_crs_chain_sh_tickets(int n, int ncols, int K, int grid_cap) { (void) grid_cap; return (uint32_t) nla_crs_chain_sh_chunks(n, ncols) * (uint32_t) K + 1u; }
typedef struct { const emu_shard *S; double *TX; int ncols, mychunks; uint32_t seq; int failed; } emu_sh_ctx;
static double emu_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec; }
static int emu_sh_wait(const uint32_t *word, uint32_t want, int exact)
{
    const double t0 = emu_now();
    for (unsigned spin = 0;; ++spin) {
        const uint32_t v = __atomic_load_n(word, __ATOMIC_ACQUIRE);
        if (exact ? (v >> 2) == (want >> 2) : (int32_t) (v - want) >= 0) return 0;
        if ((spin & 63u) == 63u) { sched_yield(); if (emu_now() - t0 > 20.) return -1; }
    }
}
static int emu_sh_exchange(void *vctx, int a, int q)
{
    emu_sh_ctx *c = (emu_sh_ctx *) vctx;
    const emu_shard *S = c->S;
    const double *mine = c->TX + (size_t) q * (size_t) S->ldf + S->c0;
    for (int r = 0; r < S->world; ++r)
        if (r != S->rank) memcpy(S->peerTX[r] + (size_t) q * (size_t) S->ldf + S->c0, mine, sizeof(double) * (size_t) c->ncols);
    for (int r = 0; r < S->world; ++r)
        for (int ch = 0; ch < c->mychunks; ++ch)
            __atomic_store_n(S->peerflags[r] + (size_t) a * (size_t) S->chunks_total + (size_t) (S->chunk0 + ch), c->seq << 2, __ATOMIC_RELEASE);
    for (int ch = 0; ch < S->chunks_total; ++ch)
        if (emu_sh_wait(S->peerflags[S->rank] + (size_t) a * (size_t) S->chunks_total + (size_t) ch, c->seq << 2, 0)) { c->failed = 1; return -1; }
    return 0;
}
int orc_k_crs_chain_cols(int obj, int n, int ncols, int c0, int ld, int ldf, const double *X, int64_t i0, double f_best, const double *xbest,
                         const int32_t *jn_ring, const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                         uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int slot_mask, const double *lb, const double *ub,
                         const double *lbf, const double *ubf, double *TX, double *TM, orc_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec,
                         int fwcap, const orc_slot_status *decide_with, uint32_t *dbg, int (*exchange)(void *, int, int), void *ctx);
int nla_k_crs_chain_sh(int obj, int n, int ncols, int ld, int ldf, const double *X, int64_t i0, double f_best, const int32_t *jn_ring,
                       const int32_t *pos_ring, const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks,
                       uint64_t first_block, int K, const int64_t *W, const double *Wf, int nW, int w_on_host, int slot_mask,
                       const double *lb, const double *ub, double *TX, double *TM, void *ctrl, uint32_t ticket_base,
                       nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero,
                       const void *table, uint32_t seq, uint32_t stopbits, int grid_cap, void *stream)
{
    const emu_shard *S = (const emu_shard *) table;
    emu_sh_ctx c;
    uint32_t forced = 0, timed = 0;
    int nc;
    (void) w_on_host; (void) ctrl; (void) ticket_base; (void) ctrl_is_zero; (void) grid_cap; (void) stream;
    EMU_LAUNCH();
    if (!S || K > 256 || nW > 256 || obj < 0 || ldf != S->ldf) return EMU_ERR;
    nc = n - S->c0 < ncols ? n - S->c0 : ncols;                     /* (the kernel's pad column is not needed here) */
    c.S = S; c.TX = TX; c.ncols = nc; c.mychunks = nla_crs_chain_sh_chunks(n, ncols); c.seq = seq; c.failed = 0;
    for (int r = 0; r < S->world; ++r)
        __atomic_store_n(S->peerstop[r] + (seq & 1u) * EMU_SH_MAXW + (uint32_t) S->rank, (seq << 2) | (stopbits & 3u), __ATOMIC_RELEASE);
    (void) orc_k_crs_chain_cols(obj, n, nc, S->c0, ld, ldf, X, i0, f_best, S->xbest, jn_ring, pos_ring, last_ring, words_ring, ring_blocks, first_block, K,
                                W, Wf, nW, slot_mask, lb, ub, S->lbf, S->ubf, TX, TM, (orc_slot_status *) status, fwcnt, fwrec, fwcap, NULL, NULL,
                                emu_sh_exchange, &c);
    for (int r = 0; r < S->world && !c.failed; ++r) {
        const uint32_t *w = S->peerstop[S->rank] + (seq & 1u) * EMU_SH_MAXW + (uint32_t) r;
        if (emu_sh_wait(w, seq << 2, 1)) { c.failed = 1; break; }
        forced |= __atomic_load_n(w, __ATOMIC_ACQUIRE) & 1u; timed |= (__atomic_load_n(w, __ATOMIC_ACQUIRE) >> 1) & 1u;
    }
    status[K].fT = forced ? 1. : 0.; status[K].fM = timed ? 1. : 0.; status[K].t = c.failed; status[K].pad = 0;
    return 0;
}
int nla_k_crs_commit_sh(int nc, int ld, int ldf, int c0, double *X, const double *TX, const double *TM, int ncommit,
                        const int32_t *h_slot, const int32_t *h_kind, const int64_t *h_row, void *zero, size_t zero_bytes,
                        int n, int best_slot, int best_kind, double *xbest, void *stream)
{
    (void) stream;
    EMU_LAUNCH();
    if (zero && zero_bytes) memset(zero, 0, zero_bytes);
    if (best_slot >= 0) memcpy(xbest, (best_kind == 1 ? TX : TM) + (size_t) best_slot * (size_t) ldf, sizeof(double) * (size_t) n);
    for (int c = 0; c < ncommit; ++c)
        memcpy(X + (size_t) h_row[c] * (size_t) ld, (h_kind[c] == 1 ? TX : TM) + (size_t) h_slot[c] * (size_t) ldf + (size_t) c0, sizeof(double) * (size_t) nc);
    return 0;
}
void nla_stream_destroy(void *st) { emu_release(st); }
int nla_stream_sync(void *st) { (void) st; return 0; }
int nla_stream_query(void *st) { (void) st; return 0; }
void *nla_event_create(void) { return emu_alloc(1); }
void nla_event_destroy(void *ev) { emu_release(ev); }
int nla_event_record(void *ev, void *st) { (void) ev; (void) st; return 0; }
int nla_event_sync(void *ev) { (void) ev; return 0; }
float nla_event_elapsed_ms(void *a, void *b) { (void) a; (void) b; return 0.f; }
int nla_stream_wait_event(void *st, void *ev) { (void) st; (void) ev; return 0; }
/* (the emulated device runs everything at once, in program order: what a gate would wait for has not been launched yet) */
int nla_k_gate(const int32_t *counter, int32_t from, int32_t need, double timeout_ms, int32_t *gave_up, void *st)
{
    (void) counter; (void) from; (void) timeout_ms; (void) st;
    if (need > 0) EMU_LAUNCH();
    if (need > 0 && gave_up && getenv("NLA_EMU_GATE_TIMEOUT")) *gave_up = 1;      /* (tests: a gate that gave up) */
    return 0;
}
const char *nla_dev_error_string(int err) { return err == EMU_ERR ? "not provided by the emulated device" : "emulated device error"; }

static double urand_from(double a, double b, uint32_t w0, uint32_t w1)       /* mt19937ar.c:186-206 */
{
    const double r = ((double) (w0 >> 5) * 67108864.0 + (double) (w1 >> 6)) * (1.0 / 9007199254740992.0);
    return a + (b - a) * r;
}

/* ---- MT19937 word stream (hip/mt_kernels.hip) ----------------------------------------------------------------------------- */
int nla_k_mt_jump(const uint64_t *poly, const uint32_t *src, uint32_t *dst, int count, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int i = 0; i < count; ++i) nla_mt_apply_jump_host(poly, src + (size_t) i * NLA_MT_N, dst + (size_t) i * NLA_MT_N);
    return 0;
}
int nla_k_mt_generate_seg(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_first, uint64_t count, uint32_t *out,
                          int seg_regens, void *st)
{
    EMU_LAUNCH();
    const uint64_t g_end = g_first + count;
    (void) st;
    if (seg_regens < 1 || seg_regens > NLA_MT_SEG_REGENS || (seg_regens & (seg_regens - 1))) return 1;
    for (int s = 0; s < nseg; ++s) {
        uint32_t mt[NLA_MT_N];
        const uint64_t g0 = (seg_first + (uint64_t) s) * ((uint64_t) NLA_MT_N * (uint64_t) seg_regens);
        memcpy(mt, seg_states + (size_t) s * NLA_MT_N, sizeof mt);
        for (int r = 0; r < seg_regens; ++r) {
            const uint64_t gb = g0 + (uint64_t) r * NLA_MT_N;
            if (gb >= g_end) break;
            if (gb + NLA_MT_N > g_first)
                for (int i = 0; i < NLA_MT_N; ++i) { const uint64_t g = gb + (uint64_t) i; if (g >= g_first && g < g_end) out[g - g_first] = nla_mt_temper(mt[i]); }
            nla_mt_regen(mt);
        }
    }
    return 0;
}

int nla_k_mt_generate(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_first, uint64_t count, uint32_t *out, void *st)
{
    return nla_k_mt_generate_seg(seg_states, seg_first, nseg, g_first, count, out, NLA_MT_SEG_REGENS, st);
}

int nla_k_mt_rankbits(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first, uint64_t count,
                      int64_t popm1, int64_t rowwords, uint64_t *bits, void *st)
{
    /* the fused kernel's contract, stated the slow way: the words, then one bit per pair */
    uint32_t *w;
    int rc;
    if (nseg <= 0 || count == 0 || popm1 <= 0) return 0;
    w = (uint32_t *) malloc(sizeof(uint32_t) * (size_t) count);
    if (!w) return EMU_ERR;
    rc = nla_k_mt_generate(seg_states, seg_first, nseg, g_first, count, w, st);
    for (uint64_t k = 0; !rc && k + 1 < count; k += 2) {
        const uint64_t s = (g_first + k - g_rank0) >> 1, row = s / (uint64_t) popm1, j = s - row * (uint64_t) popm1;
        if (urand_from(0., 1., w[k], w[k + 1]) < 0.45) bits[(size_t) row * (size_t) rowwords + (j >> 6)] |= 1ULL << (j & 63);
    }
    free(w);
    return rc;
}

int nla_rankbits_gate_target(uint64_t g_rank0, int64_t popm1, int64_t nrows, int64_t c)
{
    const uint64_t b0 = g_rank0 + 128ULL * (uint64_t) popm1 * (uint64_t) c;
    const int64_t r1 = 64 * (c + 1) < nrows ? 64 * (c + 1) : nrows;
    const uint64_t b1 = g_rank0 + 2ULL * (uint64_t) popm1 * (uint64_t) r1;
    if (b1 <= b0) return 0;
    return (int) ((b1 - 1) / NLA_MT_SEG_WORDS - b0 / NLA_MT_SEG_WORDS + 1);
}
int nla_k_mt_rankbits_gated(const uint32_t *seg_states, uint64_t seg_first, int nseg, uint64_t g_rank0, uint64_t g_first, uint64_t count,
                            int64_t popm1, int64_t rowwords, uint64_t *bits, int *gate, int *ticket, int waves_per_cu, void *st)
{
    /* the synchronous device: all bits, then every gate of the sweeps covered at its target */
    int rc = nla_k_mt_rankbits(seg_states, seg_first, nseg, g_rank0, g_first, count, popm1, rowwords, bits, st);
    (void) waves_per_cu;
    if (!rc && ticket) *ticket += nseg;
    if (!rc && gate && count) {
        const int64_t r0 = (int64_t) ((g_first - g_rank0) / (2ULL * (uint64_t) popm1)), r1 = (int64_t) ((g_first + count - g_rank0) / (2ULL * (uint64_t) popm1));
        for (int64_t c = r0 / 64; c <= (r1 - 1) / 64; ++c) gate[c] = nla_rankbits_gate_target(g_rank0, popm1, r1, c);
    }
    return rc;
}

/* ---- rows from the stream, evaluation (hip/crs_kernels.hip: crs_init_rows_kernel, eval_kernel) ------------------------------ */
/* NLA_OBJ_NEGATE: the flag stripped from obj, the factor f is multiplied by */
static double emu_obj_sign(int *obj)
{
    if (*obj >= 0 && (*obj & NLA_OBJ_NEGATE)) { *obj &= ~NLA_OBJ_NEGATE; return -1.; }
    return 1.;
}
int nla_k_crs_init_rows(int obj, int n, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t row_first,
                        int64_t nrows, double *X, double *F, void *st)
{
    EMU_LAUNCH();
    const double sign = emu_obj_sign(&obj);
    (void) st;
    double *tmp = X ? NULL : (double *) malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));      /* X == NULL: the values only */
    for (int64_t r = 0; r < nrows; ++r) {
        double *x = X ? X + (size_t) (row_first + r) * (size_t) ld : tmp;
        const uint32_t *w = words + (size_t) r * 2 * (size_t) n;
        for (int i = 0; i < n; ++i) x[i] = urand_from(lb[i], ub[i], w[2 * i], w[2 * i + 1]);
        if (obj >= 0) F[row_first + r] = sign * nla_obj_eval_seq(obj, (unsigned) n, x, NULL);
    }
    free(tmp);
    return 0;
}
int nla_k_eval(int obj, int n, int ld, const double *P, int64_t count, double *F, void *st)
{
    EMU_LAUNCH();
    const double sign = emu_obj_sign(&obj);
    (void) st;
    for (int64_t c = 0; c < count; ++c) F[c] = sign * nla_obj_eval_seq(obj, (unsigned) n, P + (size_t) c * (size_t) ld, NULL);
    return 0;
}

/* ---- MLSL (hip/mlsl_kernels.hip) --------------------------------------------------------------------------------------------- */
int nla_k_mlsl_sobol_rows(int n, int ld, const double *lb, const double *ub, const uint32_t *V, uint32_t index_first, int count, double *P, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int r = 0; r < count; ++r) {
        const uint32_t k = index_first + (uint32_t) r;
        for (int i = 0; i < n; ++i) {
            uint32_t g = k ^ (k >> 1), acc = 0;
            while (g) { const int c = __builtin_ctz(g); acc ^= V[(size_t) c * (size_t) n + (size_t) i]; g &= g - 1; }
            P[(size_t) r * (size_t) ld + (size_t) i] = lb[i] + (ub[i] - lb[i]) * ((double) acc / 4294967296.0);
        }
    }
    return 0;
}
int nla_k_mlsl_dist2(int n, int ld, const double *A, int na, const double *B, int nb, double *D, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) {
            double d = 0.;
            for (int k = 0; k < n; ++k) { const double dx = A[(size_t) i * ld + k] - B[(size_t) j * ld + k]; d += dx * dx; }   /* mlsl.c:118-127 */
            D[(size_t) i * (size_t) nb + (size_t) j] = d;
        }
    return 0;
}
int nla_k_mlsl_rowmin(const double *D, int ldd, int na, int nb, const double *FA, const double *FB, const double *init, double *out, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int i = 0; i < na; ++i) {
        double m = HUGE_VAL;
        for (int j = 0; j < nb; ++j) if (FB[j] < FA[i]) { const double d = D[(size_t) i * ldd + j]; m = d < m ? d : m; }
        { const double b = init ? init[i] : HUGE_VAL; out[i] = m < b ? m : b; }
    }
    return 0;
}
int nla_k_mlsl_colmin(const double *D, int ldd, int na, int nb, const double *FA, const double *FB, const int32_t *skip, double *inout, void *st)
{
    EMU_LAUNCH();
    (void) st;
    if (na <= 0) return 0;
    for (int j = 0; j < nb; ++j) {
        double m;
        if (skip && skip[j]) continue;
        m = inout[j];
        for (int i = 0; i < na; ++i) if (FA[i] < FB[j]) { const double d = D[(size_t) i * ldd + j]; m = d < m ? d : m; }
        inout[j] = m;
    }
    return 0;
}
int nla_k_mlsl_gather_pairs_t(const double *D, int ldd, const int64_t *rows, int nr, const int64_t *cols, int nc, double *out, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int a = 0; a < nr; ++a) for (int b = 0; b < nc; ++b) out[(size_t) b * nr + a] = D[(size_t) rows[a] * (size_t) ldd + (size_t) cols[b]];
    return 0;
}
int nla_k_mlsl_gather_pairs(const double *D, int ldd, const int64_t *rows, int nr, const int64_t *cols, int nc, double *out, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int a = 0; a < nr; ++a) for (int b = 0; b < nc; ++b) out[(size_t) a * nc + b] = D[(size_t) rows[a] * (size_t) ldd + (size_t) cols[b]];
    return 0;
}
int nla_k_mlsl_gather_rows(int n, int ld, const double *src, const int64_t *idx, int count, double *dst, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int c = 0; c < count; ++c) memmove(dst + (size_t) c * ld, src + (size_t) idx[c] * (size_t) ld, sizeof(double) * (size_t) n);
    return 0;
}
int nla_k_mlsl_near_bound(int n, int ld, const double *P, const int64_t *idx, int count, const double *lb, const double *ub, double thr,
                          int32_t *flags, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int c = 0; c < count; ++c) {
        const double *x = P + (size_t) idx[c] * (size_t) ld;
        int hit = 0;
        for (int j = 0; j < n && !hit; ++j) if ((x[j] - lb[j] <= thr || ub[j] - x[j] <= thr) && ub[j] - lb[j] > thr) hit = 1;   /* mlsl.c:211-218 */
        flags[c] = hit;
    }
    return 0;
}

int nla_k_mlsl_negate(double *F, int count, void *st) { EMU_LAUNCH(); (void) st; for (int i = 0; i < count; ++i) F[i] = -F[i]; return 0; }

/* ---- local optimisers: the oracle ports stand in for the batched kernels ------------------------------------------------------
 * Device objectives: a whole search per call, as the kernels do.  External evaluation (NLA_OBJ_EXTERNAL, include/nlopt_amd.h):
 * the port runs as a coroutine (ucontext) whose objective yields to the launcher's caller with the point in EX and is resumed by
 * the next launch with f / gradient in EF / EG — the kernels' protocol, call for call. */
typedef struct { int obj; long calls; double sign; double *ft; long ftcap; } emu_obj;
static double emu_objective(unsigned n, const double *x, double *grad, void *p)
{
    emu_obj *o = (emu_obj *) p;
    double f;
    ++o->calls;
    f = nla_obj_eval_seq(o->obj, n, x, grad);
    if (o->sign < 0) { f = -f; if (grad) for (unsigned i = 0; i < n; ++i) grad[i] = -grad[i]; }
    if (o->ft && o->calls <= o->ftcap) o->ft[o->calls - 1] = f;
    return f;
}
size_t nla_lbfgs_work_doubles(int ld, int mf, int count) { (void) ld; (void) mf; (void) count; return 8; }
size_t nla_lbfgs_hist_doubles(int ld, int mf, int count) { (void) ld; (void) mf; (void) count; return 8; }
size_t nla_mma_work_doubles(int ld, int count) { (void) ld; (void) count; return 8; }
static void local_stop(orc_stop *s, int n, double minf_max, double ftol_rel, double ftol_abs, double xtol_rel, int maxeval,
                       const double *xtol_abs, const double *x_weights)
{
    orc_stop_default(s, (unsigned) n);
    s->minf_max = minf_max; s->ftol_rel = ftol_rel; s->ftol_abs = ftol_abs; s->xtol_rel = xtol_rel; s->maxeval = maxeval; s->nevals = 0;
    s->xtol_abs = xtol_abs; s->x_weights = x_weights;
}

#include <ucontext.h>
#define EMU_CO_STACK (1 << 20)
typedef struct emu_co {
    ucontext_t co, caller;
    char *stack;
    int alg, n, ld, mf, inst, inner_gradients;
    const double *lb, *ub, *sigma_init;
    double *x;
    double tolg;
    orc_stop s;
    orc_mma_params m;
    nla_lbfgs_result *out;
    nla_local_ext E;
    long calls;
    double *ft; long ftcap;
} emu_co;
size_t nla_lbfgs_save_bytes(void) { return sizeof(emu_co *); }
size_t nla_mma_save_bytes(void) { return sizeof(emu_co *); }

static double emu_co_objective(unsigned n, const double *x, double *grad, void *p)
{
    emu_co *c = (emu_co *) p;
    /* LD_MMA's repeated call with a gradient (mma.c:337-339) is the one that asks for a gradient although inner_gradients = 0 */
    const int uncounted = c->alg == 1 && !c->inner_gradients && grad && c->calls > 0;
    memcpy(c->E.EX + (size_t) c->inst * c->ld, x, sizeof(double) * n);
    c->E.req[c->inst].state = 1;
    c->E.req[c->inst].want_grad = (grad ? 1 : 0) | (uncounted ? 2 : 0);
    ++c->calls;
    swapcontext(&c->co, &c->caller);
    if (grad) memcpy(grad, c->E.EG + (size_t) c->inst * c->ld, sizeof(double) * n);
    if (c->ft && c->calls <= c->ftcap) c->ft[c->calls - 1] = c->E.EF[c->inst];
    return c->E.EF[c->inst];
}
static void emu_co_main(unsigned lo, unsigned hi)
{
    emu_co *c = (emu_co *) (((uintptr_t) hi << 32) | (uintptr_t) lo);
    double minf = HUGE_VAL;
    nla_lbfgs_result *o = c->out + c->inst;
    if (c->alg == 1) o->ret = orc_mma_minimize(c->n, emu_co_objective, c, c->lb, c->ub, c->x, &minf, &c->s, &c->m);
    else o->ret = orc_lbfgs_minimize(c->n, emu_co_objective, c, c->lb, c->ub, c->x, &minf, &c->s, c->mf, c->tolg);
    o->f = minf; o->nevals = (int32_t) c->s.nevals; o->iterm = c->alg == 1 ? (int32_t) c->calls : 0; o->cols = 0;
    c->E.req[c->inst].state = 2;
    swapcontext(&c->co, &c->caller);
}
/* one launch of an external-evaluation batch: start (resume = 0) or continue (resume = 1) every search that is not finished */
static int emu_co_launch(int alg, int n, int ld, int mf, int count, const double *lb, const double *ub, const double *sigma_init, double *X,
                         const nla_lbfgs_params *PL, const nla_mma_params *PM, nla_lbfgs_result *out, const nla_local_ext *ext)
{
    emu_co **slot;
    if (!ext || !ext->req || !ext->EX || !ext->EG || !ext->EF || !ext->save) return EMU_ERR;
    slot = (emu_co **) ext->save;
    for (int i = 0; i < count; ++i) {
        emu_co *volatile c;             /* volatile: getcontext() may return twice */
        if (!ext->resume) {
            c = (emu_co *) calloc(1, sizeof *c);
            if (!c || !(c->stack = (char *) malloc(EMU_CO_STACK))) { free(c); return EMU_ERR; }
            slot[i] = c;
            c->alg = alg; c->n = n; c->ld = ld; c->mf = mf; c->inst = i; c->lb = lb; c->ub = ub; c->sigma_init = sigma_init;
            c->x = X + (size_t) i * ld; c->out = out; c->E = *ext;
            if (alg == 1) {
                local_stop(&c->s, n, PM->minf_max, PM->ftol_rel, PM->ftol_abs, PM->xtol_rel, PM->maxeval, PM->xtol_abs, PM->x_weights);
                c->m.rho_init = PM->rho_init; c->m.sigma_min = PM->sigma_min; c->m.inner_maxeval = PM->inner_maxeval;
                c->m.inner_gradients = PM->inner_gradients; c->m.always_improve = PM->always_improve; c->m.sigma_init = sigma_init;
                c->inner_gradients = PM->inner_gradients;
                if (PM->ftrace) { c->ft = PM->ftrace + (size_t) i * (size_t) PM->ftrace_cap; c->ftcap = (long) PM->ftrace_cap; }
            } else {
                local_stop(&c->s, n, PL->minf_max, PL->ftol_rel, PL->ftol_abs, PL->xtol_rel, PL->maxeval, PL->xtol_abs, PL->x_weights);
                c->tolg = PL->tolg;
                if (PL->ftrace) { c->ft = PL->ftrace + (size_t) i * (size_t) PL->ftrace_cap; c->ftcap = (long) PL->ftrace_cap; }
            }
            getcontext(&c->co);
            c->co.uc_stack.ss_sp = c->stack; c->co.uc_stack.ss_size = EMU_CO_STACK; c->co.uc_link = NULL;
            makecontext(&c->co, (void (*)(void)) emu_co_main, 2, (unsigned) ((uintptr_t) c & 0xffffffffu), (unsigned) ((uintptr_t) c >> 32));
        } else {
            c = slot[i];
            if (!c || ext->req[i].state != 1) continue;
            ext->req[i].state = 0;
        }
        c->s.force_stop = ext->forced;
        if (ext->timeout) { c->s.maxtime = 1e-300; c->s.start = -1e300; }
        swapcontext(&c->caller, &c->co);
        if (ext->req[i].state == 2) { free(c->stack); free(c); slot[i] = NULL; }
    }
    return 0;
}

/* test hook: the summation order (params.exact) the host drivers asked for in the last local-search launch — the emulation
 * itself always sums in the reference's order */
int nla_emu_last_exact = -1;

int nla_k_lbfgs_batch(int obj, int n, int ld, int mf, int count, const double *lb, const double *ub, double *X, double *work, int *iwork,
                      double *hist, const nla_lbfgs_params *P, nla_lbfgs_result *out, const nla_local_ext *ext, void *st)
{
    EMU_LAUNCH();
    nla_emu_last_exact = P->exact;
    (void) work; (void) iwork; (void) hist; (void) st;
    if (obj == NLA_OBJ_EXTERNAL) return emu_co_launch(0, n, ld, mf, count, lb, ub, NULL, X, P, NULL, out, ext);
    for (int i = 0; i < count; ++i) {
        emu_obj o = { obj, 0, P->sign, P->ftrace ? P->ftrace + (size_t) i * (size_t) P->ftrace_cap : NULL, (long) P->ftrace_cap };
        orc_stop s;
        double minf = HUGE_VAL;
        local_stop(&s, n, P->minf_max, P->ftol_rel, P->ftol_abs, P->xtol_rel, P->maxeval, P->xtol_abs, P->x_weights);
        if (P->abort && *P->abort == -999) s.force_stop = 1;
        out[i].ret = orc_lbfgs_minimize(n, emu_objective, &o, lb, ub, X + (size_t) i * ld, &minf, &s, mf, P->tolg);
        out[i].f = minf; out[i].nevals = (int32_t) s.nevals; out[i].iterm = 0; out[i].cols = 0;
    }
    return 0;
}
int nla_k_mma_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *sigma_init, double *X, double *work,
                    const nla_mma_params *P, nla_lbfgs_result *out, const nla_local_ext *ext, void *st)
{
    EMU_LAUNCH();
    nla_emu_last_exact = P->exact;
    (void) work; (void) st;
    if (obj == NLA_OBJ_EXTERNAL) return emu_co_launch(1, n, ld, 0, count, lb, ub, sigma_init, X, NULL, P, out, ext);
    for (int i = 0; i < count; ++i) {
        emu_obj o = { obj, 0, P->sign, P->ftrace ? P->ftrace + (size_t) i * (size_t) P->ftrace_cap : NULL, (long) P->ftrace_cap };
        orc_stop s;
        orc_mma_params m;
        double minf = HUGE_VAL;
        local_stop(&s, n, P->minf_max, P->ftol_rel, P->ftol_abs, P->xtol_rel, P->maxeval, P->xtol_abs, P->x_weights);
        if (P->abort && *P->abort == -999) s.force_stop = 1;
        memset(&m, 0, sizeof m);
        m.rho_init = P->rho_init; m.sigma_min = P->sigma_min; m.inner_maxeval = P->inner_maxeval; m.inner_gradients = P->inner_gradients;
        m.always_improve = P->always_improve; m.sigma_init = sigma_init;
        out[i].ret = orc_mma_minimize(n, emu_objective, &o, lb, ub, X + (size_t) i * ld, &minf, &s, &m);
        out[i].f = minf; out[i].nevals = (int32_t) s.nevals; out[i].iterm = (int32_t) o.calls; out[i].cols = 0;
    }
    return 0;
}

/* ---- code objects (hip/devrt.hip): none on the emulated device ------------------------------------------------------------- */
/* LN_COBYLA batched (hip/cobyla_kernels.hip): every start through the product's own host COBYLA as nlopt_optimize reaches it (default
 * initial step, memoized best point), the objective in the host callback's summation order */
typedef struct { int obj; double sign; } emu_cob_obj;
static double emu_cob_f(unsigned n, const double *x, double *g, void *p) { emu_cob_obj *o = (emu_cob_obj *) p; (void) g; return o->sign * nla_obj_eval_seq(o->obj, n, x, NULL); }
size_t nla_cobyla_work_doubles(int n, int ld, int count) { (void) n; (void) ld; (void) count; return 8; }
size_t nla_cobyla_work_ints(int n, int count) { (void) n; (void) count; return 8; }
/* (the device kernel's LDS budget, hip/cobyla_kernels.hip cw_lds_doubles with m = 2n: the host logic must take the same decisions here) */
size_t nla_cobyla_lds_bytes(int n)
{
    const size_t m = 2 * (size_t) n, ldn = (size_t) (n | 1), ldd = (m + 2) | 1, v = m + 2 > (size_t) n + 1 ? m + 2 : (size_t) n + 1;
    return 8 * (ldn * (size_t) (n + 1) + ldn * (size_t) n + ldd * (size_t) (n + 1) + ldn * (m + 1) + ldn * (size_t) n + 14 * (size_t) n + 6 * v + (m + 2 + (size_t) n + 1) / 2 + 1) + 2048;
}
int nla_cobyla_fits(int n) { return n >= 1 && nla_cobyla_lds_bytes(n) <= 160 * 1024; }
int nla_k_cobyla_batch(int obj, int n, int ld, int count, const double *lb, const double *ub, const double *dx, double *X,
                       double *work, int *iwork, const nla_cobyla_params *P, nla_lbfgs_result *out, void *st)
{
    EMU_LAUNCH();
    (void) work; (void) iwork; (void) st;
    if (!nla_cobyla_fits(n)) return EMU_ERR;                 /* (as the kernel: the state of a search must fit the LDS) */
    nla_emu_last_exact = P->exact;
    for (int i = 0; i < count; ++i) {
        emu_cob_obj o = { obj & 0xff, (P->sign == 0. ? 1. : P->sign) * ((obj & 0x100) ? -1. : 1.) };
        nlopt_opt loc = nlopt_create(NLOPT_LN_COBYLA, (unsigned) n);
        double minf = HUGE_VAL;
        if (!loc) return EMU_ERR;
        nlopt_set_min_objective(loc, emu_cob_f, &o);
        nlopt_set_lower_bounds(loc, lb); nlopt_set_upper_bounds(loc, ub);
        nlopt_set_stopval(loc, P->minf_max); nlopt_set_ftol_rel(loc, P->ftol_rel); nlopt_set_ftol_abs(loc, P->ftol_abs); nlopt_set_xtol_rel(loc, P->xtol_rel);
        if (P->xtol_abs) nlopt_set_xtol_abs(loc, P->xtol_abs);
        nlopt_set_maxeval(loc, P->maxeval);
        if (dx) nlopt_set_initial_step(loc, dx);
        if (P->abort && *P->abort == -999) { out[i].ret = NLOPT_FORCED_STOP; out[i].f = minf; out[i].nevals = out[i].iterm = 0; out[i].cols = 0; nlopt_destroy(loc); continue; }
        out[i].ret = nlopt_optimize(loc, X + (size_t) i * (size_t) ld, &minf);
        out[i].f = minf; out[i].nevals = out[i].iterm = nlopt_get_numevals(loc); out[i].cols = 0;
        nlopt_destroy(loc);
    }
    return 0;
}

void *nla_module_load_file(const char *path) { (void) path; return NULL; }
void *nla_module_load_data(const void *image) { (void) image; return NULL; }
void nla_module_unload(void *module) { (void) module; }
void *nla_module_function(void *module, const char *name) { (void) module; (void) name; return NULL; }
int nla_module_launch(void *function, unsigned gx, unsigned bx, void **params, void *st)
{ (void) function; (void) gx; (void) bx; (void) params; (void) st; return EMU_ERR; }

/* ---- ISRES (hip/isres_kernels.hip) ---------------------------------------------------------------------------------------------- */
int nla_k_isres_init(int n, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t k_first, int64_t count,
                     const double *x0, double *X, double *S, void *st)
{
    EMU_LAUNCH();
    const double sq = sqrt((double) n);
    (void) st;
    for (int64_t kl = 0; kl < count; ++kl) {
        const int64_t k = k_first + kl;
        const uint32_t *w = words + (size_t) kl * 2 * (size_t) n;
        for (int j = 0; j < n; ++j) {
            X[(size_t) k * ld + j] = (k == 0) ? x0[j] : urand_from(lb[j], ub[j], w[2 * j], w[2 * j + 1]);
            S[(size_t) k * ld + j] = (ub[j] - lb[j]) / sq;
        }
    }
    return 0;
}
int nla_k_isres_eval(int obj, int n, int ld, const double *X, int64_t pop, int m, int p, const nla_dev_constraint *con, double *F,
                     double *PEN, double *GPEN, int32_t *FEAS, void *st)
{
    EMU_LAUNCH();
    const double sign = emu_obj_sign(&obj);
    (void) st;
    for (int64_t k = 0; k < pop; ++k) {                                       /* isres.c:138-166 */
        const double *x = X + (size_t) k * (size_t) ld;
        double pen = 0, gpen = 0;
        int feas = 1;
        F[k] = sign * nla_obj_eval_seq(obj, (unsigned) n, x, NULL);
        for (int c = 0; c < m + p; ++c) {
            double g = nla_con_blocksum_seq((unsigned) n, x, NULL, con[c].q, con[c].Q);
            if (c == m) gpen = pen;
            if (c < m) { if (g > con[c].tol) feas = 0; if (g < 0) g = 0; pen += g * g; }
            else { if (fabs(g) > con[c].tol) feas = 0; pen += g * g; }
        }
        if (p == 0) gpen = pen;
        PEN[k] = pen; GPEN[k] = gpen; FEAS[k] = feas;
    }
    return 0;
}
/* ranking elements: own layout (rank_count and stochrank below are the only users): [63] penalty == 0 | [62:42] rank of the
 * penalty | [41:21] rank of f | [20:0] individual */
#define EL(idx, rf, rp, z) (((uint64_t) ((z) ? 1 : 0) << 63) | ((uint64_t) (rp) << 42) | ((uint64_t) (rf) << 21) | (uint64_t) (idx))
#define EL_IDX(e) ((uint32_t) ((e) & 0x1FFFFFu))
#define EL_RF(e) ((uint32_t) (((e) >> 21) & 0x1FFFFFu))
#define EL_RP(e) ((uint32_t) (((e) >> 42) & 0x1FFFFFu))
#define EL_Z(e) ((int) ((e) >> 63))
int nla_k_isres_rank_count(int64_t pop, const double *F, const double *PEN, uint64_t *elems, int32_t *sorted, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int64_t k = 0; k < pop; ++k) {
        uint32_t rf = 0, rp = 0, ps = 0;
        for (int64_t j = 0; j < pop; ++j) {
            rf += F[j] < F[k];
            rp += PEN[j] < PEN[k];
            ps += (F[j] < F[k]) || (F[j] == F[k] && j < k);                   /* stable sort position (qsort_r.c:190, merge sort) */
        }
        elems[k] = EL((uint32_t) k, rf, rp, PEN[k] == 0);
        sorted[ps] = (int32_t) k;
    }
    return 0;
}
int nla_k_isres_bits(const uint32_t *words, int64_t row_first, int nrows, int64_t pop, uint64_t *bits, void *st)
{
    EMU_LAUNCH();
    const int64_t popm1 = pop - 1, rowwords = (popm1 + 63) / 64;
    (void) st;
    if (popm1 <= 0) return 0;
    for (int r = 0; r < nrows; ++r) {
        uint64_t *row = bits + (size_t) (row_first + r) * (size_t) rowwords;
        const uint32_t *w = words + (size_t) r * 2 * (size_t) popm1;
        memset(row, 0, sizeof(uint64_t) * (size_t) rowwords);
        for (int64_t j = 0; j < popm1; ++j)
            if (urand_from(0., 1., w[2 * j], w[2 * j + 1]) < 0.45) row[j >> 6] |= 1ULL << (j & 63);        /* PF, isres.c:72,210 */
    }
    return 0;
}
int nla_k_isres_stochrank(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits, int *ticket,
                          uint8_t *swapped, int32_t *irank, void *st);
/* (launches run synchronously here: by the time the ranking is "launched", every block of bits enqueued before it is complete) */
int nla_isres_stochrank_handoff(void) { return 32; }
int nla_k_isres_stochrank_gated(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits, int *ticket,
                                uint8_t *swapped, int32_t *irank, const int *gate, uint64_t gate_g_rank0, int64_t gate_nrows, void *st)
{
    if (gate) for (int64_t u = 0; u < (nsweeps + 63) / 64; ++u) if (gate[u] < nla_rankbits_gate_target(gate_g_rank0, pop - 1, gate_nrows, u)) return 1;
    return nla_k_isres_stochrank(pop, nsweeps, streams, progress, bits, ticket, swapped, irank, st);
}
int nla_k_isres_stochrank(int64_t pop, int64_t nsweeps, uint64_t *streams, int *progress, const uint64_t *bits, int *ticket,
                          uint8_t *swapped, int32_t *irank, void *st)
{
    EMU_LAUNCH();
    const int64_t rowwords = (pop - 1 + 63) / 64;
    /* the kernel leaves streams[0..pop) — the elements in initial order — untouched (the driver re-runs the ranking with fewer
     * sweeps after an early exit, isres.c:227): sweep on a copy */
    uint64_t *cur = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (pop > 0 ? pop : 1));
    (void) progress; (void) ticket; (void) st;
    if (!cur) return EMU_ERR;
    memcpy(cur, streams, sizeof(uint64_t) * (size_t) pop);
    for (int64_t i = 0; i < nsweeps && pop > 1; ++i) {                        /* isres.c:206-228, every sweep (the driver cuts) */
        int sw = 0;
        for (int64_t j = 0; j < pop - 1; ++j) {
            const uint64_t a = cur[j], b = cur[j + 1];
            const int ulow = (int) ((bits[(size_t) i * (size_t) rowwords + (size_t) (j >> 6)] >> (j & 63)) & 1);
            const int gt = (ulow || (EL_Z(a) && EL_Z(b))) ? EL_RF(a) > EL_RF(b) : EL_RP(a) > EL_RP(b);
            if (gt) { cur[j] = b; cur[j + 1] = a; sw = 1; }
        }
        swapped[i] = (uint8_t) sw;
    }
    for (int64_t k = 0; k < pop; ++k) irank[k] = (int32_t) EL_IDX(cur[k]);
    free(cur);
    return 0;
}
int nla_k_isres_nrand(const uint32_t *words, int64_t nattempts, int64_t attempt_base, int32_t *counts, int64_t *ztotal, int64_t zbase,
                      double *z, int64_t *zatt, void *st)
{
    EMU_LAUNCH();
    int64_t cnt = 0;
    (void) counts; (void) st;
    for (int64_t a = 0; a < nattempts; ++a) {                                 /* nlopt_nrand(0,1), mt19937ar.c:216-232 */
        const uint32_t *w = words + 4 * a;
        const double v1 = urand_from(-1., 1., w[0], w[1]), v2 = urand_from(-1., 1., w[2], w[3]);
        const double s = v1 * v1 + v2 * v2;
        if (s >= 1.0) continue;
        z[zbase + cnt] = (s == 0) ? 0.0 : 0.0 + v1 * sqrt(-2 * log(s) / s) * 1.0;
        zatt[zbase + cnt] = attempt_base + a;
        ++cnt;
    }
    *ztotal += cnt;
    return 0;
}
int nla_k_isres_inverse(int64_t pop, const int32_t *irank, int32_t *inv, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int64_t k = 0; k < pop; ++k) inv[irank[k]] = (int32_t) k;
    return 0;
}
/* the serial evolve kernel's contract: individuals state[0].. in order, deviates from z[state[1]..); an individual whose deviates
 * run out is not written at all and state[2] = 1 */
int nla_k_isres_evolve(int n, int ld, int phase, int64_t pop, int64_t survivors, int64_t zcount, double taup, double tau, const double *lb,
                       const double *ub, const double *z, const int32_t *irank, double *X, double *S, double *scratch, int64_t *state, void *st)
{
    EMU_LAUNCH();
    const double ALPHA = 0.2, GAMMA = 0.85, sqn = sqrt((double) n);
    int64_t k = state[0], pos = state[1], kend = phase == 0 ? pop : survivors;
    double *xo = (double *) malloc(sizeof(double) * 2 * (size_t) n), *so = xo + n;
    int ranout = 0;
    (void) st;
    if (!xo) return EMU_ERR;
    if (state[14] > 0 && state[14] < kend) kend = state[14];
    if (phase == 1 && k == 0) memcpy(scratch, X, sizeof(double) * (size_t) n);           /* memcpy(x0, xs, n), isres.c:253 */
    /* NLA_EMU_EVOLVE_DUMP=<file> (analysis, tools/evolve_predict.py): per individual of the mutation phase the deviates it consumed and
     * the analytic expectation of its redraws from the parent's x and sigma — what a start predictor could know in advance */
    static FILE *dump = NULL;
    static int dump_checked = 0;
    if (!dump_checked) { const char *pth = getenv("NLA_EMU_EVOLVE_DUMP"); dump_checked = 1; if (pth) dump = fopen(pth, "ab"); }
    for (; k < kend; ++k) {
        const int64_t rk = irank[k], ri = phase == 0 ? irank[k % survivors] : rk;
        const int lastsurv = (k + 1 == survivors);
        const double *xi = X + (size_t) ri * ld, *si = S + (size_t) ri * ld, *xk1 = X + (size_t) (k + 1) * ld;
        int64_t cur;
        double taup_rand;
        if (pos >= zcount) { ranout = 1; break; }
        taup_rand = taup * z[pos];
        cur = pos + 1;
        for (int j = 0; j < n && !ranout; ++j) {
            double xnew = xi[j], snew = si[j];
            int mutate = 1;
            if (phase == 1) {
                if (!lastsurv) xnew = xi[j] + GAMMA * (scratch[j] - xk1[j]);      /* physical row k+1, possibly rewritten (isres.c:260) */
                mutate = lastsurv || xnew < lb[j] || xnew > ub[j];
            }
            if (mutate) {
                const double sigmamax = (ub[j] - lb[j]) / sqn;
                double sg;
                int t = 1;
                if (cur + 1 >= zcount) { ranout = 1; break; }
                sg = si[j] * exp(taup_rand + tau * z[cur]);
                if (sg > sigmamax) sg = sigmamax;
                for (;;) {
                    if (cur + t >= zcount) { ranout = 1; break; }
                    xnew = xi[j] + sg * z[cur + t];
                    if (!(xnew < lb[j] || xnew > ub[j])) break;
                    ++t;
                }
                if (ranout) break;
                snew = si[j] + ALPHA * (sg - si[j]);
                cur += 1 + t;
            }
            xo[j] = xnew; so[j] = snew;
        }
        if (ranout) break;
        if (dump && phase == 0) {
            double rec[4] = { (double) k, (double) (cur - pos), 0., 0. };
            for (int j = 0; j < n; ++j) {            /* P(out of the box) with sigma' ~ sigma, and with sigma' = sigma exp(taup_rand) (known once the first deviate is) */
                const double s0 = si[j] > 0 ? si[j] : 1e-300, s1 = s0 * exp(taup_rand);
                const double p0 = 0.5 * erfc((xi[j] - lb[j]) / (s0 * 1.4142135623730951)) + 0.5 * erfc((ub[j] - xi[j]) / (s0 * 1.4142135623730951));
                const double p1 = 0.5 * erfc((xi[j] - lb[j]) / (s1 * 1.4142135623730951)) + 0.5 * erfc((ub[j] - xi[j]) / (s1 * 1.4142135623730951));
                rec[2] += p0 / (1. - p0); rec[3] += p1 / (1. - p1);
            }
            fwrite(rec, sizeof rec, 1, dump);
            if (k + 1 == kend) fflush(dump);
        }
        memcpy(X + (size_t) rk * ld, xo, sizeof(double) * (size_t) n);
        memcpy(S + (size_t) rk * ld, so, sizeof(double) * (size_t) n);
        pos = cur;
    }
    free(xo);
    state[0] = k; state[1] = pos; state[2] = ranout;
    return 0;
}
/* The multi-start evolve (hip/isres_evolve2.hip) resolves up to 256 consecutive individuals per round and hands an individual
 * it cannot resolve to the serial kernel (state[10] = 1).  With NLA_EMU_EVOLVE2 set the emulated device reports it as supported
 * and plays that protocol — rounds of at most 256 individuals, an occasional forced hand-over (a fixed function of the
 * individual's index), deviates running out mid-round — so that the driver's round / refill / fallback loop is exercised; the
 * arithmetic is the serial routine's. */
int nla_isres_evolve2_supported(int n) { return getenv("NLA_EMU_EVOLVE2") != NULL && n <= 1150; }
size_t nla_isres_evolve2_ws_bytes(int n) { (void) n; return 16; }
static int emu_forced_handover(int64_t k, int phase) { return (((uint32_t) k * 2654435761u + (uint32_t) phase * 977u) >> 7) % 53u == 0; }
int nla_k_isres_evolve_rounds(int n, int ld, int phase, int64_t pop, int64_t survivors, int64_t zcount, double taup, double tau,
                              const double *lb, const double *ub, const double *z, const int32_t *irank, const int32_t *inv, double *X,
                              double *S, const double *x0c, int64_t *state, double *rho, void *ws, const double *mu_rp, int rounds, void *st)
{
    EMU_LAUNCH();
    const int64_t kend = phase == 0 ? pop : survivors;
    (void) inv; (void) rho; (void) ws; (void) mu_rp;
    for (int r = 0; r < rounds; ++r) {
        int64_t first = state[0], limit, stop = first + 256 < kend ? first + 256 : kend;
        int rc;
        state[9] = 0;
        if (state[2] || state[10] || first >= kend) continue;
        if (emu_forced_handover(first, phase)) { state[10] = 1; continue; }
        for (limit = first + 1; limit < stop && !emu_forced_handover(limit, phase); ++limit) { }
        state[14] = limit; state[11] += 1;
        rc = nla_k_isres_evolve(n, ld, phase, pop, survivors, zcount, taup, tau, lb, ub, z, irank, X, S, (double *) x0c, state, st);
        state[14] = 0;
        if (rc) return rc;
        state[9] = state[0] - first;
    }
    return 0;
}

int nla_k_isres_evolve_parent_mu(int n, int ld, int64_t survivors, const double *lb, const double *ub, const int32_t *irank, const double *X,
                                 const double *S, double *mu_rp, void *st)
{
    EMU_LAUNCH();            /* (a prediction aid of the device's look-up rounds: nothing here depends on it) */
    (void) n; (void) ld; (void) lb; (void) ub; (void) irank; (void) X; (void) S; (void) st;
    for (int64_t p = 0; p < survivors; ++p) mu_rp[p] = 0.;
    return 0;
}

/* ---- CRS2_LM (hip/crs_kernels.hip): the per-kernel CPU references of port_kernels.c behind the launchers' ring / slot addressing,
 * so that crs_engine.c (batches, rings, pending commits, kernel-argument lists) runs too ------------------------------------------ */
void orc_k_vitter(int n, int64_t N, const uint32_t *words, int nblocks, int32_t *jn, int32_t *pos, int32_t *last);
void orc_k_mutate(int n, const double *best, const double *p, const uint32_t *words, const double *lb, const double *ub, double *out);
int orc_k_advance_slot(int n, int ld, const double *X, int64_t i0, int32_t jn, const int32_t *pos, int32_t last, const int64_t *W, int nun,
                       int t0, const double *lb, const double *ub, double *acc);

int nla_k_crs_vitter(int n, int64_t N, const uint32_t *words, int nblocks, int32_t *jn, int32_t *pos, int32_t *last, void *st)
{
    EMU_LAUNCH();
    (void) st;
    orc_k_vitter(n, N, words, nblocks, jn, pos, last);
    return 0;
}
int nla_k_crs_advance(int n, int ld, const double *X, int64_t i0, const int32_t *jn_ring, const int32_t *pos_ring, const int32_t *last_ring,
                      uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *W, int nW, const int32_t *t_in, int32_t *t_out,
                      int slot_mask, const double *lb, const double *ub, double *TX, int variant, void *st)
{
    EMU_LAUNCH();
    (void) variant; (void) st;
    for (int a = 0; a < K; ++a) {
        const uint64_t block = first_block + (uint64_t) a;
        const uint32_t rb = (uint32_t) (block % ring_blocks);
        const int q = (int) (block & (uint64_t) slot_mask);
        t_out[a] = orc_k_advance_slot(n, ld, X, i0, jn_ring[rb], pos_ring + (size_t) rb * (size_t) n, last_ring[rb], W, a < nW ? a : nW,
                                      t_in[a], lb, ub, TX + (size_t) q * (size_t) ld);
    }
    return 0;
}
int nla_k_crs_advance_args(int n, int ld, const double *X, int64_t i0, const int32_t *jn_ring, const int32_t *pos_ring,
                           const int32_t *last_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *h_W, int nW,
                           const int32_t *h_t_in, int32_t *t_out, int slot_mask, const double *lb, const double *ub, double *TX, int variant, void *st)
{
    EMU_LAUNCH();
    return nla_k_crs_advance(n, ld, X, i0, jn_ring, pos_ring, last_ring, ring_blocks, first_block, K, h_W, nW, h_t_in, t_out, slot_mask, lb, ub, TX, variant, st);
}
int orc_k_advance_slot_cols(int n, int ncol, int ld, const double *X, int64_t i0, int32_t jn, const int32_t *pos, int32_t last, const int64_t *W,
                            int nun, int t0, const double *lb, const double *ub, double *acc);
int nla_k_crs_advance_cols(int n, int ncol, int ld, const double *X, int64_t i0, const int32_t *jn_ring, const int32_t *pos_ring,
                           const int32_t *last_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *W, int nW, const int32_t *t_in,
                           int32_t *t_out, int slot_mask, const double *lb, const double *ub, double *TX, int variant, void *st)
{
    EMU_LAUNCH();
    (void) variant; (void) st;
    for (int a = 0; a < K; ++a) {
        const uint64_t block = first_block + (uint64_t) a;
        const uint32_t rb = (uint32_t) (block % ring_blocks);
        const int q = (int) (block & (uint64_t) slot_mask);
        t_out[a] = orc_k_advance_slot_cols(n, ncol, ld, X, i0, jn_ring[rb], pos_ring + (size_t) rb * (size_t) n, last_ring[rb], W, a < nW ? a : nW,
                                           t_in[a], lb, ub, TX + (size_t) q * (size_t) ld);
    }
    return 0;
}

/* ---- the column-sharded CRS2_LM (hip/crs_shard.hip): slices of rows, the pass's candidates packed, gathered and evaluated ------------ */
int nla_k_crs_sh_init_rows(int n, int c0, int nc, int ld, const double *lb, const double *ub, const uint32_t *words, int64_t row_first,
                           int64_t nrows, double *X, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int64_t r = 0; r < nrows; ++r) {
        const uint32_t *w = words + (size_t) r * 2 * (size_t) n + 2 * (size_t) c0;
        double *xr = X + (size_t) (row_first + r) * (size_t) ld;
        for (int i = 0; i < ld; ++i) xr[i] = i < nc ? urand_from(lb[i], ub[i], w[2 * i], w[2 * i + 1]) : 0.0;
    }
    return 0;
}
int nla_k_crs_sh_mutate_pack(int n, int c0, int nc, int ld, int colper, const double *X, int64_t i0, const double *TX, double *TM,
                             const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *t_in,
                             const int32_t *t_out, int slot_mask, const double *lb, const double *ub, double *SEND, int flag_forced, int flag_timed,
                             void *st)
{
    EMU_LAUNCH();
    (void) st;
    if (nc > colper || nc > ld) return EMU_ERR;
    SEND[(size_t) 2 * K * colper] = (double) flag_forced; SEND[(size_t) 2 * K * colper + 1] = flag_timed ? 1. : 0.;
    for (int a = 0; a < K; ++a) {
        const uint64_t block = first_block + (uint64_t) a;
        const int q = (int) (block & (uint64_t) slot_mask);
        const double *x = TX + (size_t) q * (size_t) ld, *xb = X + (size_t) i0 * (size_t) ld;
        const uint32_t *w = words_ring + (size_t) ((block + 1) % ring_blocks) * 2 * (size_t) n + 2 * (size_t) c0;
        double *m = TM + (size_t) q * (size_t) ld, *sT = SEND + (size_t) (2 * a) * (size_t) colper, *sM = sT + colper;
        if (!(t_out[a] == n && t_in[a] != n)) continue;
        for (int i = 0; i < ld; ++i) {                                         /* crs.c:140-145 on the slice */
            double v = 0;
            if (i < nc) {
                const double wv = urand_from(0., 1., w[2 * i], w[2 * i + 1]);
                v = xb[i] * (1 + wv) - wv * x[i];
                if (v > ub[i]) v = ub[i]; else if (v < lb[i]) v = lb[i];
                sT[i] = x[i]; sM[i] = v;
            }
            m[i] = v;
        }
    }
    return 0;
}
int nla_k_crs_sh_eval(int obj, int n, int colper, uint64_t first_block, int K, const int32_t *t_in, const int32_t *t_out, int slot_mask,
                      const double *RECV, int world, double *fT_ring, double *fM_ring, nla_crs_slot_status *status, void *st)
{
    EMU_LAUNCH();
    const double sign = emu_obj_sign(&obj);
    double *p;
    (void) st;
    if (obj < 0 || obj >= NLA_OBJ_COUNT || colper < 1) return EMU_ERR;
    p = (double *) malloc(sizeof(double) * (size_t) (n > 0 ? n : 1));
    for (int a = 0; a < K; ++a) {
        const int q = (int) ((first_block + (uint64_t) a) & (uint64_t) slot_mask);
        const int t1 = t_out[a];
        double f[2] = { 0, 0 };
        if (t1 == n && t_in[a] != n) {
            for (int task = 0; task < 2; ++task) {
                for (int g = 0; g < n; ++g)
                    p[g] = RECV[(size_t) (g / colper) * ((size_t) 2 * (size_t) K * (size_t) colper + 2) + (size_t) (2 * a + task) * (size_t) colper + (size_t) (g % colper)];
                f[task] = sign * nla_obj_eval_seq(obj, (unsigned) n, p, NULL);           /* crs.c:133 / :146 on the assembled point */
            }
            fT_ring[q] = f[0]; fM_ring[q] = f[1];
        } else if (t1 == n) { f[0] = fT_ring[q]; f[1] = fM_ring[q]; }
        status[a].fT = f[0]; status[a].fM = f[1]; status[a].t = t1; status[a].pad = 0;
    }
    {
        const size_t rs = (size_t) 2 * (size_t) K * (size_t) colper + 2;
        double f0 = 0, f1 = 0;
        int failed = 0;
        for (int r = 0; r < world; ++r) {
            if (RECV[(size_t) r * rs + rs - 2] != 0.) f0 = 1.;
            if (RECV[(size_t) r * rs + rs - 2] == 2.) failed = 1;
            if (RECV[(size_t) r * rs + rs - 1] != 0.) f1 = 1.;
        }
        status[K].fT = f0; status[K].fM = f1; status[K].t = failed; status[K].pad = 0;
    }
    free(p);
    return 0;
}

void orc_k_crs_chain(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring, const int32_t *pos_ring,
                     const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K,
                     const int64_t *W, const double *Wf, int nW, int slot_mask, const double *lb, const double *ub, double *TX, double *TM,
                     orc_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, const orc_slot_status *decide_with, uint32_t *dbg);
size_t nla_crs_chain_ctrl_bytes(int K, int nW) { (void) K; (void) nW; return 64; }
int nla_crs_chain_chunks(int n, int ld) { (void) ld; return (n + 63) / 64; }
int nla_k_crs_chain(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring, const int32_t *pos_ring,
                    const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *W,
                    const double *Wf, int nW, int w_on_host, int slot_mask, const double *lb, const double *ub, double *TX, double *TM, void *ctrl,
                    uint32_t ticket_base, nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, void *st)
{
    EMU_LAUNCH();
    (void) w_on_host; (void) ctrl; (void) ticket_base; (void) st;
    if (K > 256 || nW > 256 || obj < 0) return EMU_ERR;
    orc_k_crs_chain(obj, n, ld, X, i0, f_best, jn_ring, pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, slot_mask, lb, ub,
                    TX, TM, (orc_slot_status *) status, fwcnt, fwrec, fwcap, NULL, NULL);
    return 0;
}
int nla_k_crs_chain_lean(int obj, int n, int ld, const double *X, int64_t i0, double f_best, const int32_t *jn_ring, const int32_t *pos_ring,
                         const int32_t *last_ring, const uint32_t *words_ring, uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *W,
                         const double *Wf, int nW, int w_on_host, int slot_mask, const double *lb, const double *ub, double *TX, double *TM, void *ctrl,
                         uint32_t ticket_base, nla_crs_slot_status *status, uint32_t *fwcnt, uint32_t *fwrec, int fwcap, int ctrl_is_zero, void *st)
{
    (void) ctrl_is_zero;
    return nla_k_crs_chain(obj, n, ld, X, i0, f_best, jn_ring, pos_ring, last_ring, words_ring, ring_blocks, first_block, K, W, Wf, nW, w_on_host,
                           slot_mask, lb, ub, TX, TM, ctrl, ticket_base, status, fwcnt, fwrec, fwcap, st);
}
uint32_t nla_crs_chain_tickets(int n, int ld, int K) { return (uint32_t) nla_crs_chain_chunks(n, ld) * (uint32_t) K + 1u; }
int nla_k_crs_finish(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM, const uint32_t *words_ring,
                     uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *t_in, const int32_t *t_out, int slot_mask,
                     const double *lb, const double *ub, double *fT_ring, double *fM_ring, nla_crs_slot_status *status, void *st)
{
    EMU_LAUNCH();
    const double sign = emu_obj_sign(&obj);
    (void) st;
    for (int a = 0; a < K; ++a) {
        const uint64_t block = first_block + (uint64_t) a;
        const int q = (int) (block & (uint64_t) slot_mask);
        const int t1 = t_out[a], newly = (t1 == n) && t_in[a] != n;
        const double *x = TX + (size_t) q * (size_t) ld;
        double fT = 0, fM = 0;
        if (obj >= 0) {
            if (newly) {
                double *m = TM + (size_t) q * (size_t) ld;
                fT = fT_ring[q] = sign * nla_obj_eval_seq(obj, (unsigned) n, x, NULL);                /* crs.c:133 */
                orc_k_mutate(n, X + (size_t) i0 * (size_t) ld, x, words_ring + (size_t) ((block + 1) % ring_blocks) * 2 * (size_t) n, lb, ub, m);
                fM = fM_ring[q] = sign * nla_obj_eval_seq(obj, (unsigned) n, m, NULL);                /* crs.c:139-146 */
            } else if (t1 == n) { fT = fT_ring[q]; fM = fM_ring[q]; }
        }
        status[a].fT = fT; status[a].fM = fM; status[a].t = t1; status[a].pad = 0;
    }
    return 0;
}
int nla_k_crs_finish_args(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM, const uint32_t *words_ring,
                          uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *h_t_in, const int32_t *t_out, int slot_mask,
                          const double *lb, const double *ub, double *fT_ring, double *fM_ring, nla_crs_slot_status *status, void *st)
{
    EMU_LAUNCH();
    return nla_k_crs_finish(obj, n, ld, X, i0, TX, TM, words_ring, ring_blocks, first_block, K, h_t_in, t_out, slot_mask, lb, ub, fT_ring, fM_ring, status, st);
}
int nla_k_crs_finish_args_bell(int obj, int n, int ld, const double *X, int64_t i0, const double *TX, double *TM, const uint32_t *words_ring,
                               uint32_t ring_blocks, uint64_t first_block, int K, const int32_t *h_t_in, const int32_t *t_out, int slot_mask,
                               const double *lb, const double *ub, double *fT_ring, double *fM_ring, nla_crs_slot_status *status,
                               uint32_t *bell_count, uint32_t *bell, uint32_t bell_seq, void *st)
{
    const int rc = nla_k_crs_finish_args(obj, n, ld, X, i0, TX, TM, words_ring, ring_blocks, first_block, K, h_t_in, t_out, slot_mask, lb, ub,
                                         fT_ring, fM_ring, status, st);
    (void) bell_count;
    if (!rc) __atomic_store_n(bell, bell_seq, __ATOMIC_RELEASE);
    return rc;
}
int nla_k_crs_commit(int n, int ld, double *X, const double *TX, const double *TM, int ncommit, const int32_t *slot, const int32_t *kind,
                     const int64_t *row, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int c = 0; c < ncommit; ++c)                                                                  /* crs.c:153 */
        memmove(X + (size_t) row[c] * (size_t) ld, (kind[c] == 1 ? TX : TM) + (size_t) slot[c] * (size_t) ld, sizeof(double) * (size_t) n);
    return 0;
}
int nla_k_crs_commit_zero(int n, int ld, double *X, const double *TX, const double *TM, int ncommit, const int32_t *slot, const int32_t *kind,
                          const int64_t *row, int lists_on_host, void *zero, size_t zero_bytes, void *st)
{
    if (ncommit <= 0 || (zero_bytes & 3u) || (lists_on_host && ncommit > 128)) return EMU_ERR;
    if (zero_bytes) memset(zero, 0, zero_bytes);
    return nla_k_crs_commit(n, ld, X, TX, TM, ncommit, slot, kind, row, st);
}
int nla_k_crs_commit_args(int n, int ld, double *X, const double *TX, const double *TM, int ncommit, const int32_t *h_slot, const int32_t *h_kind,
                          const int64_t *h_row, void *st)
{
    EMU_LAUNCH();
    return nla_k_crs_commit(n, ld, X, TX, TM, ncommit, h_slot, h_kind, h_row, st);
}
int nla_k_crs_advance_commit_args(int n, int ld, double *X, int64_t i0, const int32_t *jn_ring, const int32_t *pos_ring, const int32_t *last_ring,
                                  uint32_t ring_blocks, uint64_t first_block, int K, const int64_t *h_W, int nW, const int32_t *h_t_in, int32_t *t_out,
                                  int slot_mask, const double *lb, const double *ub, double *TX, const double *TM, int ncommit,
                                  const int32_t *h_slot, const int32_t *h_kind, const int64_t *h_row, int variant, void *st)
{
    /* (one launch on the device, with the rows' reads forwarded; here simply one after the other: the same values) */
    int rc = nla_k_crs_commit_args(n, ld, X, TX, TM, ncommit, h_slot, h_kind, h_row, st);
    if (rc) return rc;
    return nla_k_crs_advance_args(n, ld, X, i0, jn_ring, pos_ring, last_ring, ring_blocks, first_block, K, h_W, nW, h_t_in, t_out, slot_mask, lb, ub,
                                  TX, variant, st);
}
int nla_k_crs_mutate(int n, const double *best, double *p, const uint32_t *words, const double *lb, const double *ub, void *st)
{
    EMU_LAUNCH();
    (void) st;
    orc_k_mutate(n, best, p, words, lb, ub, p);
    return 0;
}

/* ---- ESCH (hip/esch_kernels.hip) — the serial loops of esch.c behind the launchers' contracts ------------------------------------ */
static int esch_attempt(uint32_t w0, uint32_t w1, double *v01)            /* one attempt of randcauchy (esch.c:28-50), folded to [0,1] */
{
    const double u = urand_from(0., 1., w0, w1);
    const double c = 1.0 * tan((u - 0.5) * 3.14159265358979323846) + 0.0;
    double f;
    if ((c < 0.0 - (10.0 * 0.5)) || (c > 0.0 + (10.0 * 0.5))) return 0;
    f = (c < 0) ? -c : c + (10.0 * 0.5);
    *v01 = f / 10.0;
    return 1;
}
int nla_k_esch_cauchy(const uint32_t *words, int64_t nattempts, int64_t attempt_base, int32_t *counts, int64_t *vtotal, int64_t vbase,
                      int64_t vcap, double *v, int64_t *vatt, void *st)
{
    EMU_LAUNCH();
    int64_t cnt = 0;
    (void) counts; (void) st;
    for (int64_t a = 0; a < nattempts; ++a) {
        double val;
        if (!esch_attempt(words[2 * a], words[2 * a + 1], &val)) continue;
        if (vbase + cnt < vcap) { v[vbase + cnt] = val; vatt[vbase + cnt] = attempt_base + a; }
        ++cnt;
    }
    *vtotal += cnt;
    return 0;
}
int nla_k_esch_fill_rows(int n, int ld, const double *lb, const double *ub, const double *v, int64_t e0, int64_t count, double *R, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int64_t e = e0; e < e0 + count; ++e) {
        const int64_t id = e / n;
        const int item = (int) (e - id * n);
        R[(size_t) id * ld + item] = lb[item] + (ub[item] - lb[item]) * v[e - e0];
    }
    return 0;
}
int nla_k_esch_crossover(int n, int ld, int64_t np, int64_t no, const uint32_t *words, const int32_t *slot, double *R, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int64_t id = 0; id < no; ++id) {                                  /* esch.c:192-203 */
        const uint32_t *w = words + 3 * id;
        const int64_t p1 = (int64_t) (w[0] % (uint32_t) np), p2 = (int64_t) (w[1] % (uint32_t) np);
        const int cross = (int) (w[2] % (uint32_t) n);
        const double *a = R + (size_t) slot[p1] * ld, *b = R + (size_t) slot[p2] * ld;
        double *o = R + (size_t) slot[np + id] * ld;
        for (int j = 0; j < n; ++j) o[j] = j < cross ? a[j] : b[j];
    }
    return 0;
}
size_t nla_esch_mut_scratch_bytes(int64_t M) { (void) M; return 16; }
int nla_k_esch_mutate(const uint32_t *W, int64_t M, int64_t total, int n, int ld, int64_t np, int64_t no, const double *lb, const double *ub,
                      const int32_t *slot, double *R, int32_t *last, void *scratch, int64_t *out, void *st)
{
    EMU_LAUNCH();
    int64_t p = 0, c = 0;
    (void) last; (void) scratch; (void) st;
    out[0] = 0; out[1] = 0;
    while (c < total) {                                                    /* esch.c:207-218: iurand(no), iurand(n), randcauchy */
        int64_t q = p + 2;
        double v = 0;
        int ok = 0;
        if (p + 1 >= M) break;
        for (; q + 1 < M; q += 2) if (esch_attempt(W[q], W[q + 1], &v)) { ok = 1; break; }
        if (!ok) break;                                                    /* the segment ends inside this step: the caller retries longer */
        {
            const int64_t io = (int64_t) (W[p] % (uint32_t) no);
            const int ip = (int) (W[p + 1] % (uint32_t) n);
            R[(size_t) slot[np + io] * ld + ip] = lb[ip] + (ub[ip] - lb[ip]) * v;
        }
        p = q + 2;
        ++c;
    }
    out[0] = c; out[1] = p;
    return 0;
}
int nla_k_esch_gather_rows(int n, int ld, const int32_t *slot, int64_t i0, int64_t count, const double *R, double *G, void *st)
{
    EMU_LAUNCH();
    (void) st;
    for (int64_t i = 0; i < count; ++i) memmove(G + (size_t) i * ld, R + (size_t) slot[i0 + i] * ld, sizeof(double) * (size_t) n);
    return 0;
}
size_t nla_esch_sort_scratch_bytes(int64_t count) { (void) count; return 16; }
int nla_k_esch_select(int64_t count, const int32_t *slot_in, const double *fit_in, int32_t *slot_out, double *fit_out, void *scratch,
                      size_t scratch_bytes, void *st)
{
    EMU_LAUNCH();
    /* stable ascending order by fitness (esch.c:243: nlopt_qsort_r = glibc's merge sort): insertion into the sorted prefix after
     * every element that is not greater */
    (void) scratch; (void) scratch_bytes; (void) st;
    for (int64_t i = 0; i < count; ++i) {
        int64_t k = i;
        while (k > 0 && fit_out[k - 1] > fit_in[i]) { fit_out[k] = fit_out[k - 1]; slot_out[k] = slot_out[k - 1]; --k; }
        fit_out[k] = fit_in[i]; slot_out[k] = slot_in[i];
    }
    return 0;
}
