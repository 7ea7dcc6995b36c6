/* comm.c — the collective layer of the multi-GPU runs: one process per GPU, every rank runs the same
 * deterministic host driver on the same MT19937 stream and the data-parallel device work of a step
 * is partitioned over the ranks; the only exchange the algorithms need is an ALL-GATHER (SURVEY.md
 * §8e: f / penalty of the candidates, rows of new population members, local minima).
 *
 * Three transports behind one interface:
 *   RCCL   ncclAllGather on device buffers over xGMI.  librccl is dlopen()ed on first use so that
 *          single-GPU programs never load or initialise it; the 128-byte unique id is created on
 *          rank 0 (nlopt_amd_rccl_unique_id) and handed to every rank by the launcher.
 *   host   a user-supplied all-gather on host buffers (MPI, gloo, ...); device data are staged
 *          through pinned memory.  This is what the world_size-2 tests use.
 *   shm    ranks on ONE node without a collective library: a POSIX shared-memory segment holds two sets of per-rank slots and a
 *          barrier; an all-gather is "copy in, one barrier, copy out" (the sets alternate, so the next call's writes cannot meet
 *          this call's reads).  The segment is registered with the device runtime, so device data go D2H straight into the rank's
 *          slot and H2D straight out of the slot set (rank-major = the receive layout) — no second staging copy, no Python in the
 *          exchange: what tools/shard_probe.py times a sharded pass over, and a fallback where RCCL cannot be used.
 *
 * The reference has no counterpart (it is single-threaded).
 */
#include "nla_internal.h"
#include "nla_switches.h"
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { char internal[128]; } rccl_uid;              /* ncclUniqueId, NCCL_UNIQUE_ID_BYTES = 128 */
typedef int (*fn_get_uid)(rccl_uid *);
typedef int (*fn_init_rank)(void **comm, int nranks, rccl_uid id, int rank);
typedef int (*fn_allgather)(const void *send, void *recv, size_t count, int dtype, void *comm, void *stream);
typedef int (*fn_destroy)(void *comm);
typedef const char *(*fn_errstr)(int);
#define RCCL_UINT8 1                                          /* ncclUint8 */

static struct {
    void *dl;
    fn_get_uid get_uid; fn_init_rank init_rank; fn_allgather allgather; fn_destroy destroy; fn_errstr errstr;
} R;

static pthread_once_t rccl_once = PTHREAD_ONCE_INIT;
static void rccl_load_once(void)
{
    /* NLA_RCCL_LIBRARY=<path>: the collective library to bind instead of the system's librccl (a site's own RCCL build; the tests
     * point it at a mock that checks the all-gather contract, so that this transport runs with several ranks without GPUs) */
    const char *path = NLA_DBG_ENV("NLA_RCCL_LIBRARY");
    R.dl = path && *path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!R.dl && !(path && *path)) R.dl = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!R.dl && !(path && *path)) R.dl = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!R.dl) return;
    R.get_uid = (fn_get_uid) dlsym(R.dl, "ncclGetUniqueId");
    R.init_rank = (fn_init_rank) dlsym(R.dl, "ncclCommInitRank");
    R.allgather = (fn_allgather) dlsym(R.dl, "ncclAllGather");
    R.destroy = (fn_destroy) dlsym(R.dl, "ncclCommDestroy");
    R.errstr = (fn_errstr) dlsym(R.dl, "ncclGetErrorString");
    if (!R.get_uid || !R.init_rank || !R.allgather || !R.destroy) { dlclose(R.dl); R.dl = NULL; }
}
static int rccl_load(void)                                    /* any thread may be the first to ask */
{
    pthread_once(&rccl_once, rccl_load_once);
    return R.dl ? 0 : -1;
}

/* the shared segment: header, then 2 sets x world slots of `slot` bytes */
typedef struct {
    _Atomic uint32_t magic;             /* set last by rank 0: the segment is initialised */
    uint32_t world;
    uint64_t slot;
    _Atomic uint32_t arrived, generation;
    _Atomic uint32_t attached, detached;
    _Atomic uint32_t aborted;           /* nla_comm_abort: a rank gave up in the middle of a job; every barrier from now on fails */
    /* who made the segment: rank 0's pid and the kernel's start time of that process (/proc/<pid>/stat field 22).  A segment left under the
     * same name by a run that crashed carries a valid magic, world and slot size and garbage barrier words; what tells it from this job's
     * is that its creator no longer exists — a rank > 0 attaches only to a segment whose creator is alive (shm_creator_alive) */
    uint32_t creator_pid;
    uint64_t creator_start;
    char pad[64 - 56];
} shm_header;
#define SHM_MAGIC 0x6e6c6173u
typedef struct {
    shm_header *h; char *slots; size_t bytes_mapped, slot;
    int owner, registered; unsigned parity;
    double barrier_timeout_s;           /* nlopt_amd_comm_set_timeout; 600 s unless set */
    char name[96];
} shm_state;
#if defined(__x86_64__) || defined(__i386__)
#define NLA_CPU_RELAX() __builtin_ia32_pause()
#elif defined(__aarch64__)
#define NLA_CPU_RELAX() __asm__ __volatile__("yield" ::: "memory")
#else
#define NLA_CPU_RELAX() ((void) 0)
#endif

struct nlopt_amd_comm_s {
    int rank, world;
    void *rccl;                         /* ncclComm_t, or NULL: host / shm transport */
    shm_state *shm;                     /* shm transport */
    nlopt_amd_allgather_fn fn; void *ctx;
    void *h_send, *h_recv; size_t h_cap;          /* pinned staging (host transport / host data over RCCL) */
    void *d_send, *d_recv; size_t d_cap;          /* device staging for host data over RCCL */
    uint64_t calls, bytes;
    char err[160];
};

int nlopt_amd_rccl_unique_id(void *id128)
{
    rccl_uid id;
    if (!id128 || rccl_load()) return -1;
    if (R.get_uid(&id)) return -2;
    memcpy(id128, &id, sizeof id);
    return 0;
}

static int need_dev(nlopt_amd_comm *c, size_t bytes);

nlopt_amd_comm *nlopt_amd_comm_create_rccl(int rank, int world, const void *id128)
{
    nlopt_amd_comm *c;
    rccl_uid id;
    if (world < 1 || rank < 0 || rank >= world || !id128 || rccl_load()) return NULL;
    c = (nlopt_amd_comm *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->rank = rank; c->world = world;
    memcpy(&id, id128, sizeof id);
    if (R.init_rank(&c->rccl, world, id, rank) || !c->rccl) { free(c); return NULL; }
    if (need_dev(c, 64)) c->err[0] = 0;         /* staging for the small exchanges (ready / stop agreement) now, while there is memory: a rank
                                                 * that later runs out must still be able to say so (nla_comm_agree_ready); grown on demand */
    return c;
}

nlopt_amd_comm *nlopt_amd_comm_create_host(int rank, int world, nlopt_amd_allgather_fn fn, void *ctx)
{
    nlopt_amd_comm *c;
    if (world < 1 || rank < 0 || rank >= world || (!fn && world > 1)) return NULL;
    c = (nlopt_amd_comm *) calloc(1, sizeof *c);
    if (!c) return NULL;
    c->rank = rank; c->world = world; c->fn = fn; c->ctx = ctx;
    return c;
}

/* ---- shm transport ------------------------------------------------------------------------------------------------------ */
static int shm_barrier(nlopt_amd_comm *c)
{
    shm_header *h = c->shm->h;
    const uint32_t g = atomic_load_explicit(&h->generation, memory_order_acquire);
    unsigned spins = 0;
    struct timespec t0, t1;
    if (atomic_load_explicit(&h->aborted, memory_order_acquire)) { snprintf(c->err, sizeof c->err, "shm transport: a rank aborted the job"); return -1; }
    if (atomic_fetch_add_explicit(&h->arrived, 1, memory_order_acq_rel) + 1 == (uint32_t) c->world) {
        atomic_store_explicit(&h->arrived, 0, memory_order_relaxed);
        atomic_store_explicit(&h->generation, g + 1, memory_order_release);
        return 0;
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (atomic_load_explicit(&h->generation, memory_order_acquire) == g) {
        if (++spins < 2000) { NLA_CPU_RELAX(); continue; }
        if (atomic_load_explicit(&h->aborted, memory_order_acquire)) { snprintf(c->err, sizeof c->err, "shm transport: a rank aborted the job"); return -1; }
        sched_yield();
        if ((spins & 1023) == 0) {                       /* a rank that died must not hang the others for ever */
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if ((double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec) > c->shm->barrier_timeout_s) {
                snprintf(c->err, sizeof c->err, "shm transport: a rank did not reach the barrier within %g s", c->shm->barrier_timeout_s);
                return -1;
            }
        }
    }
    return 0;
}
/* host buffers: every rank's `bytes` into its slot of the current set, one barrier, the set out in rank order; payloads larger than a
 * slot go in pieces */
static int shm_allgather(void *ctx, const void *send, void *recv, size_t bytes)
{
    nlopt_amd_comm *c = (nlopt_amd_comm *) ctx;
    shm_state *s = c->shm;
    size_t off = 0;
    do {
        const size_t part = bytes - off < s->slot ? bytes - off : s->slot;
        char *set = s->slots + (size_t) (s->parity & 1) * (size_t) c->world * s->slot;
        int r;
        if (part) memcpy(set + (size_t) c->rank * s->slot, (const char *) send + off, part);
        if (shm_barrier(c)) return -1;
        for (r = 0; r < c->world; ++r) if (part) memcpy((char *) recv + (size_t) r * bytes + off, set + (size_t) r * s->slot, part);
        ++s->parity;
        off += part;
    } while (off < bytes);
    return 0;
}
static void shm_close(shm_state *s)
{
    if (!s) return;
    if (s->h) {
        if (s->registered) nla_host_unregister(s->h);
        /* the last rank to leave removes the name (rank 0 alone could pull it away under a rank that has not opened it yet) */
        if (atomic_fetch_add_explicit(&s->h->detached, 1, memory_order_acq_rel) + 1 == s->h->world) shm_unlink(s->name);
        munmap(s->h, s->bytes_mapped);
    }
    free(s);
}
/* start time of process `pid` in clock ticks since boot (field 22 of /proc/<pid>/stat; the command name in field 2 may hold spaces and
 * parentheses, so the fields are counted from the LAST ')'); 0 if the process does not exist */
static uint64_t proc_start_time(uint32_t pid)
{
    char path[64], buf[1024], *p;
    FILE *f;
    size_t len;
    int field;
    snprintf(path, sizeof path, "/proc/%u/stat", (unsigned) pid);
    f = fopen(path, "r");
    if (!f) return 0;
    len = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[len] = 0;
    p = strrchr(buf, ')');
    if (!p) return 0;
    for (field = 2, ++p; *p && field < 22; ++p) if (*p == ' ' && p[1] != ' ') ++field;      /* p ends on the first digit of field 22 */
    return *p ? (uint64_t) strtoull(p, NULL, 10) : 0;
}
static int shm_creator_alive(const shm_header *h)
{
    const uint64_t t = h->creator_pid ? proc_start_time(h->creator_pid) : 0;
    return t != 0 && t == h->creator_start;
}

/* `name`: a POSIX shared-memory name ("/something") every rank of the job passes identically and no other job uses; slot_bytes: the
 * largest single contribution moved in one piece (larger ones are split; 0 = 16 MiB).  All ranks must live on one node (and see the same
 * /proc: one pid namespace).
 * Start-up, safe against a segment that a crashed run left under the same name and against any order in which the ranks arrive:
 * rank 0 removes the old name, builds the new segment under a private name ("<name>.<pid>"), initialises it completely — sizes, barrier
 * words, its own pid + start time, the magic — and only then gives it the job's name (link(2) on /dev/shm: the name appears atomically
 * and never points at a half-made segment).  A rank > 0 opens the name, maps it and attaches only if magic / world / slot size agree AND
 * the creator recorded in the header is a live process; otherwise (the crashed run's segment: its creator is gone) it lets go and opens
 * the name again, for up to 120 s. */
nlopt_amd_comm *nlopt_amd_comm_create_shm(int rank, int world, const char *name, size_t slot_bytes)
{
    nlopt_amd_comm *c;
    shm_state *s;
    size_t total;
    int fd = -1, tries;
    if (world < 1 || rank < 0 || rank >= world || !name || name[0] != '/' || strchr(name + 1, '/') || strlen(name) + 24 >= sizeof s->name) return NULL;
    if (!slot_bytes) slot_bytes = (size_t) 16 << 20;
    slot_bytes = (slot_bytes + 4095) & ~(size_t) 4095;
    total = sizeof(shm_header) + 2 * (size_t) world * slot_bytes;
    total = (total + 4095) & ~(size_t) 4095;
    c = (nlopt_amd_comm *) calloc(1, sizeof *c);
    s = (shm_state *) calloc(1, sizeof *s);
    if (!c || !s) { free(c); free(s); return NULL; }
    c->rank = rank; c->world = world; c->shm = s; c->fn = shm_allgather; c->ctx = c;
    snprintf(s->name, sizeof s->name, "%s", name);
    s->slot = slot_bytes; s->bytes_mapped = total; s->owner = rank == 0; s->barrier_timeout_s = 600.;
    if (rank == 0) {
        char tmp[sizeof s->name], from[sizeof s->name + 16], to[sizeof s->name + 16];
        shm_header *h;
        snprintf(tmp, sizeof tmp, "%s.%ld", name, (long) getpid());
        shm_unlink(tmp);
        shm_unlink(name);                                /* a stale segment of a run that crashed */
        fd = shm_open(tmp, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t) total)) { if (fd >= 0) { close(fd); shm_unlink(tmp); } free(c); free(s); return NULL; }
        h = (shm_header *) mmap(NULL, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (h == MAP_FAILED) { shm_unlink(tmp); free(c); free(s); return NULL; }
        h->world = (uint32_t) world; h->slot = slot_bytes;
        h->creator_pid = (uint32_t) getpid(); h->creator_start = proc_start_time((uint32_t) getpid());
        atomic_store_explicit(&h->arrived, 0, memory_order_relaxed);
        atomic_store_explicit(&h->generation, 0, memory_order_relaxed);
        atomic_store_explicit(&h->attached, 0, memory_order_relaxed);
        atomic_store_explicit(&h->detached, 0, memory_order_relaxed);
        atomic_store_explicit(&h->aborted, 0, memory_order_relaxed);
        atomic_store_explicit(&h->magic, SHM_MAGIC, memory_order_release);
        /* complete: now it gets the job's name (POSIX shm objects are files under /dev/shm on Linux) */
        snprintf(from, sizeof from, "/dev/shm%s", tmp);
        snprintf(to, sizeof to, "/dev/shm%s", name);
        if (!h->creator_start || link(from, to)) { munmap(h, total); shm_unlink(tmp); free(c); free(s); return NULL; }
        shm_unlink(tmp);
        s->h = h;
    } else {
        for (tries = 0; tries < 120000 && !s->h; ++tries) {
            struct stat sb;
            shm_header *h;
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) { usleep(1000); continue; }
            if (fstat(fd, &sb) || (size_t) sb.st_size < sizeof(shm_header)) { close(fd); usleep(1000); continue; }
            h = (shm_header *) mmap(NULL, (size_t) sb.st_size < total ? (size_t) sb.st_size : total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (h == MAP_FAILED) { usleep(1000); continue; }
            if (atomic_load_explicit(&h->magic, memory_order_acquire) == SHM_MAGIC && shm_creator_alive(h)) {
                /* this job's segment (its creator lives): ranks that disagree about the job's shape are an error, not a reason to wait */
                if (h->world != (uint32_t) world || h->slot != slot_bytes || (size_t) sb.st_size < total) { munmap(h, (size_t) sb.st_size < total ? (size_t) sb.st_size : total); break; }
                s->h = h;
                break;
            }
            munmap(h, (size_t) sb.st_size < total ? (size_t) sb.st_size : total);      /* a crashed run's segment: rank 0 will replace it */
            usleep(1000);
        }
        if (!s->h) { free(c); free(s); return NULL; }
    }
    s->slots = (char *) s->h + sizeof(shm_header);
    atomic_fetch_add_explicit(&s->h->attached, 1, memory_order_acq_rel);
    /* device copies straight into / out of the slots (no device: the transport still serves host data) */
    s->registered = nla_dev_count() > 0 && nla_host_register(s->h, total) == 0;
    if (shm_barrier(c)) { shm_close(s); free(c); return NULL; }                   /* everyone is attached before anyone may leave */
    return c;
}

/* how long a rank waits in a barrier of the shm transport for the others before it reports an error (default 600 s; the other transports
 * have their own timeouts) */
void nlopt_amd_comm_set_timeout(nlopt_amd_comm *c, double seconds)
{
    if (c && c->shm && seconds > 0) c->shm->barrier_timeout_s = seconds;
}

void nlopt_amd_comm_destroy(nlopt_amd_comm *c)
{
    if (!c) return;
    if (c->rccl) R.destroy(c->rccl);
    shm_close(c->shm);
    nla_host_free(c->h_send); nla_host_free(c->h_recv);
    nla_dev_free(c->d_send); nla_dev_free(c->d_recv);
    free(c);
}

/* A rank that fails in the middle of a multi-rank job where it cannot say so through the job's own exchange (the device is gone, an
 * allocation failed between two collectives) calls this before it leaves: on the shm transport every rank waiting in, or arriving at, a
 * barrier returns an error instead of waiting for it; the RCCL and callback transports have no such channel (their peers wait for the
 * transport's own timeout) — the per-pass error flag of the sharded CRS run (crs_engine.c) covers what can still be said in band. */
void nla_comm_abort(nlopt_amd_comm *c)
{
    if (c && c->shm && c->shm->h) atomic_store_explicit(&c->shm->h->aborted, 1, memory_order_release);
}

int nlopt_amd_comm_rank(const nlopt_amd_comm *c) { return c ? c->rank : 0; }
int nlopt_amd_comm_world(const nlopt_amd_comm *c) { return c ? c->world : 1; }
const char *nlopt_amd_comm_error(const nlopt_amd_comm *c) { return c ? c->err : ""; }
void nlopt_amd_comm_counters(const nlopt_amd_comm *c, uint64_t *calls, uint64_t *bytes)
{
    if (calls) *calls = c ? c->calls : 0;
    if (bytes) *bytes = c ? c->bytes : 0;
}

nlopt_result nlopt_amd_set_comm(nlopt_opt opt, nlopt_amd_comm *c)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    opt->comm = c;
    return NLOPT_SUCCESS;
}

nlopt_result nlopt_amd_set_progress(nlopt_opt opt, nlopt_amd_progress_fn fn, void *data)
{
    if (!opt) return NLOPT_INVALID_ARGS;
    opt->progress = fn; opt->progress_data = data;
    return NLOPT_SUCCESS;
}

static int need_host(nlopt_amd_comm *c, size_t bytes)
{
    if (bytes * (size_t) c->world <= c->h_cap) return 0;
    nla_host_free(c->h_send); nla_host_free(c->h_recv);
    c->h_cap = 2 * bytes * (size_t) c->world;
    c->h_send = nla_host_malloc(c->h_cap / (size_t) c->world);
    c->h_recv = nla_host_malloc(c->h_cap);
    if (!c->h_send || !c->h_recv) { c->h_cap = 0; snprintf(c->err, sizeof c->err, "out of pinned memory for the collective staging"); return -1; }
    return 0;
}
static int need_dev(nlopt_amd_comm *c, size_t bytes)
{
    if (bytes * (size_t) c->world <= c->d_cap) return 0;
    nla_dev_free(c->d_send); nla_dev_free(c->d_recv);
    c->d_cap = 2 * bytes * (size_t) c->world;
    c->d_send = nla_dev_malloc(c->d_cap / (size_t) c->world);
    c->d_recv = nla_dev_malloc(c->d_cap);
    if (!c->d_send || !c->d_recv) { c->d_cap = 0; snprintf(c->err, sizeof c->err, "out of device memory for the collective staging"); return -1; }
    return 0;
}

/* Set-up time: make sure an all-gather of `bytes` per rank will not have to allocate its staging later — an allocation that fails on
 * one rank in the middle of a job leaves the others in the collective it could not join (0 ok, -1 out of memory: say so in the
 * set-up's agreement) */
int nla_comm_reserve(nlopt_amd_comm *c, size_t bytes)
{
    if (!c || c->world <= 1 || c->rccl) return 0;
    if (c->shm && c->shm->registered && bytes <= c->shm->slot) return 0;
    return need_host(c, bytes);
}

/* device buffers: d_recv (world*bytes, rank-major) := all-gather of every rank's d_send (bytes).
 * RCCL: enqueued on `stream` (asynchronous).  Host transport: synchronises `stream`. */
int nla_comm_allgather_dev(nlopt_amd_comm *c, const void *d_send, void *d_recv, size_t bytes, void *stream)
{
    int rc;
    /* a 1-rank RCCL communicator still goes through ncclAllGather (that is how a 1-GPU box exercises the transport) */
    if (!c || (c->world == 1 && !c->rccl)) return (bytes && d_recv != d_send) ? nla_memcpy_d2d(d_recv, d_send, bytes, stream) : 0;
    ++c->calls; c->bytes += bytes * (size_t) c->world;
    if (c->rccl) {
        rc = R.allgather(d_send, d_recv, bytes, RCCL_UINT8, c->rccl, stream);
        if (rc) snprintf(c->err, sizeof c->err, "ncclAllGather failed: %s", R.errstr ? R.errstr(rc) : "?");
        return rc;
    }
    if (c->shm && c->shm->registered && bytes <= c->shm->slot && bytes > 0) {
        /* the slots are pinned, device-visible memory: D2H into this rank's slot, one barrier, H2D of every rank's slot */
        shm_state *s = c->shm;
        char *set = s->slots + (size_t) (s->parity & 1) * (size_t) c->world * s->slot;
        int r;
        if (nla_memcpy_d2h(set + (size_t) c->rank * s->slot, d_send, bytes, stream) || nla_stream_sync(stream)) { snprintf(c->err, sizeof c->err, "staging copy failed"); return -1; }
        if (shm_barrier(c)) return -1;
        for (r = 0; r < c->world; ++r)
            if (nla_memcpy_h2d((char *) d_recv + (size_t) r * bytes, set + (size_t) r * s->slot, bytes, stream)) { snprintf(c->err, sizeof c->err, "staging copy failed"); return -1; }
        ++s->parity;
        if (nla_stream_sync(stream)) { snprintf(c->err, sizeof c->err, "staging copy failed"); return -1; }
        return 0;
    }
    if (need_host(c, bytes)) return -1;
    if (nla_memcpy_d2h(c->h_send, d_send, bytes, stream) || nla_stream_sync(stream)) { snprintf(c->err, sizeof c->err, "staging copy failed"); return -1; }
    if ((rc = c->fn(c->ctx, c->h_send, c->h_recv, bytes))) { snprintf(c->err, sizeof c->err, "host all-gather callback failed (%d)", rc); return rc; }
    if (nla_memcpy_h2d(d_recv, c->h_recv, bytes * (size_t) c->world, stream) || nla_stream_sync(stream)) { snprintf(c->err, sizeof c->err, "staging copy failed"); return -1; }
    return 0;
}

/* host buffers (small control data); synchronous */
int nla_comm_allgather_host(nlopt_amd_comm *c, const void *h_send, void *h_recv, size_t bytes, void *stream)
{
    int rc;
    if (!c || (c->world == 1 && !c->rccl)) { memmove(h_recv, h_send, bytes); return 0; }
    ++c->calls; c->bytes += bytes * (size_t) c->world;
    if (!c->rccl) {
        if ((rc = c->fn(c->ctx, h_send, h_recv, bytes))) snprintf(c->err, sizeof c->err, "host all-gather callback failed (%d)", rc);
        return rc;
    }
    if (need_dev(c, bytes)) return -1;
    if (nla_memcpy_h2d(c->d_send, h_send, bytes, stream)) return -1;
    rc = R.allgather(c->d_send, c->d_recv, bytes, RCCL_UINT8, c->rccl, stream);
    if (rc) { snprintf(c->err, sizeof c->err, "ncclAllGather failed: %s", R.errstr ? R.errstr(rc) : "?"); return rc; }
    if (nla_memcpy_d2h(h_recv, c->d_recv, bytes * (size_t) c->world, stream) || nla_stream_sync(stream)) return -1;
    return 0;
}

/* block partition of `count` units: every rank owns `per` = ceil(count/world) slots (all-gather wants
 * equal contributions); rank r's real units are [first, first+mine). */
void nla_comm_partition(const nlopt_amd_comm *c, int64_t count, int64_t *per, int64_t *first, int64_t *mine)
{
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    const int64_t p = (count + world - 1) / world;
    int64_t f = p * rank, m;
    if (f > count) f = count;
    m = count - f < p ? count - f : p;
    if (per) *per = p;
    if (first) *first = f;
    if (mine) *mine = m;
}


/* Multi-rank runs: the wall clock and the force_stop flag belong to a process, but the ranks must leave a run at the same
 * point — one rank returning MAXTIME_REACHED while another enters the next all-gather would hang it (or they would return
 * different results).  So the two per-process stop conditions are decided COLLECTIVELY at the points where the ranks exchange
 * data anyway: every rank contributes what it sees now, the OR over all ranks is what every rank then acts on until the next
 * agreement.  Returns `stop` itself for a single process; otherwise `view` = *stop with the flag / clock replaced by the
 * agreed verdicts (NULL if the exchange failed). */
const nla_stopping *nla_comm_agree_stop(nlopt_amd_comm *c, const nla_stopping *stop, nla_stopping *view, int *force_store)
{
    int mine[2], *all, r, forced = 0, timed = 0;
    const int world = nlopt_amd_comm_world(c);
    if (world <= 1) return stop;
    mine[0] = nla_stop_forced(stop);
    mine[1] = nla_stop_time(stop);
    all = (int *) malloc(sizeof(int) * 2 * (size_t) world);
    if (!all || nla_comm_allgather_host(c, mine, all, sizeof mine, NULL)) { free(all); return NULL; }
    for (r = 0; r < world; ++r) { forced |= all[2 * r]; timed |= all[2 * r + 1]; }
    free(all);
    nla_stop_view(stop, forced, timed, view, force_store);
    return view;
}

/* Multi-rank runs: a rank whose set-up failed (out of device memory, no device visible) must not leave its peers waiting in the
 * run's first collective — they would wait for ever.  So set-up ends with one small exchange: every rank says whether it is ready,
 * and all of them go on only if all are.  Returns `ok` for a single process, otherwise 1 iff every rank reported ok (0 also when the
 * exchange itself failed). */
int nla_comm_agree_ready(nlopt_amd_comm *c, int ok) { return nla_comm_agree_same(c, ok, 0) > 0; }

/* The same exchange carrying a fingerprint of what this rank was asked to do (nla_problem_fingerprint): one job over several
 * ranks means the identical problem, stopping criteria and generator state on every rank — ranks seeded differently would take
 * different decisions and pass each other in the collectives (a hang, or worse, a result).  Returns 1 = all ready and the same,
 * 0 = some rank is not ready (or the exchange failed), -1 = all ready but the fingerprints differ. */
int nla_comm_agree_same(nlopt_amd_comm *c, int ok, uint64_t fingerprint)
{
    uint64_t stack[2 * 128], *all = stack, mine[2];
    int r, res = 1;
    const int world = nlopt_amd_comm_world(c);
    if (world <= 1) return ok ? 1 : 0;
    mine[0] = ok ? 1 : 0; mine[1] = fingerprint;
    if (world > 128 && !(all = (uint64_t *) malloc(sizeof mine * (size_t) world))) {
        /* out of memory here must not leave the other ranks waiting in the exchange: join it piecewise-free through the comm's own
         * staging (nla_comm_allgather_host allocates / reuses it) by reporting "not ready" from a stack slot and discarding the rest */
        static uint64_t sink[2 * 4096];
        mine[0] = 0;
        if (world <= 4096) (void) nla_comm_allgather_host(c, mine, sink, sizeof mine, NULL);
        return 0;
    }
    if (nla_comm_allgather_host(c, mine, all, sizeof mine, NULL)) res = 0;
    else {
        for (r = 0; r < world; ++r) if (!all[2 * r]) res = 0;
        for (r = 0; r < world && res == 1; ++r) if (all[2 * r + 1] != fingerprint) res = -1;
    }
    if (all != stack) free(all);
    return res;
}

/* what must be identical on every rank of one job: algorithm, dimension, population, objective, box, starting point, the stopping
 * criteria every rank tests on its own (NOT maxtime / force_stop: those are agreed collectively while the run goes on) and the
 * calling thread's MT19937 state (FNV-1a over the bytes) */
static uint64_t fnv(uint64_t h, const void *p, size_t bytes)
{
    const unsigned char *b = (const unsigned char *) p;
    while (bytes--) { h ^= *b++; h *= 1099511628211ULL; }
    return h;
}
uint64_t nla_problem_fingerprint(int algorithm, int n, int population, int obj, const double *lb, const double *ub, const double *x,
                                 const nla_stopping *stop)
{
    uint64_t h = 14695981039346656037ULL;
    uint32_t mt[NLA_MT_N];
    int head[5], pos = 0;
    head[0] = algorithm; head[1] = n; head[2] = population; head[3] = obj; head[4] = stop ? stop->maxeval : 0;
    h = fnv(h, head, sizeof head);
    if (lb) h = fnv(h, lb, sizeof(double) * (size_t) n);
    if (ub) h = fnv(h, ub, sizeof(double) * (size_t) n);
    if (x) h = fnv(h, x, sizeof(double) * (size_t) n);
    if (stop) {
        double t[4];
        t[0] = stop->minf_max; t[1] = stop->ftol_rel; t[2] = stop->ftol_abs; t[3] = stop->xtol_rel;
        h = fnv(h, t, sizeof t);
        if (stop->xtol_abs) h = fnv(h, stop->xtol_abs, sizeof(double) * (size_t) n);
        if (stop->x_weights) h = fnv(h, stop->x_weights, sizeof(double) * (size_t) n);
    }
    nla_mt_export(mt, &pos);
    h = fnv(h, mt, sizeof mt);
    h = fnv(h, &pos, sizeof pos);
    return h ? h : 1;
}

/* the options of `opt` that shape a run's control flow and the sizes of its collectives (every nlopt_set_param value: the window depth,
 * the sharding / summation / pipeline switches ...): ranks that differ in one of them issue different all-gathers.  Order-independent
 * (a sum of per-parameter hashes), 0 for no parameters; the callers fold it into nla_problem_fingerprint's value. */
uint64_t nla_params_fingerprint(const nlopt_opt opt)
{
    uint64_t sum = 0;
    unsigned i;
    if (!opt) return 0;
    for (i = 0; i < opt->nparams; ++i) {
        uint64_t h = 14695981039346656037ULL;
        h = fnv(h, opt->params[i].name, strlen(opt->params[i].name));
        h = fnv(h, &opt->params[i].val, sizeof(double));
        sum += h;
    }
    return sum;
}

/* *stop with its two per-process conditions replaced by agreed verdicts */
void nla_stop_view(const nla_stopping *stop, int forced, int timed, nla_stopping *view, int *force_store)
{
    *view = *stop;
    *force_store = forced;
    view->force_stop = force_store;
    if (timed) { view->maxtime = 1e-300; view->start = -1e300; }      /* nla_stop_time(view) is true from now on */
    else view->maxtime = 0;                                           /* ... or false until the next agreement */
}
