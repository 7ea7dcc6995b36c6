/* port_objs.c — CPU ORACLE (test infrastructure): objective-zoo callbacks for the CPU reference
 * and the port, compiled from the same objfuncs.h the product and the HIP kernels use
 * (SURVEY.md §7.1 step 1).  Plus a recording wrapper used to pin the port against the real
 * reference evaluation by evaluation (f value and a hash of x of every callback, in order). */
#include "port_oracle.h"
#include "objfuncs.h"
#include <string.h>

#define DEF(name, id) static double name(unsigned n, const double *x, double *g, void *d) { (void) d; return nla_obj_eval_seq(id, n, x, g); }
DEF(obj_rastrigin, NLA_OBJ_RASTRIGIN)
DEF(obj_ackley, NLA_OBJ_ACKLEY)
DEF(obj_griewank, NLA_OBJ_GRIEWANK)
DEF(obj_rosenbrock, NLA_OBJ_ROSENBROCK)
DEF(obj_levy, NLA_OBJ_LEVY)
DEF(obj_sphere, NLA_OBJ_SPHERE)

orc_func orc_objective(int id)
{
    switch (id) {
    case NLA_OBJ_RASTRIGIN: return obj_rastrigin;
    case NLA_OBJ_ACKLEY: return obj_ackley;
    case NLA_OBJ_GRIEWANK: return obj_griewank;
    case NLA_OBJ_ROSENBROCK: return obj_rosenbrock;
    case NLA_OBJ_LEVY: return obj_levy;
    case NLA_OBJ_SPHERE: return obj_sphere;
    default: return NULL;
    }
}

double orc_con_blocksum(unsigned n, const double *x, double *grad, void *data)
{
    const unsigned *qQ = (const unsigned *) data;
    return nla_con_blocksum_seq(n, x, grad, qQ[0], qQ[1]);
}

void orc_obj_box(int id, double *lo, double *hi) { nla_obj_box(id, lo, hi); }

static uint64_t hash_bits(const double *x, unsigned n)      /* FNV-1a over the raw bytes */
{
    uint64_t h = 1469598103934665603ULL;
    const unsigned char *b = (const unsigned char *) x;
    for (size_t i = 0; i < (size_t) n * sizeof(double); ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}
uint64_t orc_hash_doubles(const double *x, unsigned n) { return hash_bits(x, n); }

double orc_recording_callback(unsigned n, const double *x, double *grad, void *data)
{
    orc_recorder *r = (orc_recorder *) data;
    double f = r->inner(n, x, grad, r->inner_data);
    if (r->len < r->cap) {
        if (r->fbuf) r->fbuf[r->len] = f;
        if (r->xhash) r->xhash[r->len] = hash_bits(x, n);
    }
    ++r->len;
    return f;
}

/* timing wrapper for the cpu_baseline leg of bench.py: wall-clock stamps at evaluation #mark and
 * at the last evaluation, so the trial phase can be timed without the init phase */
#include <sys/time.h>
typedef struct { orc_func inner; void *inner_data; long mark, count; double t_first, t_mark, t_last; } orc_timer;
static double now_s(void) { struct timeval tv; gettimeofday(&tv, NULL); return (double) tv.tv_sec + 1e-6 * (double) tv.tv_usec; }
double orc_timing_callback(unsigned n, const double *x, double *grad, void *data)
{
    orc_timer *t = (orc_timer *) data;
    double f = t->inner(n, x, grad, t->inner_data);
    double now = now_s();
    if (t->count == 0) t->t_first = now;
    ++t->count;
    if (t->count == t->mark) t->t_mark = now;
    t->t_last = now;
    return f;
}
