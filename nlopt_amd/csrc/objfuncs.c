/* objfuncs.c — registry of device objectives: the host callbacks a user hands to
 * nlopt_set_min_objective(), recognised by pointer identity so the dispatcher can run the HIP
 * evaluator for the same formula (include/nlopt_amd.h part 1; SURVEY.md §8b "required
 * extension").  The callbacks themselves are the sequential evaluators of objfuncs.h — the same
 * source the oracle compiles — and are what the host-callback path calls. */
#include "nla_internal.h"
#include "objfuncs.h"

#define DEF(name, id) static double name(unsigned n, const double *x, double *g, void *d) { (void) d; return nla_obj_eval_seq(id, n, x, g); }
DEF(cb_rastrigin, NLA_OBJ_RASTRIGIN)
DEF(cb_ackley, NLA_OBJ_ACKLEY)
DEF(cb_griewank, NLA_OBJ_GRIEWANK)
DEF(cb_rosenbrock, NLA_OBJ_ROSENBROCK)
DEF(cb_levy, NLA_OBJ_LEVY)
DEF(cb_sphere, NLA_OBJ_SPHERE)

static const nlopt_func g_cb[NLA_OBJ_COUNT] = { cb_rastrigin, cb_ackley, cb_griewank, cb_rosenbrock, cb_levy, cb_sphere };
static const char *const g_name[NLA_OBJ_COUNT] = { "rastrigin", "ackley", "griewank", "rosenbrock", "levy", "sphere" };

nlopt_func nlopt_amd_objective(int id) { return (id >= 0 && id < NLA_OBJ_COUNT) ? g_cb[id] : NULL; }

int nlopt_amd_objective_id(nlopt_func f)
{
    int i;
    if (!f) return -1;
    for (i = 0; i < NLA_OBJ_COUNT; ++i) if (g_cb[i] == f) return i;
    return -1;
}

const char *nlopt_amd_objective_name(int id) { return (id >= 0 && id < NLA_OBJ_COUNT) ? g_name[id] : NULL; }
void nlopt_amd_objective_box(int id, double *lo, double *hi) { nla_obj_box(id, lo, hi); }

static double cb_blocksum(unsigned n, const double *x, double *grad, void *data)
{
    const unsigned *qQ = (const unsigned *) data;
    return nla_con_blocksum_seq(n, x, grad, qQ[0], qQ[1]);
}
nlopt_func nlopt_amd_constraint_blocksum(void) { return cb_blocksum; }
int nlopt_amd_constraint_id(nlopt_func f) { return f == cb_blocksum ? NLA_CON_BLOCKSUM : -1; }

int nlopt_amd_device_count(void) { return nla_dev_count(); }
