/* port_sobol.c — CPU ORACLE (test infrastructure): restatement of the reference's stateful Gray-code Sobol
 * generator, src/util/sobolseq.c:109-264 (sobol_gen :112-135 with the moving binary point b[i], sobol_init
 * :139-198, nlopt_sobol_next :236-242, nlopt_sobol_skip :255-264), over the Joe-Kuo 2003 data re-packed in
 * nlopt_amd/csrc/sobol_jk2003_table.h.  Pinned against the real reference's nlopt_sobol_* in
 * tests/test_oracle_pins.py.  The product computes points by index instead (nlopt_amd/csrc/sobol.c). */
#include "port_oracle.h"
#include "sobol_jk2003_table.h"
#include <stdlib.h>

struct orc_sobol_s { unsigned sdim; uint32_t *m; uint32_t *x; unsigned *b; uint32_t n; };

orc_sobol *orc_sobol_create(unsigned sdim)
{
    orc_sobol *s;
    unsigned i, j, k, pos = 0;
    if (!sdim || sdim > NLA_SOBOL_MAXDIM) return NULL;                       /* sobolseq.c:143-144 */
    s = (orc_sobol *) calloc(1, sizeof *s);
    s->sdim = sdim;
    s->m = (uint32_t *) calloc((size_t) 32 * sdim, sizeof(uint32_t));
    s->x = (uint32_t *) calloc(sdim, sizeof(uint32_t));
    s->b = (unsigned *) calloc(sdim, sizeof(unsigned));
    for (j = 0; j < 32; ++j) s->m[(size_t) j * sdim] = 1;
    for (i = 1; i < sdim; ++i) {
        uint32_t a = nla_sobol_packed[pos];
        unsigned d = 0;
        while (a) { ++d; a >>= 1; }
        d--;
        for (j = 0; j < d; ++j) s->m[(size_t) j * sdim + i] = nla_sobol_packed[pos + 1 + j];
        for (j = d; j < 32; ++j) {
            a = nla_sobol_packed[pos];
            s->m[(size_t) j * sdim + i] = s->m[(size_t) (j - d) * sdim + i];
            for (k = 0; k < d; ++k) {
                s->m[(size_t) j * sdim + i] ^= ((a & 1) * s->m[(size_t) (j - d + k) * sdim + i]) << (d - k);
                a >>= 1;
            }
        }
        pos += 1 + d;
    }
    return s;
}
void orc_sobol_destroy(orc_sobol *s) { if (s) { free(s->m); free(s->x); free(s->b); free(s); } }

static int gen(orc_sobol *s, double *x)                                       /* sobol_gen, :112-135 */
{
    unsigned c, b, i;
    if (s->n == 4294967295U) return 0;
    c = (unsigned) __builtin_ctz(~s->n);
    s->n++;
    for (i = 0; i < s->sdim; ++i) {
        b = s->b[i];
        if (b >= c) {
            s->x[i] ^= s->m[(size_t) c * s->sdim + i] << (b - c);
            x[i] = ((double) s->x[i]) / (1U << (b + 1));
        } else {
            s->x[i] = (s->x[i] << (c - b)) ^ s->m[(size_t) c * s->sdim + i];
            s->b[i] = c;
            x[i] = ((double) s->x[i]) / (1U << (c + 1));
        }
    }
    return 1;
}
void orc_sobol_next(orc_sobol *s, double *x, const double *lb, const double *ub)   /* :236-242 */
{
    unsigned i;
    gen(s, x);
    for (i = 0; i < s->sdim; ++i) x[i] = lb[i] + (ub[i] - lb[i]) * x[i];
}
void orc_sobol_next01(orc_sobol *s, double *x) { gen(s, x); }
void orc_sobol_skip(orc_sobol *s, unsigned n, double *x)                      /* :255-264; no-op on a NULL generator */
{
    if (s) { unsigned k = 1; while (k * 2 < n) k *= 2; while (k-- > 0) gen(s, x); }
}
