/* sobol.c — the low-discrepancy sampler of NLOPT_G*_MLSL_LDS (src/util/sobolseq.c:109-264), host side.
 *
 * The reference walks the Sobol sequence in Gray-code order with a stateful generator: x_{k+1} = x_k XOR v_c,
 * c = position of the rightmost zero bit of k (sobol_gen :112-135), keeping each coordinate as an integer with a
 * moving binary point.  In fixed 32-bit fraction form that is x_k = XOR over the set bits c of gray(k) = k ^ (k>>1)
 * of V[c] = m_c << (31 - c), and the value it returns is exactly x_k / 2^32 (its integer is x_k with the trailing
 * zero bits dropped, divided by the matching power of two) — so any point of the sequence is a pure function of
 * its index and the batch of an MLSL iteration is generated in one launch (hip/mlsl_kernels.hip).
 * This file builds the direction table V from the Joe-Kuo 2003 data (sobol_init :139-198) and mirrors
 * nlopt_sobol_skip's count (:255-264).  Dimension 1..1111, at most 2^32-1 points, as in the reference. */
#include "nla_internal.h"
#include "sobol_jk2003_table.h"
#include <stdlib.h>

/* V[c*sdim + i], c < 32: direction numbers as 32-bit fractions.  0 = unsupported dimension (the reference returns
 * a NULL generator and MLSL falls back to pseudo-random sampling, mlsl.c:306,355-359). */
int nla_sobol_directions(unsigned sdim, uint32_t *V)
{
    unsigned i, j, k, pos = 0;
    uint32_t m[32];
    if (!sdim || sdim > NLA_SOBOL_MAXDIM) return 0;
    for (j = 0; j < 32; ++j) V[(size_t) j * sdim] = 1u << (31 - j);                 /* first dimension: m_j = 1 */
    for (i = 1; i < sdim; ++i) {
        const uint32_t a = nla_sobol_packed[pos];
        unsigned d = 0;
        uint32_t t = a;
        while (t) { ++d; t >>= 1; }
        --d;                                                                        /* degree of the polynomial */
        for (j = 0; j < d; ++j) m[j] = nla_sobol_packed[pos + 1 + j];
        for (j = d; j < 32; ++j) {                                                  /* recurrence, sobolseq.c:165-173 */
            uint32_t ac = a, v = m[j - d];
            for (k = 0; k < d; ++k) { v ^= ((ac & 1) * m[j - d + k]) << (d - k); ac >>= 1; }
            m[j] = v;
        }
        for (j = 0; j < 32; ++j) V[(size_t) j * sdim + i] = m[j] << (31 - j);
        pos += 1 + d;
    }
    return 1;
}

/* points discarded by nlopt_sobol_skip(s, n, .): the largest power of two smaller than n (sobolseq.c:255-264) */
uint32_t nla_sobol_skip_count(unsigned n)
{
    uint32_t k = 1;
    while (k * 2 < n) k *= 2;
    return k;
}

/* point number `index` (1-based: the index-th call of nlopt_sobol_next01) of the sdim-dimensional sequence */
void nla_sobol_point01(unsigned sdim, const uint32_t *V, uint32_t index, double *x)
{
    const uint32_t g = index ^ (index >> 1);
    unsigned i, c;
    for (i = 0; i < sdim; ++i) {
        uint32_t acc = 0;
        for (c = 0; c < 32; ++c) if ((g >> c) & 1u) acc ^= V[(size_t) c * sdim + i];
        x[i] = (double) acc / 4294967296.0;
    }
}
