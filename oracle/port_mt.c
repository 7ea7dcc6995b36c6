/* port_mt.c — CPU ORACLE (test infrastructure): MT19937 word stream and the three samplers.
 * Restates src/util/mt19937ar.c:80-131 (seeding, block regeneration, tempering) and :194-232
 * (res53 / urand / iurand / nrand).  One global stream (the reference's is thread-local; the
 * oracle is single-threaded).  Pinned by tests against the classic KAT (seed 5489 -> 3499211612)
 * and against oracle/_ref's nlopt_urand/nlopt_iurand/nlopt_nrand word for word. */
#include "port_oracle.h"
#include <math.h>
#include <string.h>

enum { MTN = 624, MTM = 397 };
static uint32_t g_mt[MTN];
static int g_mti = MTN + 1;           /* MTN+1: never seeded (mt19937ar.c:77) */
static uint64_t g_drawn = 0;

void orc_srand(unsigned long seed)    /* mt19937ar.c:80-93 */
{
    uint32_t prev = (uint32_t) (seed & 0xffffffffUL);
    g_mt[0] = prev;
    for (int i = 1; i < MTN; ++i) {
        prev = 1812433253U * (prev ^ (prev >> 30)) + (uint32_t) i;
        g_mt[i] = prev;
    }
    g_mti = MTN;
    g_drawn = 0;
}

static inline uint32_t twist(uint32_t hi, uint32_t lo, uint32_t far)
{
    uint32_t y = (hi & 0x80000000U) | (lo & 0x7fffffffU);
    return far ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
}

static void regenerate(void)          /* mt19937ar.c:102-120 */
{
    int k;
    if (g_mti == MTN + 1) orc_srand(5489UL);
    for (k = 0; k < MTN - MTM; ++k) g_mt[k] = twist(g_mt[k], g_mt[k + 1], g_mt[k + MTM]);
    for (; k < MTN - 1; ++k)        g_mt[k] = twist(g_mt[k], g_mt[k + 1], g_mt[k + MTM - MTN]);
    g_mt[MTN - 1] = twist(g_mt[MTN - 1], g_mt[0], g_mt[MTM - 1]);
    g_mti = 0;
}

uint32_t orc_genrand_int32(void)      /* mt19937ar.c:97-131 */
{
    uint32_t y;
    if (g_mti >= MTN) regenerate();
    y = g_mt[g_mti++];
    ++g_drawn;
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= y >> 18;
    return y;
}

static double res53(void)             /* mt19937ar.c:194-198: first word -> high 27 bits */
{
    uint32_t a = orc_genrand_int32() >> 5;
    uint32_t b = orc_genrand_int32() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

double orc_urand(double a, double b) { return a + (b - a) * res53(); }              /* :203-206 */
int orc_iurand(int n) { return (int) (orc_genrand_int32() % (uint32_t) n); }         /* :209-212 */

double orc_nrand(double mean, double stddev)                                        /* :216-232 */
{
    double v1, v2, s;
    do {
        v1 = orc_urand(-1, 1);
        v2 = orc_urand(-1, 1);
        s = v1 * v1 + v2 * v2;
    } while (s >= 1.0);
    if (s == 0) return mean;
    return mean + v1 * sqrt(-2 * log(s) / s) * stddev;
}

void orc_mt_get_state(uint32_t mt[624], int *mti) { memcpy(mt, g_mt, sizeof g_mt); *mti = g_mti; }
void orc_mt_set_state(const uint32_t mt[624], int mti) { memcpy(g_mt, mt, sizeof g_mt); g_mti = mti; }
uint64_t orc_mt_words_drawn(void) { return g_drawn; }
