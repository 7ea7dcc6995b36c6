/* mt_host.c — host side of the Mersenne-Twister stream.
 *
 * (1) The thread-local MT19937 generator behind nlopt_srand / nlopt_urand / nlopt_iurand /
 *     nlopt_nrand: same word stream, same word-to-sample rules as the reference
 *     (src/util/mt19937ar.c:80-131 seeding/regeneration/tempering, :194-232 samplers;
 *     src/api/general.c:230-246 srand bookkeeping).
 * (2) What the reference does not have: GF(2) jump-ahead.  The device consumes the *same* stream
 *     at known word offsets, so it needs the generator state at arbitrary
 *     block offsets.  MT19937's word sequence satisfies a linear recurrence over GF(2) whose
 *     characteristic polynomial phi(t) has degree 19937; with g(t) = t^J mod phi(t),
 *         x[m+J] = XOR_{i : g_i = 1} x[m+i]        for every in-sequence word index m,
 *     so a state J words ahead is a GF(2) combination of 19937+624 consecutive words.  phi is
 *     recovered once per process by Berlekamp-Massey on one output bit; t^(624*2^k) mod phi by
 *     repeated squaring (phi is sparse, so reduction is cheap).  The device applies g with a
 *     kernel (hip/mt_kernels.hip); nla_mt_apply_jump_host() is the host twin used to place the
 *     host generator after a run and by the CPU tests.
 */
#include "nla_internal.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * (1) the generator
 * ------------------------------------------------------------------------------------------- */
static __thread uint32_t t_mt[NLA_MT_N];
static __thread int t_mti = NLA_MT_N + 1;      /* N+1 = never seeded (mt19937ar.c:77) */
static __thread int t_srand_called = 0;        /* general.c:230 */

void nla_mt_seed_array(uint32_t mt[NLA_MT_N], unsigned long seed)   /* mt19937ar.c:80-93 */
{
    uint32_t prev = (uint32_t) (seed & 0xffffffffUL);
    mt[0] = prev;
    for (int i = 1; i < NLA_MT_N; ++i) {
        prev = 1812433253U * (prev ^ (prev >> 30)) + (uint32_t) i;
        mt[i] = prev;
    }
}

static inline uint32_t mt_twist(uint32_t hi, uint32_t lo, uint32_t far)
{
    uint32_t y = (hi & 0x80000000U) | (lo & 0x7fffffffU);
    return far ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
}

void nla_mt_regen(uint32_t mt[NLA_MT_N])                            /* mt19937ar.c:108-117 */
{
    int k;
    for (k = 0; k < NLA_MT_N - NLA_MT_M; ++k) mt[k] = mt_twist(mt[k], mt[k + 1], mt[k + NLA_MT_M]);
    for (; k < NLA_MT_N - 1; ++k) mt[k] = mt_twist(mt[k], mt[k + 1], mt[k + NLA_MT_M - NLA_MT_N]);
    mt[NLA_MT_N - 1] = mt_twist(mt[NLA_MT_N - 1], mt[0], mt[NLA_MT_M - 1]);
}

uint32_t nla_mt_temper(uint32_t y)                                  /* mt19937ar.c:125-128 */
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= y >> 18;
    return y;
}

void nlopt_srand(unsigned long seed)                                /* general.c:231-235 */
{
    t_srand_called = 1;
    nla_mt_seed_array(t_mt, seed);
    t_mti = NLA_MT_N;
}

void nla_init_genrand(unsigned long seed)                           /* mt19937ar.c:80-95 as a call of its own (the shim's nlopt_init_genrand) */
{
    nla_mt_seed_array(t_mt, seed);
    t_mti = NLA_MT_N;
}

void nlopt_srand_time(void)                                         /* general.c:237-240 */
{
    nlopt_srand(nla_time_seed() + (unsigned long) nla_thread_id() * 314159);
}

void nla_srand_time_default(void)                                   /* general.c:242-246 */
{
    if (!t_srand_called) nlopt_srand_time();
}

uint32_t nla_genrand_int32(void)                                    /* mt19937ar.c:97-131 */
{
    if (t_mti >= NLA_MT_N) {
        if (t_mti == NLA_MT_N + 1) { nla_mt_seed_array(t_mt, 5489UL); }
        nla_mt_regen(t_mt);
        t_mti = 0;
    }
    return nla_mt_temper(t_mt[t_mti++]);
}

static double res53(void)                                           /* mt19937ar.c:194-198 */
{
    uint32_t a = nla_genrand_int32() >> 5;       /* drawn first: high 27 bits */
    uint32_t b = nla_genrand_int32() >> 6;
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

double nlopt_urand(double a, double b) { return a + (b - a) * res53(); }          /* :203-206 */
int nlopt_iurand(int n) { return (int) (nla_genrand_int32() % (uint32_t) n); }      /* :209-212 */

double nlopt_nrand(double mean, double stddev)                                    /* :216-232 */
{
    double v1, v2, s;
    do {
        v1 = nlopt_urand(-1, 1);
        v2 = nlopt_urand(-1, 1);
        s = v1 * v1 + v2 * v2;
    } while (s >= 1.0);
    if (s == 0) return mean;
    return mean + v1 * sqrt(-2 * log(s) / s) * stddev;
}

/* Export the generator as (block array, words already consumed from it): the device stream's
 * global word index g addresses word g%624 of the (g/624)-th regeneration of this array, and
 * the next word the host would draw is g = *consumed.  A freshly seeded array (mti == 624) is
 * regenerated first so that block 0 is an output block whose every word is in-sequence. */
void nla_mt_export(uint32_t mt[NLA_MT_N], int *consumed)
{
    if (t_mti >= NLA_MT_N) {
        if (t_mti == NLA_MT_N + 1) nla_mt_seed_array(t_mt, 5489UL);
        nla_mt_regen(t_mt);
        t_mti = 0;
    }
    memcpy(mt, t_mt, sizeof t_mt);
    *consumed = t_mti;
}

void nla_mt_import(const uint32_t mt[NLA_MT_N], int consumed)
{
    memcpy(t_mt, mt, sizeof t_mt);
    t_mti = consumed;
}

/* ---------------------------------------------------------------------------------------------
 * (2) GF(2) polynomials of degree < 19937, bit i of word i/64 = coefficient of t^i
 * ------------------------------------------------------------------------------------------- */
#define DEG NLA_MT_DEG
#define PW  NLA_MT_POLYWORDS          /* 312 words = 19968 bits >= 19937 */

static uint64_t g_phi[PW + 1];        /* phi, including the t^19937 term */
static int g_phi_exp[512];            /* exponents of phi's nonzero terms, ascending */
static int g_phi_terms = 0;
static pthread_once_t g_phi_once = PTHREAD_ONCE_INIT;

static inline int bit_get(const uint64_t *p, int i) { return (int) ((p[i >> 6] >> (i & 63)) & 1U); }
static inline void bit_flip(uint64_t *p, int i) { p[i >> 6] ^= (uint64_t) 1 << (i & 63); }

/* Berlekamp-Massey over GF(2) on s_k = bit 0 of the k-th in-sequence untempered word.  The
 * minimal polynomial of any nonzero MT19937 sequence is the (irreducible) characteristic
 * polynomial itself; we assert degree 19937. */
static void compute_phi(void)
{
    enum { NS = 2 * DEG + 64, W = (DEG + 64) / 64 + 2 };
    uint32_t mt[NLA_MT_N];
    unsigned char *s = (unsigned char *) malloc(NS);
    uint64_t *C = (uint64_t *) calloc(W, 8), *B = (uint64_t *) calloc(W, 8), *T = (uint64_t *) calloc(W, 8);
    uint64_t *R = (uint64_t *) calloc(W, 8);     /* R bit i = s[n-i] */
    int L = 0, m = 1, k = 0;
    if (!s || !C || !B || !T || !R) abort();
    nla_mt_seed_array(mt, 4357UL);
    nla_mt_regen(mt);                            /* block of in-sequence words */
    for (int n = 0; n < NS; ++n) {
        if (k == NLA_MT_N) { nla_mt_regen(mt); k = 0; }
        s[n] = (unsigned char) (mt[k++] & 1U);
    }
    C[0] = B[0] = 1;
    for (int n = 0; n < NS; ++n) {
        uint64_t acc = 0, carry = s[n];
        int lw = (L >> 6) + 1, d;
        for (int w = 0; w < W; ++w) {            /* R <<= 1, insert s[n] */
            uint64_t nc = R[w] >> 63;
            R[w] = (R[w] << 1) | carry;
            carry = nc;
        }
        for (int w = 0; w <= lw && w < W; ++w) acc ^= C[w] & R[w];
        d = __builtin_parityll(acc);
        if (!d) { ++m; continue; }
        if (2 * L <= n) memcpy(T, C, (size_t) W * 8);
        {   /* C ^= B << m */
            int ws = m >> 6, bs = m & 63;
            for (int w = W - 1; w >= ws; --w) {
                uint64_t v = B[w - ws] << bs;
                if (bs && w - ws - 1 >= 0) v |= B[w - ws - 1] >> (64 - bs);
                C[w] ^= v;
            }
        }
        if (2 * L <= n) { L = n + 1 - L; memcpy(B, T, (size_t) W * 8); m = 1; }
        else ++m;
    }
    if (L != DEG) abort();
    /* connection polynomial C (s[n] = XOR_{i>=1} C_i s[n-i]) -> characteristic phi_j = C_{L-j} */
    memset(g_phi, 0, sizeof g_phi);
    g_phi_terms = 0;
    for (int j = 0; j <= DEG; ++j)
        if (bit_get(C, DEG - j)) {
            bit_flip(g_phi, j);
            if (g_phi_terms >= (int) (sizeof g_phi_exp / sizeof g_phi_exp[0])) abort();
            g_phi_exp[g_phi_terms++] = j;
        }
    free(s); free(C); free(B); free(T); free(R);
}

int nla_mt_charpoly_terms(const int **exps)
{
    pthread_once(&g_phi_once, compute_phi);
    if (exps) *exps = g_phi_exp;
    return g_phi_terms;
}

/* reduce a polynomial of degree < 2*DEG (in buf, 2*PW+2 words) modulo phi, using sparsity */
static void poly_reduce(uint64_t *buf)
{
    for (int d = 2 * DEG - 1; d >= DEG; --d)
        if (bit_get(buf, d)) {
            int sh = d - DEG;
            for (int k = 0; k < g_phi_terms; ++k) bit_flip(buf, sh + g_phi_exp[k]);
        }
}

static uint64_t spread32(uint32_t v)      /* interleave zeros: bit i -> bit 2i */
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000ffff0000ffffULL;
    x = (x | (x << 8)) & 0x00ff00ff00ff00ffULL;
    x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0fULL;
    x = (x | (x << 2)) & 0x3333333333333333ULL;
    x = (x | (x << 1)) & 0x5555555555555555ULL;
    return x;
}

static void poly_square_mod(const uint64_t a[PW], uint64_t out[PW])
{
    uint64_t buf[2 * PW + 2];
    memset(buf, 0, sizeof buf);
    for (int w = 0; w < PW; ++w) {          /* squaring over GF(2) = spreading the bits */
        buf[2 * w] = spread32((uint32_t) a[w]);
        buf[2 * w + 1] = spread32((uint32_t) (a[w] >> 32));
    }
    poly_reduce(buf);
    memcpy(out, buf, PW * 8);
}

static void poly_mul_t_mod(uint64_t a[PW])   /* a <- a * t mod phi */
{
    uint64_t carry = 0;
    for (int w = 0; w < PW; ++w) {
        uint64_t nc = a[w] >> 63;
        a[w] = (a[w] << 1) | carry;
        carry = nc;
    }
    if (bit_get(a, DEG))
        for (int w = 0; w < PW; ++w) a[w] ^= g_phi[w];     /* clears bit DEG (phi_DEG = 1) */
}

/* g = t^J mod phi for an arbitrary word count J (square-and-multiply, MSB first) */
void nla_mt_jump_poly_words(uint64_t J, uint64_t g[PW])
{
    pthread_once(&g_phi_once, compute_phi);
    memset(g, 0, PW * 8);
    g[0] = 1;
    for (int b = 63; b >= 0; --b) {
        uint64_t tmp[PW];
        poly_square_mod(g, tmp);
        memcpy(g, tmp, PW * 8);
        if ((J >> b) & 1U) poly_mul_t_mod(g);
    }
}

/* cached table: pow2[k] = t^(624 * 2^k) mod phi  (jump by 2^k regenerations) */
static uint64_t *g_pow2[NLA_MT_MAXPOW2];
static pthread_mutex_t g_pow2_lock = PTHREAD_MUTEX_INITIALIZER;

const uint64_t *nla_mt_jump_poly_pow2(int k)
{
    if (k < 0 || k >= NLA_MT_MAXPOW2) return NULL;
    pthread_once(&g_phi_once, compute_phi);
    pthread_mutex_lock(&g_pow2_lock);
    for (int j = 0; j <= k; ++j) {
        if (g_pow2[j]) continue;
        g_pow2[j] = (uint64_t *) calloc(PW, 8);
        if (!g_pow2[j]) abort();
        if (j == 0) bit_flip(g_pow2[0], NLA_MT_N);            /* t^624, degree < 19937: no reduction */
        else poly_square_mod(g_pow2[j - 1], g_pow2[j]);
    }
    pthread_mutex_unlock(&g_pow2_lock);
    return g_pow2[k];
}

/* dst = block array J words ahead of src, where g = t^J mod phi and src is an in-sequence block */
void nla_mt_apply_jump_host(const uint64_t g[PW], const uint32_t src[NLA_MT_N], uint32_t dst[NLA_MT_N])
{
    enum { NB = (DEG + NLA_MT_N - 1) / NLA_MT_N + 1 };        /* 33 blocks >= 19937+624 words */
    uint32_t *x = (uint32_t *) malloc(sizeof(uint32_t) * NB * NLA_MT_N);
    uint32_t blk[NLA_MT_N];
    if (!x) abort();
    memcpy(blk, src, sizeof blk);
    for (int b = 0; b < NB; ++b) {
        memcpy(x + b * NLA_MT_N, blk, sizeof blk);
        nla_mt_regen(blk);
    }
    memset(dst, 0, sizeof(uint32_t) * NLA_MT_N);
    for (int i = 0; i < DEG; ++i)
        if (bit_get(g, i)) {
            const uint32_t *xi = x + i;
            for (int j = 0; j < NLA_MT_N; ++j) dst[j] ^= xi[j];
        }
    free(x);
}

/* block array `regens` regenerations after src (binary decomposition over the pow2 table) */
void nla_mt_advance_blocks_host(const uint32_t src[NLA_MT_N], uint64_t regens, uint32_t dst[NLA_MT_N])
{
    uint32_t cur[NLA_MT_N], nxt[NLA_MT_N];
    memcpy(cur, src, sizeof cur);
    /* small remainders are cheaper by plain regeneration than by a 6M-word-xor jump */
    while (regens & 0xfff) { nla_mt_regen(cur); --regens; }
    for (int k = 12; regens >> k; ++k)
        if ((regens >> k) & 1U) {
            nla_mt_apply_jump_host(nla_mt_jump_poly_pow2(k), cur, nxt);
            memcpy(cur, nxt, sizeof cur);
        }
    memcpy(dst, cur, sizeof cur);
}
