/*
 * dreamzs_oracle.c -- CPU restatement of PyDREAM's MT-DREAM(ZS) hot path
 * (pydream/Dream.py:193-422 `Dream.astep` and everything it calls, driven as
 * pydream/core.py:89-129 `_sample_dream` drives it).
 *
 * TEST INFRASTRUCTURE ONLY -- see dreamzs_oracle.h.  Parity status: PINNED
 * against the reference run in the build container: the committed fixtures
 * (tests/golden/*.npz, made by tests/golden/make_golden.py) and, beyond them,
 * 2440 random configurations run through the reference itself
 * (tests/golden/fuzz_reference.py; profiles/r03_fuzz_parity.txt).
 *
 * Plain scalar C, one chain at a time, written to be read next to the
 * reference.  The only things that are NOT the reference's are the ones the
 * reference leaves unpinned: the random bit-stream
 * (here: Philox4x32-10 counters, DESIGN.md "Random contract") and the order of
 * floating-point reductions (DESIGN.md "Reduction contract").  Both are
 * restated here independently of the HIP sources from the written contract.
 *
 * Compile with -ffp-contract=off (see oracle/Makefile): every fused
 * multiply-add below is an explicit fma()/fmaf() call.
 */
#include "dreamzs_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_VERSION 1
#define HOT __attribute__((target_clones("default", "fma")))

static __thread char g_err[512];
static int fail(const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return -1; }
int orc_version(void) { return ORC_VERSION; }
int orc_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
const char* orc_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ */
/* Random contract                                                     */
/* ------------------------------------------------------------------ */

/* Philox4x32-10 (Salmon et al., SC'11).  key = 64-bit seed. */
void orc_philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4])
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* stream id = kind | try<<4 | phase<<12 | round<<16 */
enum { K_CTRL = 0, K_PT = 1, K_DIM = 2, K_BND = 3 };
uint32_t orc_stream_id(int kind, int tr, int phase, int round)
{
    return (uint32_t)kind | ((uint32_t)tr << 4) | ((uint32_t)phase << 12) | ((uint32_t)round << 16);
}

/* 53-bit uniform in [0,1) from two words (27 high bits of hi, 26 of lo). */
double orc_u53(uint32_t hi, uint32_t lo)
{
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}
/* 32-bit uniform in (0,1): (w + 1/2) 2^-32, exact in double. */
double orc_u32(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* Two independent standard normals in binary32 from two words (contract v2, DESIGN.md "Random contract"):
 * Box-Muller in a form made of +, *, fma, one correctly rounded sqrt and integer bit operations only.
 *   radius  u = n 2^-24 with n = (w1 >> 8) | 1, an odd integer in [1, 2^24): with n = 2^k m, m in [sqrt(1/2), sqrt 2),
 *           t = -ln u = (24 - k) ln2 - ln m;  ln m = f g(f), f = m - 1, g a degree-8 polynomial (|error| < 2e-8; no
 *           cancellation as u -> 1, where k = 24);
 *           r = sqrt(t) (so that r sqrt2 cos = sqrt(-2 ln u) cos);  |z| <= sqrt(2 * 24 ln 2) = 5.77
 *   angle   theta = pi/4 + phi, phi = (pi/2) y, y = 1.fraction(w2 >> 9) - 1.5 in [-1/2, 1/2):
 *           sqrt2 cos theta = cos phi - sin phi, sqrt2 sin theta = cos phi + sin phi (polynomials on |phi| <= pi/4);
 *           the quadrant comes from two independent sign bits (bit 0 of w2 for the cosine branch, bit 0 of w1 for the sine
 *           branch), which makes the angle uniform on the whole circle.
 * z0 = +-r (cos phi - sin phi), z1 = +-r (cos phi + sin phi). */
HOT void orc_normal32_pair(uint32_t w1, uint32_t w2, float* z0, float* z1)
{
    const uint32_t n = (w1 >> 8) | 1u;
    const float nf = (float)n;                                      /* exact: 24 bits */
    const uint32_t ix = f2u(nf) + 0x004afb0du;                      /* + (bits(1) - bits(sqrt(1/2))): exponent field = that of n / sqrt(1/2) */
    const float Ef = (float)(151 - (int)(ix >> 23));                /* 24 - k */
    const float f = u2f((ix & 0x007fffffu) + 0x3f3504f3u) - 1.0f;   /* m - 1, m in [sqrt(1/2), sqrt 2); exact */
    float g = 0.08743945509195328f;
    g = fmaf(g, f, -0.14377330243587494f);
    g = fmaf(g, f, 0.14949095249176025f);
    g = fmaf(g, f, -0.16560696065425873f);
    g = fmaf(g, f, 0.19956977665424347f);
    g = fmaf(g, f, -0.2500215470790863f);
    g = fmaf(g, f, 0.3333418369293213f);
    g = fmaf(g, f, -0.49999988079071045f);
    g = fmaf(g, f, 1.0f);
    const float t = fmaf(-f, g, Ef * 0.693147182464599609375f);
    int32_t tb = (int32_t)f2u(t);
    if (tb < 0) tb = 0;                                             /* a rounding-level negative becomes +0 */
    const float rad = sqrtf(u2f((uint32_t)tb));                     /* IEEE, correctly rounded */
    const float y = u2f(0x3f800000u | (w2 >> 9)) - 1.5f;            /* exact */
    const float ph = y * 1.57079637050628662109375f;
    const float y2 = ph * ph;
    float sp = -0.00019598381186369807f;
    sp = fmaf(sp, y2, 0.008332823403179646f);
    sp = fmaf(sp, y2, -0.1666666567325592f);
    const float sn = fmaf(ph * y2, sp, ph);
    float cp = 2.447330734867137e-05f;
    cp = fmaf(cp, y2, -0.0013887685490772128f);
    cp = fmaf(cp, y2, 0.041666653007268906f);
    cp = fmaf(cp, y2, -0.5f);
    const float cs = fmaf(cp, y2, 1.0f);
    const float r0 = u2f(f2u(rad) | (w2 << 31)), r1 = u2f(f2u(rad) | (w1 << 31));
    *z0 = r0 * (cs - sn); *z1 = r1 * (cs + sn);
}
/* npairs Box-Muller pairs from consecutive Philox counters (words 2 and 3, as the DIM stream uses them): for the
 * statistical tests of the generator */
void orc_normal32_fill(uint64_t seed, int64_t npairs, float* out)
{
    for (int64_t i = 0; i < npairs; ++i) {
        uint32_t w[4];
        orc_philox4x32_10(seed, (uint32_t)i, (uint32_t)(i >> 32), 7u, 11u, w);
        orc_normal32_pair(w[2], w[3], out + 2 * i, out + 2 * i + 1);
    }
}
float orc_normal32(uint32_t w1, uint32_t w2) { float a, b; orc_normal32_pair(w1, w2, &a, &b); return a; }
float orc_normal32_sin(uint32_t w1, uint32_t w2) { float a, b; orc_normal32_pair(w1, w2, &a, &b); return b; }
/* 16-bit uniform in (0,1): (h + 1/2) 2^-16 */
double orc_u16(uint32_t h) { return ((double)(h & 0xffffu) + 0.5) * (1.0 / 65536.0); }
/* np.random.uniform(low, high) (Dream.py:696) from a 16-bit draw h: low + (high - low) (h + 1/2) 2^-16 evaluated as ONE
 * fused multiply-add, h c1 + c0 with c1 = (high - low) 2^-16 and c0 = low + (high - low) 2^-17 (contract v2) */
double orc_uniform16(uint32_t h, double low, double high)
{
    const double c1 = (high - low) * (1.0 / 65536.0), c0 = low + (high - low) * (1.0 / 131072.0);
    return fma((double)(h & 0xffffu), c1, c0);
}

/* ------------------------------------------------------------------ */
/* Elementary functions (binary64), DESIGN.md "Elementary functions"   */
/* ------------------------------------------------------------------ */
#define LN2_HI 6.93147180369123816490e-01 /* low 32 bits zero */
#define LN2_LO 1.90821492927058770002e-10

HOT double orc_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.2) return 0.0;
    double kf = floor(x * 1.4426950408889634 + 0.5);
    double r = fma(-kf, LN2_HI, x);
    r = fma(-kf, LN2_LO, r);
    double p = 1.0 / 6227020800.0;        /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);     /* 1/12! */
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kf;
    if (k < -1000) { p = p * u2d((uint64_t)(1023 - 1000) << 52); k += 1000; }
    return p * u2d((uint64_t)(k + 1023) << 52);
}

HOT double orc_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return NAN;
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    uint64_t b = d2u(x);
    if ((b >> 52) == 0) { x = x * 18014398509481984.0; e = -54; b = d2u(x); } /* 2^54 */
    e += (int)(b >> 52) - 1023;
    double m = u2d((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 23.0;
    p = fma(p, z, 1.0 / 21.0);
    p = fma(p, z, 1.0 / 19.0);
    p = fma(p, z, 1.0 / 17.0);
    p = fma(p, z, 1.0 / 15.0);
    p = fma(p, z, 1.0 / 13.0);
    p = fma(p, z, 1.0 / 11.0);
    p = fma(p, z, 1.0 / 9.0);
    p = fma(p, z, 1.0 / 7.0);
    p = fma(p, z, 1.0 / 5.0);
    p = fma(p, z, 1.0 / 3.0);
    double t = 2.0 * s;
    double lm = fma(t * z, p, t);
    double ef = (double)e;
    return fma(ef, LN2_HI, fma(ef, LN2_LO, lm));
}

/* numpy.nan_to_num (Dream.py:323, 332, 334, 481, 820) */
static inline double nan_to_num(double x)
{
    if (x != x) return 0.0;
    if (x == INFINITY) return DBL_MAX;
    if (x == -INFINITY) return -DBL_MAX;
    return x;
}

/* ------------------------------------------------------------------ */
/* Reduction contract: lane(j) = (j>>1)&63, sequential within a lane,   */
/* xor-butterfly over the 64 lane partials.                            */
/* ------------------------------------------------------------------ */
static inline double butterfly64(double* p)
{
    double q[64];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof q);
    }
    return p[0];
}
HOT double orc_wave_dot(const double* a, const double* b, int d)
{
    double p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0;
    for (int j = 0; j < d; ++j) { int l = (j >> 1) & 63; p[l] = fma(a[j], b[j], p[l]); }
    return butterfly64(p);
}
static double wave_sum(const double* a, int d)
{
    double p[64];
    for (int l = 0; l < 64; ++l) p[l] = 0.0;
    for (int j = 0; j < d; ++j) { int l = (j >> 1) & 63; p[l] = p[l] + a[j]; }
    return butterfly64(p);
}

/* first category m with u < p0+..+pm (sequential cumulative sum); stands in
 * for np.random.multinomial(1, p) (Dream.py:545, 565, 595, 615, 908). */
int orc_invcdf(const double* p, int n, double u)
{
    double c = 0.0;
    for (int m = 0; m < n; ++m) { c = c + p[m]; if (u < c) return m; }
    return n - 1;
}

/* n distinct indices from range(M); stands in for random.sample(range(M), n)
 * (Dream.py:662): t-th index = mulhi(word_t, M-t), then skipped over the
 * previously chosen ones in ascending order. */
int orc_sample_distinct(const uint32_t* words, int n, uint32_t M, uint32_t* out)
{
    uint32_t sorted[16];
    if (n > 16 || (uint32_t)n > M) return -1;
    for (int t = 0; t < n; ++t) {
        uint32_t r = (uint32_t)(((uint64_t)words[t] * (uint64_t)(M - (uint32_t)t)) >> 32);
        int pos = 0;
        for (int s = 0; s < t; ++s) { if (r >= sorted[s]) { r++; pos = s + 1; } else break; }
        for (int s = t; s > pos; --s) sorted[s] = sorted[s - 1];
        sorted[pos] = r;
        out[t] = r;
    }
    return 0;
}

/* gamma table, Dream.py:172-179 */
void orc_gamma_table(int ngamma, int depairs, int d, double* out)
{
    double dec = 1.0;
    for (int lev = 1; lev <= ngamma; ++lev) {
        for (int delta = 1; delta <= depairs; ++delta)
            for (int dp = 1; dp <= d; ++dp)
                out[((size_t)(lev - 1) * depairs + (delta - 1)) * d + (dp - 1)] =
                    (2.38 / sqrt((double)(2 * delta) * (double)dp)) / dec;
        dec = dec * 2.0;
    }
}

/* ------------------------------------------------------------------ */
/* Engine state                                                        */
/* ------------------------------------------------------------------ */
enum { LK_NONE = 0, LK_MVN_DENSE = 1, LK_MVN_TRI = 2, LK_MIX = 3, LK_HOST = 4 };

struct orc_engine {
    orc_config c;
    int d, nl, N, k;
    double *mins, *maxs;
    double* gtab;
    double* Z; int64_t M; int64_t napp;     /* rows written; appends made by this run (history_lag: the last `lag` of them are not sampleable yet) */
    double *X, *lprior, *llike; int have_logp;
    double *cp_prev, *cp_new;   /* [N,d] published positions (Dream.py:424-449) */
    double *cr_probs, *cr_delta, *cr_n;         /* shared (core.py:287-289) */
    double *g_probs, *g_delta, *g_n;            /* shared (core.py:291-293) */
    double *own_cr, *own_g;                     /* per-chain copies, schedule S1 only */
    int32_t* pkind; double *pa, *pb; int have_prior;
    int lk; double* mu; double* Mx; double logF; int J; double* mixF;
    orc_logp_cb cb; void* cb_user;
    orc_exchange_cb xcb; void* xcb_user;
    int64_t gen;         /* generations done */
    int64_t ntrace;
    double *tX, *tlogp; uint8_t *tmoved, *tsnk; int32_t *ttry, *tcr;
    /* parallel tempering (core.py:131-236): per-chain temperatures, one swap attempt per generation */
    double* Tc; int tempering; int32_t* tswap;     /* tswap [trace_capacity][3]: chain a, chain b, accepted */
    /* adapt_lag (schedule S2): the totals of burn-in generations whose update has not been applied yet, oldest first:
     * pend_tot [adapt_lag + 1][nq d], pend_cnt [adapt_lag + 1][ncr + ngamma], generations pend_g0 .. pend_g0 + npend - 1; x0ring
     * [2 (adapt_lag + 1)][d]: global chain 0's published position after generation h at slot h mod 2 (adapt_lag + 1) (x0start: before
     * generation 0) -- the shift of the column sums of generation h + 1 + adapt_lag */
    double *pend_tot, *pend_cnt; int npend; int64_t pend_g0;
    double *x0ring, *x0start;
    /* scratch */
    double *pts, *refs, *work;
};

static void* zalloc(size_t n) { void* p = calloc(n ? n : 1, 1); return p; }

int orc_create(const orc_config* cfg, orc_engine** out)
{
    if (!cfg || !out) return fail("null argument");
    if (cfg->ndim < 1 || cfg->nchains < 1 || cfg->nchains_local < 1 || cfg->multitry < 1) return fail("bad sizes");
    if (cfg->multitry == 2) return fail("multitry=2 is broken in the reference (Dream.py:867-868); rejected");
    if (cfg->depairs < 1 || cfg->depairs > 8) return fail("DEpairs must be 1..8");
    if (cfg->ncr < 1 || cfg->ngamma < 1) return fail("bad nCR/gamma_levels");
    if (cfg->schedule != 1 && cfg->schedule != 2) return fail("schedule must be 1 or 2");
    if (cfg->history_lag < 0 || (cfg->history_lag && cfg->schedule != 2)) return fail("history_lag needs schedule 2 and must be >= 0");
    if (cfg->adapt_lag < 0 || cfg->adapt_lag > 4096 || (cfg->adapt_lag && cfg->schedule != 2)) return fail("adapt_lag needs schedule 2 and must be 0..4096");
    if (cfg->chain_offset < 0 || cfg->chain_offset + cfg->nchains_local > cfg->nchains) return fail("bad shard");
    orc_engine* e = zalloc(sizeof *e);
    e->c = *cfg; e->d = cfg->ndim; e->nl = cfg->nchains_local; e->N = cfg->nchains; e->k = cfg->multitry;
    int d = e->d, nl = e->nl, N = e->N, k = e->k;
    e->mins = zalloc(sizeof(double) * d); e->maxs = zalloc(sizeof(double) * d);
    for (int j = 0; j < d; ++j) { e->mins[j] = -INFINITY; e->maxs[j] = INFINITY; }
    e->gtab = zalloc(sizeof(double) * cfg->ngamma * cfg->depairs * d);
    orc_gamma_table(cfg->ngamma, cfg->depairs, d, e->gtab);
    e->Z = zalloc(sizeof(double) * (size_t)cfg->history_capacity * d);
    e->X = zalloc(sizeof(double) * nl * d); e->lprior = zalloc(sizeof(double) * nl); e->llike = zalloc(sizeof(double) * nl);
    e->cp_prev = zalloc(sizeof(double) * N * d); e->cp_new = zalloc(sizeof(double) * N * d);
    e->cr_probs = zalloc(sizeof(double) * cfg->ncr); e->cr_delta = zalloc(sizeof(double) * cfg->ncr); e->cr_n = zalloc(sizeof(double) * cfg->ncr);
    e->g_probs = zalloc(sizeof(double) * cfg->ngamma); e->g_delta = zalloc(sizeof(double) * cfg->ngamma); e->g_n = zalloc(sizeof(double) * cfg->ngamma);
    for (int m = 0; m < cfg->ncr; ++m) e->cr_probs[m] = 1.0 / (double)cfg->ncr;          /* Dream.py:134 */
    for (int m = 0; m < cfg->ngamma; ++m) e->g_probs[m] = 1.0 / (double)cfg->ngamma;     /* Dream.py:143 */
    e->own_cr = zalloc(sizeof(double) * nl * cfg->ncr); e->own_g = zalloc(sizeof(double) * nl * cfg->ngamma);
    e->pkind = zalloc(sizeof(int32_t) * d); e->pa = zalloc(sizeof(double) * d); e->pb = zalloc(sizeof(double) * d);
    size_t tc = (size_t)cfg->trace_capacity;
    e->tX = zalloc(sizeof(double) * tc * nl * d); e->tlogp = zalloc(sizeof(double) * tc * nl);
    e->tmoved = zalloc(tc * nl); e->tsnk = zalloc(tc * nl);
    e->ttry = zalloc(sizeof(int32_t) * tc * nl); e->tcr = zalloc(sizeof(int32_t) * tc * nl);
    e->pts = zalloc(sizeof(double) * k * d); e->refs = zalloc(sizeof(double) * k * d); e->work = zalloc(sizeof(double) * 8 * d);
    {
        const size_t L1 = (size_t)cfg->adapt_lag + 1, nq = 2 + (size_t)cfg->ncr + cfg->ngamma;
        e->pend_tot = zalloc(sizeof(double) * L1 * nq * d); e->pend_cnt = zalloc(sizeof(double) * L1 * (cfg->ncr + cfg->ngamma));
        e->x0ring = zalloc(sizeof(double) * 2 * L1 * d); e->x0start = zalloc(sizeof(double) * d);
    }
    *out = e;
    return 0;
}

int orc_destroy(orc_engine* e)
{
    if (!e) return 0;
    free(e->mins); free(e->maxs); free(e->gtab); free(e->Z); free(e->X); free(e->lprior); free(e->llike);
    free(e->cp_prev); free(e->cp_new); free(e->cr_probs); free(e->cr_delta); free(e->cr_n);
    free(e->g_probs); free(e->g_delta); free(e->g_n); free(e->own_cr); free(e->own_g);
    free(e->pkind); free(e->pa); free(e->pb); free(e->mu); free(e->Mx); free(e->mixF);
    free(e->tX); free(e->tlogp); free(e->tmoved); free(e->tsnk); free(e->ttry); free(e->tcr);
    free(e->Tc); free(e->tswap);
    free(e->pend_tot); free(e->pend_cnt); free(e->x0ring); free(e->x0start);
    free(e->pts); free(e->refs); free(e->work); free(e);
    return 0;
}

int orc_set_bounds(orc_engine* e, const double* mins, const double* maxs)
{   /* Dream.py:86-105 */
    memcpy(e->mins, mins, sizeof(double) * e->d); memcpy(e->maxs, maxs, sizeof(double) * e->d); return 0;
}
int orc_set_gamma_table(orc_engine* e, const double* t)
{
    if (t) memcpy(e->gtab, t, sizeof(double) * e->c.ngamma * e->c.depairs * e->d);
    else orc_gamma_table(e->c.ngamma, e->c.depairs, e->d, e->gtab);
    return 0;
}
int orc_set_history(orc_engine* e, const double* Z, int64_t rows)
{   /* core.py:255-263, 281-283: seed rows first */
    if (rows > e->c.history_capacity) return fail("history exceeds capacity");
    memcpy(e->Z, Z, sizeof(double) * (size_t)rows * e->d); e->M = rows; e->napp = 0; return 0;
}
int orc_set_state(orc_engine* e, const double* X, const double* prior, const double* like)
{
    memcpy(e->X, X, sizeof(double) * e->nl * e->d);
    if (prior && like) { memcpy(e->lprior, prior, sizeof(double) * e->nl); memcpy(e->llike, like, sizeof(double) * e->nl); e->have_logp = 1; }
    else e->have_logp = 0;       /* evaluated at the first step, Dream.py:266-268 */
    return 0;
}
int orc_set_cr_probs(orc_engine* e, const double* p, int32_t n)
{   /* Dream.py:128-134 */
    if (n != e->c.ncr) return fail("nCR mismatch");
    memcpy(e->cr_probs, p, sizeof(double) * n);
    for (int c = 0; c < e->nl; ++c) memcpy(e->own_cr + (size_t)c * n, p, sizeof(double) * n);
    return 0;
}
int orc_set_gamma_probs(orc_engine* e, const double* p, int32_t n)
{   /* Dream.py:138-143 */
    if (n != e->c.ngamma) return fail("ngamma mismatch");
    memcpy(e->g_probs, p, sizeof(double) * n);
    for (int c = 0; c < e->nl; ++c) memcpy(e->own_g + (size_t)c * n, p, sizeof(double) * n);
    return 0;
}
int orc_set_prior(orc_engine* e, const int32_t* kind, const double* a, const double* b)
{
    memcpy(e->pkind, kind, sizeof(int32_t) * e->d); memcpy(e->pa, a, sizeof(double) * e->d); memcpy(e->pb, b, sizeof(double) * e->d);
    e->have_prior = 1; return 0;
}
int orc_set_likelihood_mvn(orc_engine* e, const double* mu, const double* M, int32_t kind, double log_F)
{
    int d = e->d;
    free(e->mu); free(e->Mx);
    e->mu = zalloc(sizeof(double) * d); e->Mx = zalloc(sizeof(double) * d * d);
    memcpy(e->mu, mu, sizeof(double) * d); memcpy(e->Mx, M, sizeof(double) * d * d);
    e->logF = log_F; e->lk = kind == 0 ? LK_MVN_DENSE : LK_MVN_TRI; return 0;
}
int orc_set_likelihood_mixture(orc_engine* e, int32_t J, const double* mu, const double* log_F)
{
    int d = e->d;
    free(e->mu); free(e->mixF);
    e->mu = zalloc(sizeof(double) * J * d); e->mixF = zalloc(sizeof(double) * J);
    memcpy(e->mu, mu, sizeof(double) * J * d); memcpy(e->mixF, log_F, sizeof(double) * J);
    e->J = J; e->lk = LK_MIX; return 0;
}
int orc_set_likelihood_host(orc_engine* e, orc_logp_cb cb, void* user) { e->cb = cb; e->cb_user = user; e->lk = LK_HOST; return 0; }
int orc_set_exchange(orc_engine* e, orc_exchange_cb cb, void* user) { e->xcb = cb; e->xcb_user = user; return 0; }
int orc_set_temperatures(orc_engine* e, const double* T /* [N] */, int32_t swaps)
{   /* core.py:133-136 ladder, handed over by the host; swaps != 0 enables the swap step */
    if (e->nl != e->N && swaps) return fail("temperature swaps need all chains in one engine");
    if (e->c.schedule != 2 && swaps) return fail("temperature swaps need schedule S2");
    if (!e->Tc) e->Tc = zalloc(sizeof(double) * e->N);
    memcpy(e->Tc, T, sizeof(double) * e->N);
    if (swaps && e->c.adapt_lag) return fail("adapt_lag is not available with temperature swaps");
    e->tempering = swaps != 0;
    if (e->tempering && !e->tswap) e->tswap = zalloc(sizeof(int32_t) * 3 * (size_t)(e->c.trace_capacity ? e->c.trace_capacity : 1));
    return 0;
}
int orc_get_swaps(orc_engine* e, int64_t g0, int64_t ng, int32_t* out /* [ng][3] */)
{
    if (!e->tswap || g0 < 0 || g0 + ng > e->ntrace) return fail("swap log range");
    memcpy(out, e->tswap + 3 * (size_t)g0, sizeof(int32_t) * 3 * (size_t)ng);
    return 0;
}
int64_t orc_generation(orc_engine* e) { return e->gen; }

/* ------------------------------------------------------------------ */
/* Log densities                                                       */
/* ------------------------------------------------------------------ */

/* Built-in target densities.
 * MVN: examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52
 *      logp = log_F - .5 * sum(x * dot(invC, x));  kind 0 takes invC itself,
 *      kind 1 an upper-triangular U with invC = U^T U (Q = |U v|^2).
 *      Order (contract v2): y_r = sum_c M[r][c] v_c ascending c (fma chain); Q = q_0 + q_1 + ... over row
 *      tiles of 16 in ascending order; inside tile t, with r = 16 t + kq + 4 e: four partial sums
 *      s_kq = fma chain over e = 0..3 of y_r s_r, then q_t = (s_0 + s_2) + (s_1 + s_3) -- the xor butterfly
 *      (2, 1) over kq.  s = v (dense) or y (triangular).  Rows r >= d contribute y_r s_r = 0.
 * Mixture: examples/mixturemodel/mixturemodel.py:37-48; the squared distances are
 *      summed in the lane/butterfly order of the reduction contract. */
HOT double orc_loglike(orc_engine* e, const double* x)
{
    int d = e->d;
    if (e->lk == LK_MVN_DENSE || e->lk == LK_MVN_TRI) {
        int tri = e->lk == LK_MVN_TRI;
        double* v = e->work + 6 * (size_t)d;
        for (int j = 0; j < d; ++j) v[j] = x[j] - e->mu[j];
        double Q = 0.0;
        for (int t0 = 0; t0 < d; t0 += 16) {                       /* row tiles of 16, ascending */
            double sk[4];
            for (int kq = 0; kq < 4; ++kq) {
                double sacc = 0.0;
                for (int el = 0; el < 4; ++el) {
                    const int r = t0 + kq + 4 * el;
                    if (r < d) {
                        const double* row = e->Mx + (size_t)r * d;
                        double y = 0.0;
                        for (int c = tri ? r : 0; c < d; ++c) y = fma(row[c], v[c], y);
                        sacc = fma(y, tri ? y : v[r], sacc);
                    }
                }
                sk[kq] = sacc;
            }
            Q = Q + ((sk[0] + sk[2]) + (sk[1] + sk[3]));
        }
        return e->logF - 0.5 * Q;
    }
    if (e->lk == LK_MIX) {
        double lh[64]; double mx = -INFINITY;
        for (int j = 0; j < e->J; ++j) {
            const double* mu = e->mu + (size_t)j * d;
            double* t = e->work + 6 * (size_t)d;
            for (int i = 0; i < d; ++i) t[i] = x[i] - mu[i];
            double S = orc_wave_dot(t, t, d);          /* lane/butterfly order */
            lh[j] = -0.5 * S + e->mixF[j];
            if (lh[j] > mx) mx = lh[j];
        }
        double dens = 0.0;
        for (int j = 0; j < e->J; ++j) dens = dens + orc_exp(lh[j] - mx);
        return orc_log(dens) + mx;
    }
    return 0.0;
}

/* Per-dimension priors: 0 flat (parameters.py:62-63), 1 scipy.stats.norm(loc=a,
 * scale=b), 2 scipy.stats.uniform(loc=a, scale=b); summed over dims
 * (parameters.py:37-47) in the lane/butterfly order. */
static double prior_logp(orc_engine* e, const double* x)
{
    if (!e->have_prior) return 0.0;
    int d = e->d; double* t = e->work + 7 * (size_t)d;
    for (int j = 0; j < d; ++j) {
        if (e->pkind[j] == 1) {
            /* scipy.stats.norm.logpdf (parameters.py:45): z = (x - loc) / scale, here by the reciprocal of the scale made ONCE
             * (contract, round 4: no IEEE division per try and dimension on the GPU; within 1 ulp of the quotient) */
            const double ib = 1.0 / e->pb[j];
            double z = (x[j] - e->pa[j]) * ib;
            t[j] = (-(z * z) / 2.0 - 0.91893853320467274178) - orc_log(e->pb[j]);
        } else if (e->pkind[j] == 2) {
            t[j] = (x[j] >= e->pa[j] && x[j] <= e->pa[j] + e->pb[j]) ? -orc_log(e->pb[j]) : -INFINITY;
        } else t[j] = 0.0;
    }
    return wave_sum(t, d);
}

/* Model.total_logp (model.py:17-32) over a batch; NaN is mapped to -inf. */
static int eval_points(orc_engine* e, const double* P, int n, double* prior, double* like)
{
    if (e->lk == LK_HOST) {
        int rc = e->cb(P, n, e->d, prior, like, e->cb_user);
        if (rc) return fail("host likelihood callback failed");
        if (e->have_prior) for (int i = 0; i < n; ++i) prior[i] = prior[i] + prior_logp(e, P + (size_t)i * e->d);   /* built-in priors add to the callback's */
    } else {
        for (int i = 0; i < n; ++i) { prior[i] = prior_logp(e, P + (size_t)i * e->d); like[i] = orc_loglike(e, P + (size_t)i * e->d); }
    }
    for (int i = 0; i < n; ++i) { if (prior[i] != prior[i]) prior[i] = -INFINITY; if (like[i] != like[i]) like[i] = -INFINITY; }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Proposal generation (Dream.py:670-796, 798-837)                     */
/* ------------------------------------------------------------------ */
typedef struct { double u_snk, u_cr, u_de, u_glev, u_sel, u_acc; } ctrl_draws;

static void draw_ctrl(uint64_t seed, uint32_t gc, uint32_t g, ctrl_draws* o)
{
    uint32_t w[4]; uint32_t s = orc_stream_id(K_CTRL, 0, 0, 0);
    orc_philox4x32_10(seed, 0, s, gc, g, w); o->u_snk = orc_u53(w[0], w[1]); o->u_cr = orc_u53(w[2], w[3]);
    orc_philox4x32_10(seed, 1, s, gc, g, w); o->u_de = orc_u53(w[0], w[1]); o->u_glev = orc_u53(w[2], w[3]);
    orc_philox4x32_10(seed, 2, s, gc, g, w); o->u_sel = orc_u53(w[0], w[1]); o->u_acc = orc_u53(w[2], w[3]);
}

/* generate_proposal_points(n, base, CR, DEpairs, gamma_level, snooker)
 * out: pts[n,d]; slogp[n] (snooker_logp, 0 for DE); gammas[n]; cur_norm_log =
 * (d-1) log|base - z_0| for the single-try snooker ratio (Dream.py:328-329). */
static int gen_points(orc_engine* e, uint32_t gc, uint32_t g, int phase, int n, const double* base, int64_t M,
                      int snk, int cr_idx, int delta, int glev,
                      double* pts, double* slogp, double* gammas, double* cur_snk_logp, int64_t* zidx)
{
    const int d = e->d; const uint64_t seed = e->c.seed;
    const double CR = (double)(cr_idx + 1) / (double)e->c.ncr;     /* Dream.py:146 */
    double *U = e->work, *e1 = e->work + d, *zt = e->work + 2 * (size_t)d, *v = e->work + 3 * (size_t)d,
           *dz = e->work + 4 * (size_t)d, *tmp = e->work + 5 * (size_t)d;
    uint32_t w[4];
    if (M > 0xffffffffll) return fail("history too long for 32-bit index draw");
    double gamma_snk = 0.0;
    if (snk) {   /* one set_gamma call per generate call, Dream.py:730 -> :615-618 (try 0's point stream) */
        orc_philox4x32_10(seed, 0, orc_stream_id(K_PT, 0, phase, 0), gc, g, w);
        double u_gs = orc_u53(w[2], w[3]);
        gamma_snk = 1.2 + (2.2 - 1.2) * u_gs;
    }
    for (int i = 0; i < n; ++i) {
        double* P = pts + (size_t)i * d;
        uint32_t s_pt = orc_stream_id(K_PT, i, phase, 0), s_dim = orc_stream_id(K_DIM, i, phase, 0), s_bnd = orc_stream_id(K_BND, i, phase, 0);
        if (!snk) {
            /* sample_from_history, Dream.py:646-668: 2*delta distinct rows */
            uint32_t words[16], rows[16];
            for (int q = 0; q * 4 < 2 * delta; ++q) orc_philox4x32_10(seed, 1 + q, s_pt, gc, g, words + 4 * q);
            if (orc_sample_distinct(words, 2 * delta, (uint32_t)M, rows)) return fail("history shorter than 2*DEpairs");
            if (zidx) for (int t = 0; t < 2 * delta; ++t) zidx[(size_t)i * 16 + t] = rows[t];
            /* chain_differences, Dream.py:692 (rows added in order) */
            for (int j = 0; j < d; ++j) {
                double a = e->Z[(size_t)rows[0] * d + j], b = e->Z[(size_t)rows[delta] * d + j];
                for (int t = 1; t < delta; ++t) { a = a + e->Z[(size_t)rows[t] * d + j]; b = b + e->Z[(size_t)rows[delta + t] * d + j]; }
                dz[j] = a - b;
            }
            /* zeta, e, U: Dream.py:694-700 */
            int dprime = 0;
            for (int j = 0; j < d; ++j) {
                /* one Philox call per PAIR of dimensions (2q, 2q+1): w0 = the two 16-bit crossover uniforms,
                 * w1 = the two 16-bit e uniforms, (w2,w3) = one Box-Muller pair (cos -> even, sin -> odd dim) */
                const int h = j & 1;
                float z0, z1;
                orc_philox4x32_10(seed, (uint32_t)(j >> 1), s_dim, gc, g, w);
                orc_normal32_pair(w[2], w[3], &z0, &z1);
                U[j] = orc_u16(w[0] >> (16 * h));
                e1[j] = orc_uniform16(w[1] >> (16 * h), -e->c.lamb, e->c.lamb) + 1.0;                     /* :696-697 */
                zt[j] = e->c.zeta * (double)(h ? z1 : z0);                                     /* :694 */
                if (U[j] < CR) dprime++;                                                       /* :704, :709 */
            }
            /* set_gamma, Dream.py:601-626 */
            orc_philox4x32_10(seed, 0, s_pt, gc, g, w);
            double u_gu = orc_u53(w[0], w[1]);
            double gamma;
            if (u_gu < e->c.p_gamma_unity) gamma = 1.0;
            else { int dp = dprime == 0 ? d : dprime;   /* index -1 wraps to the last entry, :624 */
                   gamma = e->gtab[((size_t)(glev - 1) * e->c.depairs + (delta - 1)) * d + (dp - 1)]; }
            gammas[i] = gamma; slogp[i] = 0.0;
            /* :714/:717 then crossover :720-726 */
            for (int j = 0; j < d; ++j) {
                double t = e1[j] * gamma; t = t * dz[j];
                double p = base[j] + t; p = p + zt[j];
                P[j] = (U[j] < CR) ? p : base[j];
            }
        } else {
            /* snooker_update, Dream.py:798-837 */
            orc_philox4x32_10(seed, 1, s_pt, gc, g, w);
            uint32_t iz = (uint32_t)(((uint64_t)w[0] * (uint64_t)M) >> 32);
            uint32_t i1 = (uint32_t)(((uint64_t)w[1] * (uint64_t)M) >> 32);
            uint32_t i2 = (uint32_t)(((uint64_t)w[2] * (uint64_t)M) >> 32);
            if (zidx) { zidx[(size_t)i * 16] = iz; zidx[(size_t)i * 16 + 1] = i1; zidx[(size_t)i * 16 + 2] = i2; }
            const double *z = e->Z + (size_t)iz * d, *r1 = e->Z + (size_t)i1 * d, *r2 = e->Z + (size_t)i2 * d;
            for (int j = 0; j < d; ++j) { v[j] = base[j] - z[j]; dz[j] = r1[j] - r2[j]; }      /* :813, :819 */
            double D = orc_wave_dot(v, v, d);                                                 /* :816 / :827 */
            if (n > 1) {
                double s = orc_wave_dot(dz, v, d);
                double c = s / D;
                for (int j = 0; j < d; ++j) { double zp = nan_to_num(c * v[j]); P[j] = base[j] + gamma_snk * zp; }   /* :820-822 */
            } else {
                double s = 0.0;
                if (D != 0.0) { for (int j = 0; j < d; ++j) tmp[j] = (dz[j] * v[j]) / D; s = wave_sum(tmp, d); }     /* :831 */
                s = nan_to_num(s);
                for (int j = 0; j < d; ++j) { double zp = s * v[j]; P[j] = base[j] + gamma_snk * zp; }
            }
            for (int j = 0; j < d; ++j) tmp[j] = P[j] - z[j];
            double norm = sqrt(orc_wave_dot(tmp, tmp, d));                                    /* :823 / :834 */
            slogp[i] = norm != 0.0 ? orc_log(norm) * (double)(d - 1) : 0.0;                   /* :824 / :835 */
            gammas[i] = gamma_snk;
            if (i == 0 && cur_snk_logp) { double nc = sqrt(D); *cur_snk_logp = nc != 0.0 ? orc_log(nc) * (double)(d - 1) : 0.0; }  /* :328-329 */
        }
        /* hard boundaries, Dream.py:733-791: reflect once, then redraw uniformly */
        if (e->c.hardboundaries) {
            for (int j = 0; j < d; ++j) {
                double x = P[j];
                int lo = x < e->mins[j], hi = x > e->maxs[j];
                if (lo) x = 2 * e->mins[j] - x;
                if (hi) x = 2 * e->maxs[j] - x;
                if (x < e->mins[j] || x > e->maxs[j]) {
                    orc_philox4x32_10(seed, (uint32_t)j, s_bnd, gc, g, w);
                    x = e->mins[j] + orc_u32(w[0]) * (e->maxs[j] - e->mins[j]);
                }
                P[j] = x;
            }
        }
    }
    return 0;
}

int orc_debug_propose(orc_engine* e, int32_t chain_local, int64_t gen, int phase, const double* base,
                      int run_snooker, int cr_idx, int delta, int glev,
                      double* pts, double* slogp, double* gammas, int64_t* zidx)
{
    int n = phase == 0 ? e->k : e->k - 1;
    return gen_points(e, (uint32_t)(e->c.chain_offset + chain_local), (uint32_t)gen, phase, n, base, e->M,
                      run_snooker, cr_idx, delta, glev, pts, slogp, gammas, NULL, zidx);
}

/* ------------------------------------------------------------------ */
/* One chain transition, Dream.py:238-347                              */
/* ------------------------------------------------------------------ */
typedef struct {
    int snk, cr_idx, delta, glev, sel, accept, moved, gamma_unity, redraws;
    double prior_new, like_new;
} step_res;

static int chain_step(orc_engine* e, int c, uint32_t g, int64_t M, const double* cr_probs, const double* g_probs,
                      double* xnew, step_res* R)
{
    const int d = e->d, k = e->k; const double T = e->Tc ? e->Tc[e->c.chain_offset + c] : e->c.temperature;
    const uint32_t gc = (uint32_t)(e->c.chain_offset + c);
    const double* q0 = e->X + (size_t)c * d;
    ctrl_draws u; draw_ctrl(e->c.seed, gc, g, &u);
    R->snk = (e->c.snooker != 0.0) && (u.u_snk < e->c.snooker);                   /* set_snooker :542-554 */
    R->cr_idx = orc_invcdf(cr_probs, e->c.ncr, u.u_cr);                            /* set_CR :556-569 */
    R->delta = e->c.depairs > 1 ? 1 + (int)floor(u.u_de * (double)e->c.depairs) : 1;   /* set_DEpair :571-583 */
    R->glev = 1 + orc_invcdf(g_probs, e->c.ngamma, u.u_glev);                      /* set_gamma_level :585-599 */
    double slp[64], slr[64], gam[64], cur_snk = 0.0;
    double pri[64], lik[64], rpri[64], rlik[64];
    R->redraws = 0;
    if (k > 63) return fail("multitry too large");
    if (gen_points(e, gc, g, 0, k, q0, M, R->snk, R->cr_idx, R->delta, R->glev, e->pts, slp, gam, &cur_snk, NULL)) return -1;   /* :258-264 */
    R->gamma_unity = 0;
    for (int i = 0; i < k; ++i) if (gam[i] == 1.0) R->gamma_unity = 1;
    if (eval_points(e, e->pts, k, pri, lik)) return -1;                                                             /* :270-279 */
    double last_logp = T * e->llike[c] + e->lprior[c];                                                              /* :243, :268 */
    double ratio; const double* qprop; int sel = 0;
    if (k == 1) {
        double q_logp = T * lik[0] + pri[0];                                                                        /* :274 */
        if (R->snk) ratio = nan_to_num((q_logp + slp[0]) - (last_logp + cur_snk));                                    /* :326-332 */
        else ratio = nan_to_num(q_logp) - nan_to_num(last_logp);                                                    /* :334 */
        qprop = e->pts;
    } else {
        double lp[64] = {0}; int anyfinite = 0;
        for (int i = 0; i < k; ++i) { lp[i] = pri[i] + T * lik[i]; if (isfinite(lp[i])) anyfinite = 1; }            /* :279, :900 */
        /* "all logps are -inf ... generate more proposal points" (:281-289): the whole proposal set is drawn again, same
         * decisions (snooker, CR, DE pairs, gamma level), until one try is finite.  Redraw round r >= 1 takes its point and
         * dimension streams from the Philox key seed + r * ORC_REDRAW_KEY_STEP; the reference's loop is
         * unbounded, here it gives up after ORC_MAX_REDRAWS rounds and the step is a forced reject (deviation D1). */
        for (int round = 1; !anyfinite && round <= ORC_MAX_REDRAWS; ++round) {
            const uint64_t seed0 = e->c.seed;
            e->c.seed = seed0 + (uint64_t)round * ORC_REDRAW_KEY_STEP;
            int grc = gen_points(e, gc, g, 0, k, q0, M, R->snk, R->cr_idx, R->delta, R->glev, e->pts, slp, gam, &cur_snk, NULL);
            e->c.seed = seed0;
            if (grc) return -1;
            if (eval_points(e, e->pts, k, pri, lik)) return -1;
            for (int i = 0; i < k; ++i) { lp[i] = pri[i] + T * lik[i]; if (isfinite(lp[i])) anyfinite = 1; }
            R->redraws = round;
        }
        /* mt_choose_proposal_pt :883-917 */
        double mx = lp[0]; for (int i = 1; i < k; ++i) if (lp[i] > mx) mx = lp[i];
        double wgt[64], S = 0.0;
        for (int i = 0; i < k; ++i) { wgt[i] = orc_exp(lp[i] - mx); S = S + wgt[i]; }
        for (int i = 0; i < k; ++i) wgt[i] = wgt[i] / S;
        sel = orc_invcdf(wgt, k, u.u_sel);
        qprop = e->pts + (size_t)sel * d;
        /* reference set :295-303 */
        if (gen_points(e, gc, g, 1, k - 1, qprop, M, R->snk, R->cr_idx, R->delta, R->glev, e->refs, slr, gam, NULL, NULL)) return -1;
        R->gamma_unity = 0;                                   /* self.gamma is overwritten by this call, :705/:730 */
        for (int i = 0; i < k - 1; ++i) if (gam[i] == 1.0) R->gamma_unity = 1;
        if (eval_points(e, e->refs, k - 1, rpri, rlik)) return -1;
        double A[64], B[64];
        for (int i = 0; i < k - 1; ++i) B[i] = T * rlik[i] + rpri[i];
        B[k - 1] = T * e->llike[c] + e->lprior[c];                                                                   /* :877-879, :303 */
        if (R->snk) {                                                                                                /* :306-313 */
            slr[k - 1] = 0.0;
            for (int i = 0; i < k; ++i) { A[i] = lp[i] + slp[i]; B[i] = (B[i] + slr[i]) + slp[i]; }
        } else for (int i = 0; i < k; ++i) A[i] = lp[i];
        double m2 = A[0]; for (int i = 0; i < k; ++i) { if (A[i] > m2) m2 = A[i]; if (B[i] > m2) m2 = B[i]; }          /* :320 */
        double SA = 0.0, SB = 0.0;
        for (int i = 0; i < k; ++i) SA = SA + orc_exp(A[i] - m2);                                                   /* :321 */
        for (int i = 0; i < k; ++i) SB = SB + orc_exp(B[i] - m2);                                                   /* :322 */
        ratio = nan_to_num(orc_log(SA / SB));                                                                       /* :323 */
        if (!anyfinite) ratio = -INFINITY;   /* DESIGN.md deviation D1: after ORC_MAX_REDRAWS redraw rounds the reference's unbounded loop (:282-289) ends in a forced reject */
    }
    /* metrop_select :980-998 */
    R->accept = isfinite(ratio) && (orc_log(u.u_acc) < ratio);
    R->sel = sel;
    R->moved = 0;
    if (R->accept) {
        memcpy(xnew, qprop, sizeof(double) * d);
        for (int j = 0; j < d; ++j) if (xnew[j] != q0[j]) R->moved = 1;
        R->prior_new = pri[sel]; R->like_new = lik[sel];                                                             /* :345-347 */
    } else {
        memcpy(xnew, q0, sizeof(double) * d);
        R->prior_new = e->lprior[c]; R->like_new = e->llike[c];
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Adaptation (Dream.py:451-499, 501-540)                              */
/* ------------------------------------------------------------------ */
/* population std per dimension of pos[N,d] (np.std axis=0, ddof=0, :476).
 * strip > 0: two-level sums (strips of `strip` rows, then strips in order);
 * strip == 0: plain row order (numpy's order; schedule S1). */
static void std_by_dim(const double* pos, int N, int d, int strip, double* sd)
{
    for (int j = 0; j < d; ++j) {
        double tot = 0.0;
        if (strip) { for (int s = 0; s < N; s += strip) { double ps = 0.0; for (int c = s; c < N && c < s + strip; ++c) ps = ps + pos[(size_t)c * d + j]; tot = tot + ps; } }
        else for (int c = 0; c < N; ++c) tot = tot + pos[(size_t)c * d + j];
        double mean = tot / (double)N;
        tot = 0.0;
        if (strip) { for (int s = 0; s < N; s += strip) { double ps = 0.0; for (int c = s; c < N && c < s + strip; ++c) { double t = pos[(size_t)c * d + j] - mean; ps = fma(t, t, ps); } tot = tot + ps; } }
        else for (int c = 0; c < N; ++c) { double t = pos[(size_t)c * d + j] - mean; tot = tot + t * t; }
        sd[j] = sqrt(tot / (double)N);
    }
}
/* sum(((q_new - q0)/sd)**2) (:481, :527) in the lane/butterfly order */
static double jump_norm(const double* xn, const double* xo, const double* sd, int d, double* tmp)
{
    for (int j = 0; j < d; ++j) tmp[j] = (xn[j] - xo[j]) / sd[j];
    return nan_to_num(orc_wave_dot(tmp, tmp, d));
}
static void renorm_probs(double* probs, const double* delta, const double* n, int nb, int N)
{   /* :487-493 / :531-536 */
    for (int m = 0; m < nb; ++m) if (delta[m] == 0.0) return;
    double S = 0.0;
    for (int m = 0; m < nb; ++m) { probs[m] = (delta[m] / n[m]) * (double)N; S = S + probs[m]; }
    for (int m = 0; m < nb; ++m) probs[m] = probs[m] / S;
}
static int adapt_cr_now(const orc_engine* e, uint32_t g, int gamma_unity)
{   /* :371 and :385/:395 */
    if (!e->c.adapt_crossover) return 0;
    if ((int64_t)g == e->c.crossover_burnin) return 1;
    return g > 10 && (int64_t)g < e->c.crossover_burnin && !gamma_unity;
}
static int adapt_gamma_now(const orc_engine* e, uint32_t g, int gamma_unity, int snk)
{   /* :381 and :385/:391 */
    if (!e->c.adapt_gamma) return 0;
    if ((int64_t)g == e->c.crossover_burnin) return 1;
    return g > 10 && (int64_t)g < e->c.crossover_burnin && !gamma_unity && !snk;
}

/* Adaptation of a LOCKSTEP generation (schedule S2), reduction contract v3.  The reference adds, chain by chain,
 * delta_m[m] += sum_j ((q_new - q0)_j / sd_j)^2 with sd = np.std(current_positions, axis=0) (:476-481, :522-527).  With every chain
 * of the generation updating at once the same quantity is  sum_j D[m][j] / sd_j^2,  D[m][j] = sum over the chains of bin m of
 * (q_new - q0)_j^2 -- column sums that do NOT need sd, so they are taken in the same pass over the positions as the sums sd itself
 * needs (one grid-wide reduction on the GPU instead of three: mean, deviations, bins):
 *   s_j   = cp_prev[chain 0][j]                          a shift near the population (any value is exact in exact arithmetic)
 *   S1_j  = sum_c (x_cj - s_j),  S2_j = sum_c (x_cj - s_j)^2 [fma],  Dc[m][j] / Dg[m][j] = sum_{c in bin m} (x_cj - xprev_cj)^2 [fma]
 *   every sum over chains: units of 16 consecutive global chains added in chain order from 0.0, groups of 16 units added in order,
 *   then the groups in order;
 *   a_j = S1_j / N,  var_j = fma(-a_j, a_j, S2_j / N) clamped at 0,  sd_j = sqrt(var_j);  crossover: sd_j == 0 -> 1e-12 (:479)
 *   w_j = 1 / (sd_j sd_j);  delta_m[m] += nan_to_num(sum_j Dc[m][j] w_j)  (lane/butterfly order, fma), ncr_updates[m] += the bin's count.
 * Differences from the chain-by-chain form: rounding (1e-15 relative), and nan_to_num applied once per bin instead of once per chain
 * -- visible only when a coordinate has population sd exactly 0 in the gamma statistic or a jump overflows. */
/* adapt_lag (schedule S2): the update a generation makes is the one above, from that generation's own positions,
 * jumps and bins, accumulated in generation order; what the lag changes is WHEN the chains see it -- generation g <= crossover_burnin
 * decides with the probabilities as they were after the updates of generations <= g - 1 - adapt_lag, and from the hand-over on
 * (g > crossover_burnin: the barrier of Dream.py:385-415, where every chain adopts the shared vector) with all of them.  So the
 * statistic is split: adapt_totals makes a generation's column totals and bin counts (nothing of the shared state is read),
 * adapt_apply adds them to delta_m / ncr_updates and renormalises (:481-499, :527-538).  The shift of generation g's column sums is
 * global chain 0's published position after generation g - 1 - adapt_lag (its start position before there is one): with lag 0 the
 * previous published position, as before. */
static void adapt_totals(orc_engine* e, const int* binc, const int* bing, const double* shift, double* tot /* [nq d] */, double* cnt /* [ncr + ngamma] */)
{
    const int N = e->N, d = e->d, ncr = e->c.ncr, ng = e->c.ngamma, nq = 2 + ncr + ng;
    for (int q = 0; q < nq; ++q)
        for (int j = 0; j < d; ++j) {
            double t = 0.0;
            for (int G = 0; G < N; G += 256) {
                double gs = 0.0;
                for (int u = G; u < N && u < G + 256; u += 16) {
                    double us = 0.0;
                    for (int c = u; c < N && c < u + 16; ++c) {
                        const double x = e->cp_new[(size_t)c * d + j];
                        if (q == 0) us = us + (x - shift[j]);
                        else if (q == 1) { const double v = x - shift[j]; us = fma(v, v, us); }
                        else {
                            const int m = q - 2, in = m < ncr ? binc[c] == m : bing[c] == m - ncr;
                            if (in) { const double df = x - e->cp_prev[(size_t)c * d + j]; us = fma(df, df, us); }
                        }
                    }
                    gs = gs + us;
                }
                t = t + gs;
            }
            tot[(size_t)q * d + j] = t;
        }
    for (int b = 0; b < ncr + ng; ++b) cnt[b] = 0.0;
    for (int c = 0; c < N; ++c) { if (binc[c] >= 0) cnt[binc[c]] += 1.0; if (bing[c] >= 0) cnt[ncr + bing[c]] += 1.0; }
}
static void adapt_apply(orc_engine* e, const double* tot, const double* cnt)
{
    const int N = e->N, d = e->d, ncr = e->c.ncr, ng = e->c.ngamma;
    double* wc = zalloc(sizeof(double) * d); double* wg = zalloc(sizeof(double) * d);
    for (int j = 0; j < d; ++j) {
        const double a = tot[j] / (double)N;
        double var = fma(-a, a, tot[(size_t)d + j] / (double)N);
        if (!(var > 0.0)) var = 0.0;
        const double sd = sqrt(var), sdc = sd == 0.0 ? 1e-12 : sd;      /* :479 (crossover only) */
        wc[j] = 1.0 / (sdc * sdc); wg[j] = 1.0 / (sd * sd);
    }
    int anyc = 0, anyg = 0;
    for (int m = 0; m < ncr; ++m) if (cnt[m] > 0.0) {
        e->cr_delta[m] = e->cr_delta[m] + nan_to_num(orc_wave_dot(tot + (size_t)(2 + m) * d, wc, d)); e->cr_n[m] += cnt[m]; anyc = 1;
    }
    for (int m = 0; m < ng; ++m) if (cnt[ncr + m] > 0.0) {
        e->g_delta[m] = e->g_delta[m] + nan_to_num(orc_wave_dot(tot + (size_t)(2 + ncr + m) * d, wg, d)); e->g_n[m] += cnt[ncr + m]; anyg = 1;
    }
    if (anyc) renorm_probs(e->cr_probs, e->cr_delta, e->cr_n, ncr, N);
    if (anyg) renorm_probs(e->g_probs, e->g_delta, e->g_n, ng, N);
    free(wc); free(wg);
}
/* end of generation g: its totals join the pending ones; those of generations <= g - adapt_lag (all of them at the hand-over,
 * g == crossover_burnin) are applied, oldest first -- the state generation g + 1 decides with */
static void adapt_lockstep(orc_engine* e, uint32_t g, const int* binc, const int* bing)
{
    const int d = e->d, L = e->c.adapt_lag, nb = e->c.ncr + e->c.ngamma, nq = 2 + nb, R = 2 * (L + 1);
    const int64_t hs = (int64_t)g - 1 - L;
    /* (lag 0: the previous published positions' row 0 itself -- after a temperature swap that row holds the swapped-in state, core.py:204-215) */
    const double* shift = L == 0 ? e->cp_prev : (hs < 0 ? e->x0start : e->x0ring + (size_t)(hs % R) * d);
    if (e->npend == 0) e->pend_g0 = (int64_t)g;
    const int slot = (int)(((int64_t)g - e->pend_g0));                   /* (npend <= L + 1: the oldest ones were applied below) */
    adapt_totals(e, binc, bing, shift, e->pend_tot + (size_t)slot * nq * d, e->pend_cnt + (size_t)slot * nb);
    e->npend = slot + 1;
    memcpy(e->x0ring + (size_t)(g % R) * d, e->cp_new, sizeof(double) * d);      /* global chain 0's position after generation g */
    const int64_t through = (int64_t)g == (int64_t)e->c.crossover_burnin ? (int64_t)g : (int64_t)g - L;
    int napply = 0;
    while (napply < e->npend && e->pend_g0 + napply <= through) {
        adapt_apply(e, e->pend_tot + (size_t)napply * nq * d, e->pend_cnt + (size_t)napply * nb);
        ++napply;
    }
    if (napply) {
        memmove(e->pend_tot, e->pend_tot + (size_t)napply * nq * d, sizeof(double) * (size_t)(e->npend - napply) * nq * d);
        memmove(e->pend_cnt, e->pend_cnt + (size_t)napply * nb, sizeof(double) * (size_t)(e->npend - napply) * nb);
        e->npend -= napply; e->pend_g0 += napply;
    }
}

/* chain flags of ANY global chain at generation g, recomputed from the random
 * contract (used for chains owned by other ranks in schedule S2). */
static void chain_flags(const orc_engine* e, uint32_t gc, uint32_t g, int* snk, int* cr_idx, int* glev, int* gamma_unity)
{
    ctrl_draws u; draw_ctrl(e->c.seed, gc, g, &u);
    *snk = (e->c.snooker != 0.0) && (u.u_snk < e->c.snooker);
    *cr_idx = orc_invcdf(e->cr_probs, e->c.ncr, u.u_cr);
    *glev = 1 + orc_invcdf(e->g_probs, e->c.ngamma, u.u_glev);
    int phase = e->k > 1 ? 1 : 0, n = e->k > 1 ? e->k - 1 : 1;
    *gamma_unity = 0;
    if (!*snk) for (int i = 0; i < n; ++i) {
        uint32_t w[4]; orc_philox4x32_10(e->c.seed, 0, orc_stream_id(K_PT, i, phase, 0), gc, g, w);
        if (orc_u53(w[0], w[1]) < e->c.p_gamma_unity) *gamma_unity = 1;
    }
}

static int exchange(orc_engine* e, const double* send_local, double* recv_global, size_t row_doubles)
{
    size_t bytes = sizeof(double) * (size_t)e->nl * row_doubles;
    if (e->nl == e->N) { memcpy(recv_global, send_local, bytes); return 0; }
    if (!e->xcb) return fail("sharded run needs an exchange callback");
    if (e->xcb(send_local, recv_global, (int64_t)bytes, e->xcb_user)) return fail("exchange callback failed");
    return 0;
}

static void record_trace(orc_engine* e, int c, const double* xnew, const step_res* R)
{   /* core.py:114-116 */
    if (e->c.trace_capacity == 0) return;
    size_t t = (size_t)e->ntrace, nl = (size_t)e->nl, d = (size_t)e->d;
    memcpy(e->tX + (t * nl + c) * d, xnew, sizeof(double) * d);
    {   /* core.py:115 (like + prior); with a temperature ladder core.py:178 (T*like + prior); 1.0*x == x */
        const double T = e->Tc ? e->Tc[e->c.chain_offset + c] : e->c.temperature;
        e->tlogp[t * nl + c] = T * R->like_new + R->prior_new;
    }
    e->tmoved[t * nl + c] = (uint8_t)R->moved; e->ttry[t * nl + c] = R->sel; e->tcr[t * nl + c] = R->cr_idx; e->tsnk[t * nl + c] = (uint8_t)R->snk;
}

static int first_logp(orc_engine* e)
{   /* Dream.py:266-268 */
    if (e->have_logp) return 0;
    for (int c = 0; c < e->nl; ++c) if (eval_points(e, e->X + (size_t)c * e->d, 1, e->lprior + c, e->llike + c)) return -1;
    e->have_logp = 1; return 0;
}

/* schedule S2: every chain of generation g sees the shared state as of the end
 * of generation g-1; end-of-generation updates in the order positions ->
 * adaptation -> history append. */
static int generation_s2(orc_engine* e)
{
    const int d = e->d, nl = e->nl, N = e->N; const uint32_t g = (uint32_t)e->gen;
    double* Xn = zalloc(sizeof(double) * nl * d); step_res* R = zalloc(sizeof(step_res) * nl);
    int rc = 0;
    if (g == 0 && (e->c.adapt_crossover || e->c.adapt_gamma)) {
        rc = exchange(e, e->X, e->cp_new, d);
        memcpy(e->x0start, e->cp_new, sizeof(double) * d); e->npend = 0;      /* global chain 0's start position */
    }
    /* history_lag: the rows of the last `lag` appends exist (their place in Z is fixed by the append order) but are not sampled yet */
    const int64_t lagged = e->napp < (int64_t)e->c.history_lag ? e->napp : (int64_t)e->c.history_lag;
    const int64_t Mvis = e->M - (int64_t)N * lagged;
    /* The chains of a generation are independent (schedule S2), so the CPU baseline may spread them over the host's
     * cores the way the reference spreads them over processes: each thread works on a shallow copy of the engine with
     * private scratch.  Results do not depend on the thread count.  The Python likelihood callback stays serial. */
#ifdef _OPENMP
    int nthr = omp_get_max_threads();
    if (nthr > nl / 16) nthr = nl / 16;              /* at least 16 chains per thread, or it is not worth a team */
    if (e->lk != LK_HOST && nthr > 1) {
        #pragma omp parallel num_threads(nthr)
        {
            orc_engine te = *e;
            te.pts = zalloc(sizeof(double) * e->k * d); te.refs = zalloc(sizeof(double) * e->k * d); te.work = zalloc(sizeof(double) * 8 * d);
            int trc = 0;
            #pragma omp for schedule(static)
            for (int c = 0; c < nl; ++c) if (!trc) trc = chain_step(&te, c, g, Mvis, e->cr_probs, e->g_probs, Xn + (size_t)c * d, &R[c]);
            if (trc) {
                #pragma omp critical
                rc = trc;
            }
            free(te.pts); free(te.refs); free(te.work);
        }
    } else
#endif
    for (int c = 0; c < nl && !rc; ++c) rc = chain_step(e, c, g, Mvis, e->cr_probs, e->g_probs, Xn + (size_t)c * d, &R[c]);
    if (rc) { free(Xn); free(R); return rc; }
    for (int c = 0; c < nl; ++c) {
        record_trace(e, c, Xn + (size_t)c * d, &R[c]);
        memcpy(e->X + (size_t)c * d, Xn + (size_t)c * d, sizeof(double) * d);
        e->lprior[c] = R[c].prior_new; e->llike[c] = R[c].like_new;
    }
    if (e->c.trace_capacity) e->ntrace++;
    /* set_current_position_arr :364-366 */
    if ((e->c.adapt_crossover || e->c.adapt_gamma) && (int64_t)g < (int64_t)e->c.crossover_burnin + 1) {
        double* t = e->cp_prev; e->cp_prev = e->cp_new; e->cp_new = t;
        rc = exchange(e, e->X, e->cp_new, d);
        if (rc) { free(Xn); free(R); return rc; }
        /* estimate_crossover_probabilities / estimate_gamma_level_probs for all N chains of the lockstep generation */
        int* binc = zalloc(sizeof(int) * N); int* bing = zalloc(sizeof(int) * N);
        for (int gcn = 0; gcn < N; ++gcn) {
            int snk, cr, gl, gu; chain_flags(e, (uint32_t)gcn, g, &snk, &cr, &gl, &gu);
            binc[gcn] = adapt_cr_now(e, g, gu) ? (snk ? e->c.ncr - 1 : cr) : -1;     /* :374-378 */
            bing[gcn] = adapt_gamma_now(e, g, gu, snk) ? gl - 1 : -1;
        }
        adapt_lockstep(e, g, binc, bing);
        free(binc); free(bing);
    }
    /* record_history :360-362, :919-938 */
    if (g % (uint32_t)e->c.history_thin == 0) {
        if (e->M + N > e->c.history_capacity) { free(Xn); free(R); return fail("history capacity exceeded"); }
        rc = exchange(e, e->X, e->Z + (size_t)e->M * d, d);
        e->M += N; e->napp += 1;
    }
    /* temperature swap (core.py:185-221): one random pair per generation, after every chain's step */
    if (e->tempering && !rc) {
        uint32_t w[4];
        orc_philox4x32_10(e->c.seed, 0u, orc_stream_id(4 /* K_SWAP */, 0, 0, 0), 0u, g, w);
        const uint32_t a = (uint32_t)(((uint64_t)w[0] * (uint64_t)N) >> 32);
        uint32_t b = (uint32_t)(((uint64_t)w[1] * (uint64_t)(N - 1)) >> 32);
        if (b >= a) b++;                                                        /* np.random.choice(nchains, 2, replace=False) :185 */
        const double u = orc_u53(w[2], w[3]);
        const double T1 = e->Tc[a], T2 = e->Tc[b], l1 = e->llike[a], l2 = e->llike[b];
        const double alpha = ((T1 * l2) + (T2 * l1)) - ((T1 * l1) + (T2 * l2));  /* :195 */
        const int acc = orc_log(u) < alpha;                                     /* :197 */
        if (acc) {
            for (int j = 0; j < d; ++j) { double t = e->X[(size_t)a * d + j]; e->X[(size_t)a * d + j] = e->X[(size_t)b * d + j]; e->X[(size_t)b * d + j] = t; }
            double t = e->llike[a]; e->llike[a] = e->llike[b]; e->llike[b] = t;
            t = e->lprior[a]; e->lprior[a] = e->lprior[b]; e->lprior[b] = t;
            /* the next generation's jumps are measured from the states astep is handed, i.e. after the swap
             * (Dream.py:371-378: q0 is the post-swap point, core.py:204-215): the published copy that serves as that
             * baseline follows the exchange; the column statistics of this generation were taken before it */
            if ((e->c.adapt_crossover || e->c.adapt_gamma) && (int64_t)g < (int64_t)e->c.crossover_burnin + 1)
                for (int j = 0; j < d; ++j) { double q = e->cp_new[(size_t)a * d + j]; e->cp_new[(size_t)a * d + j] = e->cp_new[(size_t)b * d + j]; e->cp_new[(size_t)b * d + j] = q; }
        }
        if (e->c.trace_capacity) { int32_t* q = e->tswap + 3 * (size_t)(e->ntrace - 1); q[0] = (int32_t)a; q[1] = (int32_t)b; q[2] = acc; }
    }
    free(Xn); free(R);
    e->gen++;
    return rc;
}

/* schedule S1: chains run to completion one after the other inside a
 * generation, exactly what the unmodified reference does when `astep` is
 * driven round-robin in one process. */
static int generation_s1(orc_engine* e)
{
    const int d = e->d, nl = e->nl, N = e->N; const uint32_t g = (uint32_t)e->gen;
    if (nl != N) return fail("schedule S1 is single-rank only");
    double* xn = zalloc(sizeof(double) * d); double* sd = zalloc(sizeof(double) * d); double* sdz = zalloc(sizeof(double) * d); double* tmp = zalloc(sizeof(double) * d);
    int rc = 0;
    for (int c = 0; c < nl && !rc; ++c) {
        step_res R; double* q0 = e->X + (size_t)c * d;
        double* ocr = e->own_cr + (size_t)c * e->c.ncr; double* og = e->own_g + (size_t)c * e->c.ngamma;
        if (g == 0) { memcpy(ocr, e->cr_probs, sizeof(double) * e->c.ncr); memcpy(og, e->g_probs, sizeof(double) * e->c.ngamma); }
        rc = chain_step(e, c, g, e->M, ocr, og, xn, &R);
        if (rc) break;
        record_trace(e, c, xn, &R);
        if (g % (uint32_t)e->c.history_thin == 0) {                                   /* :360-362 */
            if (e->M + 1 > e->c.history_capacity) { rc = fail("history capacity exceeded"); break; }
            memcpy(e->Z + (size_t)e->M * d, xn, sizeof(double) * d); e->M += 1;
        }
        if ((int64_t)g < (int64_t)e->c.crossover_burnin + 1) memcpy(e->cp_new + (size_t)c * d, xn, sizeof(double) * d);   /* :364-366 */
        if (adapt_cr_now(e, g, R.gamma_unity)) {                                      /* :371-378, :395-401 */
            int m = R.snk ? e->c.ncr - 1 : R.cr_idx;
            e->cr_n[m] += 1.0;
            std_by_dim(e->cp_new, N, d, 0, sd);
            for (int j = 0; j < d; ++j) if (sd[j] == 0.0) sd[j] = 1e-12;
            e->cr_delta[m] = e->cr_delta[m] + jump_norm(xn, q0, sd, d, tmp);
            renorm_probs(e->cr_probs, e->cr_delta, e->cr_n, e->c.ncr, N);
            memcpy(ocr, e->cr_probs, sizeof(double) * e->c.ncr);                      /* :497 */
        }
        if (adapt_gamma_now(e, g, R.gamma_unity, R.snk)) {                             /* :381-383, :391-393 */
            int m = R.glev - 1;
            std_by_dim(e->cp_new, N, d, 0, sdz);
            e->g_n[m] += 1.0;
            e->g_delta[m] = e->g_delta[m] + jump_norm(xn, q0, sdz, d, tmp);
            renorm_probs(e->g_probs, e->g_delta, e->g_n, e->c.ngamma, N);
            memcpy(og, e->g_probs, sizeof(double) * e->c.ngamma);
        }
        if ((int64_t)g == e->c.crossover_burnin) {                                     /* :409-415 */
            if (e->c.adapt_gamma) memcpy(og, e->g_probs, sizeof(double) * e->c.ngamma);
            if (e->c.adapt_crossover) memcpy(ocr, e->cr_probs, sizeof(double) * e->c.ncr);
        }
        memcpy(q0, xn, sizeof(double) * d);
        e->lprior[c] = R.prior_new; e->llike[c] = R.like_new;
    }
    if (!rc && e->c.trace_capacity) e->ntrace++;
    free(xn); free(sd); free(sdz); free(tmp);
    if (!rc) e->gen++;
    return rc;
}

int orc_step(orc_engine* e, int64_t generations)
{
    if (!e) return fail("null engine");
    if (e->lk == LK_NONE) return fail("no likelihood set");
    if (e->M < 2 * e->c.depairs) return fail("history not seeded");
    if (e->c.trace_capacity && e->ntrace + generations > e->c.trace_capacity) return fail("trace capacity exceeded");
    if (first_logp(e)) return -1;
    for (int64_t i = 0; i < generations; ++i) {
        int rc = e->c.schedule == 1 ? generation_s1(e) : generation_s2(e);
        if (rc) return rc;
    }
    return 0;
}
int orc_trace_reset(orc_engine* e) { e->ntrace = 0; return 0; }

int orc_get_state(orc_engine* e, double* X, double* prior, double* like)
{
    if (X) memcpy(X, e->X, sizeof(double) * e->nl * e->d);
    if (prior) memcpy(prior, e->lprior, sizeof(double) * e->nl);
    if (like) memcpy(like, e->llike, sizeof(double) * e->nl);
    return 0;
}
int orc_get_trace(orc_engine* e, int64_t g0, int64_t ng, double* X, double* logp, uint8_t* moved, int32_t* try_idx, int32_t* cr_idx, uint8_t* snooker)
{
    if (g0 < 0 || g0 + ng > e->ntrace) return fail("trace range");
    size_t nl = (size_t)e->nl, d = (size_t)e->d, o = (size_t)g0 * nl, n = (size_t)ng * nl;
    if (X) memcpy(X, e->tX + o * d, sizeof(double) * n * d);
    if (logp) memcpy(logp, e->tlogp + o, sizeof(double) * n);
    if (moved) memcpy(moved, e->tmoved + o, n);
    if (try_idx) memcpy(try_idx, e->ttry + o, sizeof(int32_t) * n);
    if (cr_idx) memcpy(cr_idx, e->tcr + o, sizeof(int32_t) * n);
    if (snooker) memcpy(snooker, e->tsnk + o, n);
    return 0;
}
int orc_get_history(orc_engine* e, double* Z, int64_t cap_rows, int64_t* rows)
{
    if (rows) *rows = e->M;
    if (Z) { if (cap_rows < e->M) return fail("buffer too small"); memcpy(Z, e->Z, sizeof(double) * (size_t)e->M * e->d); }
    return 0;
}
int orc_get_cr_state(orc_engine* e, double* probs, double* delta_m, double* n_updates)
{
    if (probs) memcpy(probs, e->cr_probs, sizeof(double) * e->c.ncr);
    if (delta_m) memcpy(delta_m, e->cr_delta, sizeof(double) * e->c.ncr);
    if (n_updates) memcpy(n_updates, e->cr_n, sizeof(double) * e->c.ncr);
    return 0;
}
int orc_get_gamma_state(orc_engine* e, double* probs, double* delta_m, double* n_updates)
{
    if (probs) memcpy(probs, e->g_probs, sizeof(double) * e->c.ngamma);
    if (delta_m) memcpy(delta_m, e->g_delta, sizeof(double) * e->c.ngamma);
    if (n_updates) memcpy(n_updates, e->g_n, sizeof(double) * e->c.ngamma);
    return 0;
}

/* Gelman_Rubin, convergence.py:3-20: second half of each chain, ddof=0
 * variances, var_est = W (1 - 1/n) + B with n the FULL length. */
int orc_gelman_rubin(const double* tr, int nchains, int nsamples, int d, double* rhat)
{
    int nb = nsamples / 2, n2 = nsamples - nb;
    double* means = zalloc(sizeof(double) * nchains);
    for (int j = 0; j < d; ++j) {
        /* sums over chains (W, the mean of the chain means, B): strips of 64 chains in chain order, then the strips in order -- what lets
         * the GPU give every strip its own thread (k_rhat) */
        double W = 0.0, Ws = 0.0;
        for (int c = 0; c < nchains; ++c) {
            const double* x = tr + ((size_t)c * nsamples + nb) * d + j;
            double s = 0.0; for (int t = 0; t < n2; ++t) s = s + x[(size_t)t * d];
            double mean = s / (double)n2; means[c] = mean;
            double v = 0.0; for (int t = 0; t < n2; ++t) { double q = x[(size_t)t * d] - mean; v = v + q * q; }
            Ws = Ws + v / (double)n2;
            if ((c & 63) == 63 || c == nchains - 1) { W = W + Ws; Ws = 0.0; }
        }
        W = W / (double)nchains;
        double mm = 0.0;
        for (int s = 0; s < nchains; s += 64) { double ps = 0.0; for (int c = s; c < nchains && c < s + 64; ++c) ps = ps + means[c]; mm = mm + ps; }
        mm = mm / (double)nchains;
        double B = 0.0;
        for (int s = 0; s < nchains; s += 64) { double ps = 0.0; for (int c = s; c < nchains && c < s + 64; ++c) { double q = means[c] - mm; ps = ps + q * q; } B = B + ps; }
        B = B / (double)nchains;
        double var_est = W * (1.0 - 1.0 / (double)nsamples) + B;
        rhat[j] = sqrt(var_est / W);
    }
    free(means);
    return 0;
}
int orc_get_rhat(orc_engine* e, double* rhat)
{
    int nl = e->nl, d = e->d, n = (int)e->ntrace;
    if (n < 2) return fail("need at least 2 traced generations");
    double* tr = zalloc(sizeof(double) * (size_t)nl * n * d);
    for (int t = 0; t < n; ++t) for (int c = 0; c < nl; ++c)
        memcpy(tr + ((size_t)c * n + t) * d, e->tX + ((size_t)t * nl + c) * d, sizeof(double) * d);
    int rc = orc_gelman_rubin(tr, nl, n, d, rhat);
    free(tr); return rc;
}
