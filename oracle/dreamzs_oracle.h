/*
 * dreamzs_oracle.h -- CPU restatement of PyDREAM's MT-DREAM(ZS) hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it, and only as the checker / timed CPU baseline.
 *
 * Parity status: PINNED.  The restatement is checked against outputs of the
 * reference itself (imported from /root/reference in the build container by
 * tests/golden/make_golden.py, driven by this file's counter-based random
 * contract through the numpy/random module proxies of SURVEY.md App. D.2);
 * the resulting vectors are committed under tests/golden/ and replayed by
 * tests/test_oracle_golden.py.  The reference's own known-answer tests
 * (pydream/tests/test_dream.py:68-76, :345-355, :357-395, :397-439) are
 * restated in tests/test_oracle_reference_kat.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).
 */
#ifndef DREAMZS_ORACLE_H
#define DREAMZS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same field layout as dz_config in include/dreamzs.h (declared separately on
 * purpose: the oracle shares no source with the engine). */
typedef struct orc_config {
    int32_t nchains;          /* global number of chains N                      */
    int32_t nchains_local;    /* chains owned by this instance                  */
    int32_t chain_offset;     /* global id of local chain 0                     */
    int32_t ndim;             /* d                                              */
    int32_t multitry;         /* k >= 1 (Dream.py:155-161)                      */
    int32_t depairs;          /* DEpairs; delta ~ U{1..depairs} (Dream.py:150)  */
    int32_t ncr;              /* nCR (Dream.py:108-113)                         */
    int32_t ngamma;           /* gamma_levels (Dream.py:120)                    */
    int32_t history_thin;     /* Dream.py:188, :360                             */
    int32_t crossover_burnin; /* Dream.py:122, core.py:299-300                  */
    int32_t adapt_crossover;  /* Dream.py:125                                   */
    int32_t adapt_gamma;      /* Dream.py:137                                   */
    int32_t hardboundaries;   /* Dream.py:80                                    */
    int32_t schedule;         /* 1 = S1 sequential round-robin, 2 = S2 lockstep */
    int32_t device;           /* unused by the oracle                           */
    int32_t history_lag;      /* schedule S2 only: appended rows become sampleable `history_lag` appends late (0 = at once) */
    int32_t adapt_lag;        /* schedule S2 only: generation g <= crossover_burnin decides with the crossover / gamma-level probabilities
                               * as they were after the updates of generations <= g - 1 - adapt_lag (0 = after every earlier one); from
                               * the hand-over on (g > crossover_burnin, Dream.py:385-415) with all of them */
    int32_t reserved0;        /* keeps the 64-bit fields aligned; must be 0 */
    int64_t history_capacity; /* rows Z can hold                                */
    int64_t trace_capacity;   /* generations the trace buffer can hold          */
    uint64_t seed;
    double lamb;              /* Dream.py:164                                   */
    double zeta;              /* Dream.py:165                                   */
    double snooker;           /* Dream.py:152                                   */
    double p_gamma_unity;     /* Dream.py:153                                   */
    double temperature;       /* T of astep (Dream.py:193); 1.0                 */
} orc_config;

typedef struct orc_engine orc_engine;

/* Redraw of a proposal set whose tries are all impossible (Dream.py:281-289): round r >= 1 uses the Philox key
 * seed + r * ORC_REDRAW_KEY_STEP (mod 2^64); after ORC_MAX_REDRAWS rounds the step is a forced reject. */
#define ORC_REDRAW_KEY_STEP 0x9E3779B97F4A7C15ull
#define ORC_MAX_REDRAWS 64

/* batch log-density callback: X is [n,d] row-major; fill prior[n], like[n]. */
typedef int (*orc_logp_cb)(const double* X, int64_t n, int32_t d, double* prior, double* like, void* user);
/* all-gather hook for rank-sharded runs: send = this rank's block, recv = all
 * ranks' blocks in rank order, bytes = size of one block. */
typedef int (*orc_exchange_cb)(const void* send, void* recv, int64_t bytes, void* user);

int         orc_version(void);
int         orc_threads(void);     /* OpenMP threads the S2 chain loop will use (OMP_NUM_THREADS; 1 without OpenMP) */
const char* orc_last_error(void);

int orc_create(const orc_config* cfg, orc_engine** out);
int orc_destroy(orc_engine* e);

int orc_set_bounds(orc_engine* e, const double* mins, const double* maxs);
int orc_set_gamma_table(orc_engine* e, const double* table /* [ngamma,depairs,d] or NULL = compute */);
int orc_set_history(orc_engine* e, const double* Z, int64_t rows);
int orc_set_state(orc_engine* e, const double* X, const double* prior, const double* like);
int orc_set_cr_probs(orc_engine* e, const double* p, int32_t ncr);
int orc_set_gamma_probs(orc_engine* e, const double* p, int32_t ng);
int orc_set_prior(orc_engine* e, const int32_t* kind, const double* a, const double* b);
int orc_set_likelihood_mvn(orc_engine* e, const double* mu, const double* M, int32_t kind, double log_F);
int orc_set_likelihood_mixture(orc_engine* e, int32_t J, const double* mu, const double* log_F);
int orc_set_likelihood_host(orc_engine* e, orc_logp_cb cb, void* user);
int orc_set_exchange(orc_engine* e, orc_exchange_cb cb, void* user);
int orc_set_temperatures(orc_engine* e, const double* T /* [nchains] */, int32_t swaps);   /* core.py:131-236 */
int orc_get_swaps(orc_engine* e, int64_t g0, int64_t ng, int32_t* out /* [ng][3]: a, b, accepted */);

int orc_step(orc_engine* e, int64_t generations);
int orc_trace_reset(orc_engine* e);

int orc_get_state(orc_engine* e, double* X, double* prior, double* like);
int orc_get_trace(orc_engine* e, int64_t g0, int64_t ng, double* X, double* logp,
                  uint8_t* moved, int32_t* try_idx, int32_t* cr_idx, uint8_t* snooker);
int orc_get_history(orc_engine* e, double* Z, int64_t cap_rows, int64_t* rows);
int orc_get_cr_state(orc_engine* e, double* probs, double* delta_m, double* n_updates);
int orc_get_gamma_state(orc_engine* e, double* probs, double* delta_m, double* n_updates);
int orc_get_rhat(orc_engine* e, double* rhat);
int64_t orc_generation(orc_engine* e);

/* building blocks exposed for the function-level fixtures */
void     orc_philox4x32_10(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
double   orc_u53(uint32_t hi, uint32_t lo);
double   orc_u32(uint32_t w);
float    orc_normal32(uint32_t w1, uint32_t w2);      /* cosine branch of the Box-Muller pair */
float    orc_normal32_sin(uint32_t w1, uint32_t w2);  /* sine branch */
void     orc_normal32_fill(uint64_t seed, int64_t npairs, float* out /* [2 npairs] */);
double   orc_u16(uint32_t h);
double   orc_uniform16(uint32_t h, double low, double high);   /* uniform(low, high) from a 16-bit draw, one fma */
double   orc_exp(double x);
double   orc_log(double x);
uint32_t orc_stream_id(int kind, int tr, int phase, int round);
void     orc_gamma_table(int ngamma, int depairs, int d, double* out);
double   orc_wave_dot(const double* a, const double* b, int d);
int      orc_sample_distinct(const uint32_t* words, int n, uint32_t M, uint32_t* out);
int      orc_invcdf(const double* p, int n, double u);
double   orc_loglike(orc_engine* e, const double* x);
int      orc_gelman_rubin(const double* traces /* [nchains,nsamples,d] */, int nchains, int nsamples, int d, double* rhat);

/* one chain's proposal set, for fixtures: phase 0 (k tries from base) or 1 (k-1) */
int orc_debug_propose(orc_engine* e, int32_t chain_local, int64_t gen, int phase, const double* base,
                      int run_snooker, int cr_idx, int delta, int glev,
                      double* pts, double* slogp, double* gammas, int64_t* zidx);

#ifdef __cplusplus
}
#endif
#endif
