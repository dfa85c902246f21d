/*
 * dreamzs.h -- C ABI of libdreamzs.so, the MI355X (gfx950) MT-DREAM(ZS) engine.
 *
 * The reference (LoLab-MSM/PyDREAM) is pure Python and has no FFI of its own: its
 * boundary for the hot path is the Python API `pydream.core.run_dream()` /
 * `pydream.Dream.Dream.astep()` (pydream/core.py:11-86, pydream/Dream.py:193-422).
 * This header is the interface a maintainer binds with ctypes underneath that API
 * (INTEGRATION.md shows the stub); each entry point names the reference code it
 * replaces.  Conventions: every function returns 0 on success and a negative code on
 * error (text via dz_last_error(), thread-local); nothing throws across the boundary;
 * host pointers are borrowed for the duration of the call; one handle per GPU; calls
 * on one handle are not thread-safe, distinct handles are.  All matrices are
 * row-major float64.
 */
#ifndef DREAMZS_H
#define DREAMZS_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DZ_VERSION 1

/* Sampler configuration: the keyword arguments of Dream.__init__ (Dream.py:63-67) that
 * act on the hot path, plus the sizes `_setup_mp_dream_pool` derives (core.py:250-314). */
typedef struct dz_config {
    int32_t nchains;          /* global number of chains N (run_dream nchains, core.py:11)     */
    int32_t nchains_local;    /* chains owned by this handle (= N on one GPU)                  */
    int32_t chain_offset;     /* global id of local chain 0 (rank * nchains_local)             */
    int32_t ndim;             /* total_var_dimension (Dream.py:81-83)                          */
    int32_t multitry;         /* 1, or >= 3 (Dream.py:155-161; 2 is broken upstream, :867)     */
    int32_t depairs;          /* DEpairs (Dream.py:150)                                        */
    int32_t ncr;              /* nCR (Dream.py:108-113)                                        */
    int32_t ngamma;           /* gamma_levels (Dream.py:120)                                   */
    int32_t history_thin;     /* Dream.py:188, :360                                            */
    int32_t crossover_burnin; /* Dream.py:122; default niterations/10 (core.py:299-300)        */
    int32_t adapt_crossover;  /* Dream.py:125                                                  */
    int32_t adapt_gamma;      /* Dream.py:137                                                  */
    int32_t hardboundaries;   /* Dream.py:80                                                   */
    int32_t schedule;         /* must be 2 (lockstep generations, DESIGN.md "Schedule")        */
    int32_t device;           /* HIP device ordinal                                            */
    int32_t history_lag;      /* 0: the rows a generation appends are sampleable from the next generation on (the schedule the
                               * reference has when driven in lockstep).  L >= 1: they become sampleable L appends later, i.e. the
                               * generations between two appends sample from the archive as it was L appends ago -- which lets the
                               * exchange of appended rows between GPUs run behind the next thin-cycle's generations instead of
                               * between two launches (the reference's own chains see each other's appends with
                               * an arbitrary, scheduler-dependent delay, Dream.py:646-668 under core.py:80), and on ONE GPU lets a
                               * launch of the persistent kernels run on past its appends: (L + 1) history_thin generations per
                               * launch instead of history_thin */
    int32_t adapt_lag;        /* 0: generation g of the crossover burn-in decides with the crossover / gamma-level probabilities as every
                               * earlier generation's update left them (estimate_crossover_probabilities, Dream.py:451-499, in lockstep).
                               * L >= 1: with the probabilities as they were after the updates of generations <= g - 1 - L; the update a
                               * generation makes is unchanged (its own positions, jumps and bins, accumulated in generation order), and
                               * from the hand-over on (g > crossover_burnin: the barrier of Dream.py:385-415, where every chain adopts
                               * the shared vector) every update is in.  The reference's own chains see each other's updates with a
                               * scheduler-dependent delay (Dream.py:371-378 under core.py:80); a fixed L lets one launch of the
                               * persistent kernels hold L + 1 burn-in generations instead of one -- on one
                               * GPU, every configuration the persistent kernels run (mixture or MVN likelihood, d <= 256; burn-in 1.3x to
                               * 2.4x faster at L = 19).  Several GPUs: ranks that own whole groups of 256 chains do the same (their
                               * groups' sums of a whole launch travel in one exchange).  The multi-kernel path (d > 256, host likelihoods) and
                               * other shards keep one generation per launch under the same schedule (no gain, 6-12 % slower: leave it 0) */
    int32_t reserved0;        /* must be 0 (keeps the 64-bit fields aligned)                   */
    int64_t history_capacity; /* rows the Z archive can hold (core.py:260-268)                 */
    int64_t trace_capacity;   /* generations the device trace buffer holds (0 = no trace)      */
    uint64_t seed;            /* key of the counter-based random contract                      */
    double lamb;              /* Dream.py:164                                                  */
    double zeta;              /* Dream.py:165                                                  */
    double snooker;           /* Dream.py:152                                                  */
    double p_gamma_unity;     /* Dream.py:153                                                  */
    double temperature;       /* T of astep (Dream.py:193); 1.0                                */
} dz_config;

typedef struct dz_engine dz_engine;

/* Batch replacement for Model.total_logp (model.py:17-32) evaluated on the host:
 * X is [n,d]; fill prior[n] and like[n]; return 0. */
typedef int (*dz_logp_cb)(const double* X, int64_t n, int32_t d, double* prior, double* like, void* user);
/* Host-staged all-gather used when no RCCL communicator is attached (tests, gloo):
 * send = this rank's block, recv = all blocks in rank order, bytes = one block. */
typedef int (*dz_exchange_cb)(const void* send, void* recv, int64_t bytes, void* user);

int         dz_version(void);
const char* dz_last_error(void);
int         dz_device_count(int32_t* count);

/* Dream.__init__ + _setup_mp_dream_pool + _mp_dream_init: allocate the Z archive, chain
 * states and adaptation accumulators in HBM (replaces core.py:281-297, 316-327). */
int dz_create(const dz_config* cfg, dz_engine** out);
int dz_destroy(dz_engine* e);

int dz_set_bounds(dz_engine* e, const double* mins, const double* maxs);            /* Dream.py:86-105; [d] each, +-inf ok */
int dz_set_gamma_table(dz_engine* e, const double* table);                          /* Dream.py:172-179; [ngamma,depairs,d]; NULL = compute */
int dz_set_history(dz_engine* e, const double* Z, int64_t rows);                    /* core.py:255-263, 282-283: seed rows */
int dz_set_state(dz_engine* e, const double* X, const double* prior, const double* like); /* start points (core.py:74-78); NULL logps => evaluated at first step (Dream.py:266-268) */
int dz_set_cr_probs(dz_engine* e, const double* p, int32_t ncr);                    /* Dream.py:128-134 */
int dz_set_gamma_probs(dz_engine* e, const double* p, int32_t ngamma);              /* Dream.py:138-143 */
/* per-dimension priors evaluated on the device (parameters.py:37-47, 62-63):
 * kind 0 flat, 1 scipy.stats.norm(loc=a, scale=b), 2 scipy.stats.uniform(loc=a, scale=b) */
int dz_set_prior(dz_engine* e, const int32_t* kind, const double* a, const double* b);
/* built-in device likelihoods ("batched device callback"):
 * MVN, examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52: kind 0 = precision matrix
 * invC [d,d]; kind 1 = upper-triangular U with invC = U^T U.  logp = log_F - Q/2. */
int dz_set_likelihood_mvn(dz_engine* e, const double* mu, const double* M, int32_t kind, double log_F);
/* Gaussian mixture with identity covariances, examples/mixturemodel/mixturemodel.py:37-48 */
int dz_set_likelihood_mixture(dz_engine* e, int32_t J, const double* mu, const double* log_F);
/* arbitrary host likelihood (any Python callable behind ctypes) */
int dz_set_likelihood_host(dz_engine* e, dz_logp_cb cb, void* user);
/* arbitrary DEVICE likelihood -- the "batched device callback" for a model that is not one of the two descriptors above (the reference takes
 * any callable: model.py:17-32 `self.likelihood(q0)`).  The caller builds a gfx950 code object (hipcc --offload-arch=gfx950 --genco) that
 * exports
 *     extern "C" __global__ void NAME(const double* X, long long n, int d, int ld, double* like, const void* data);
 * X: the batch's points, row i at X + i * ld (row-major, rows padded to ld doubles); like[i] receives log L(X_i) (may be -inf; nan is
 * treated as -inf); data: a copy on the device of the `data_bytes` bytes at `data` (model constants, observations; NULL if none).  The
 * engine launches it once per batch where its own k_logp_* kernels run, with 256 threads per block and
 *     lanes_per_point = 1:  thread blockIdx.x * 256 + threadIdx.x evaluates point i (grid = ceil(n / 256));
 *     lanes_per_point = 64: wave (blockIdx.x * 4 + threadIdx.x / 64) evaluates point i with its 64 lanes (grid = ceil(n / 4)).
 * Priors given by dz_set_prior are added by the engine.  Generations with such a likelihood run the multi-kernel path -- unless the code
 * object also exports the persistent generation kernel instantiated around the same density (round 6):
 *     dz_user_generations_v<N>, dz_user_generations_full_v<N>      (N = DZ_USER_ABI of csrc/dz_kernels.h: the layout generation of the
 *     dz_user_generations_wide_v<N>, ..._wide_full_v<N>             structures they take; other names are simply not found; wide: 128 < d <= 256)
 * = csrc/dz_megakernel.h generations_wave_body<false | true, false, UserLike> -- the kernel the built-in mixture runs in, one wave per chain,
 * the user's wave-level device function `double f(const double* x, int d, const void* data, int lane)` in the likelihood's place
 * (pydream_amd.likelihoods.DeviceFunctionLogLike writes and compiles that translation unit at run time, against the headers next to this
 * library).  They carry the generations where the mixture's would (d <= 256, multitry 1 or 3..32, DZ_LIKE_ALWAYS_FINITE): same decisions,
 * same bits as through the batch kernel, at the persistent kernels' rate (DZ_MEGA_USER=0 in the environment: never).  flags: DZ_LIKE_ALWAYS_FINITE promises that the density is finite wherever the priors are, so
 * the engine need not check every proposal set for "all tries impossible" (Dream.py:281-289) with a read-back per generation. */
#define DZ_LIKE_ALWAYS_FINITE 1
int dz_set_likelihood_module(dz_engine* e, const char* code_object_path, const char* kernel_name, int32_t lanes_per_point, int32_t flags,
                             const void* data, int64_t data_bytes);

/* Multi-GPU: chains are sharded, Z / positions are replicated by an all-gather at the
 * end of appending generations (replaces the multiprocessing shared arrays,
 * Dream.py:919-938, 424-449).  Attach ONE of the two transports before dz_step. */
const char* dz_hip_library(void);      /* path of the HIP runtime this library's calls are bound to */
const char* dz_comm_library(void);     /* path of the librccl in use: the one next to that HIP runtime (and checked to resolve the same one) */
int dz_comm_unique_id(void* id128);                                                 /* rank 0: 128-byte RCCL unique id */
int dz_comm_init_rccl(dz_engine* e, int32_t rank, int32_t world, const void* id128);
int dz_comm_count(dz_engine* e, int32_t* ranks);       /* ncclCommCount of the engine's communicator (0: none): what RCCL itself says about the ranks taking part */
int dz_set_exchange(dz_engine* e, dz_exchange_cb cb, void* user);
/* Third transport, "peer": every rank maps the other ranks' archive (and published-position buffers) through HIP IPC and its COPY
 * ENGINES push the rank's rows into them on streams of their own, each push followed by the rank's flag word -- no compute unit is
 * involved, so the transfer runs while the next persistent launch (which owns every CU's LDS) computes.  A one-wave gate kernel in front
 * of the first launch that samples the new rows waits for the flags.  With dz_config.history_lag = 0 that is right behind the append
 * (replaces the shared arrays of core.py:281-297 / Dream.py:919-945 like the all-gather does); with history_lag = 1 a whole thin-cycle
 * later, i.e. the exchange is hidden.  Bootstrap: dz_peer_export fills this rank's blob, the control plane all-gathers the blobs (rank
 * order), dz_peer_attach maps them and runs a self-test (every rank pushes one flag word to every peer and waits for theirs: a refusal at
 * attach time instead of a timeout in the middle of a run).  A gate that has waited DZ_PEER_TIMEOUT_S seconds (environment; default 600) for a
 * rank gives up: dz_step queues nothing more and every synchronising call fails naming the silent rank.  dz_exchange_stats (after dz_sync): exchanges queued, gates passed and the time the gates spent
 * waiting -- the exposed part of the exchange. */
#define DZ_PEER_BLOB_BYTES 512
int dz_peer_export(dz_engine* e, void* blob /* DZ_PEER_BLOB_BYTES */);
int dz_peer_attach(dz_engine* e, int32_t rank, int32_t world, const void* blobs /* world x DZ_PEER_BLOB_BYTES */);
int dz_peer_detach(dz_engine* e);      /* stop using it (e.g. another rank could not attach and all ranks fall back to the same other transport) */
int dz_exchange_stats(dz_engine* e, int64_t* exchanges, int64_t* gates, double* gate_wait_us);
/* Bytes this rank has handed to the transport for EACH other rank so far (any transport), by what they were: appended history rows
 * (record_history, Dream.py:919-945), published positions (set_current_position_arr, Dream.py:424-449) and -- ranks that own whole groups
 * of 256 chains -- the crossover burn-in's group sums (estimate_crossover_probabilities, Dream.py:451-499: (2 + nCR + ngamma) ld + 16
 * doubles per group and generation instead of the group's 256 position rows; then the positions travel only once, at generation 0). */
int dz_exchange_bytes(dz_engine* e, int64_t* history_bytes, int64_t* position_bytes, int64_t* sums_bytes);
int dz_comm_barrier(dz_engine* e);     /* device-side rendezvous of the ranks (RCCL: a one-element all-gather; peer transport: every rank pushes a flag word to every peer and a gate kernel waits for theirs; then a stream sync) -- the ranks leave it within microseconds of each other, which a timed region of a few hundred microseconds wants in front of it; no transport: a sync */

/* Parallel tempering (core.py:131-236).  T[nchains]: the temperature of every (global) chain -- Dream.astep's T argument
 * (Dream.py:193), the ladder of core.py:133-136 is computed by the caller.  swaps != 0 adds the swap step of
 * core.py:185-221 after every generation (all chains on one GPU); dz_get_swaps returns, per traced generation, the pair
 * that was drawn and whether the swap was accepted. */
int dz_set_temperatures(dz_engine* e, const double* T, int32_t swaps);
int dz_get_swaps(dz_engine* e, int64_t g0, int64_t ng, int32_t* out /* [ng][3]: chain a, chain b, accepted */);

/* _sample_dream's loop (core.py:103-116): advance every local chain by `generations`
 * MT-DREAM(ZS) transitions (Dream.astep, Dream.py:193-422). Asynchronous on the engine's
 * stream; dz_sync() or any getter waits. */
int     dz_step(dz_engine* e, int64_t generations);
/* Dream.astep for ONE chain (or a contiguous range): a transition whose end-of-step updates
 * (history append, published position, crossover statistics) take effect immediately, i.e. the
 * reference's semantics when astep is driven round-robin in one process (test_dream.py:507-518). */
int     dz_step_range(dz_engine* e, int32_t chain0, int32_t nchains);
int     dz_set_chain_state(dz_engine* e, int32_t chain, const double* x, const double* prior, const double* like); /* NULL logps => evaluate */
int     dz_get_chain_state(dz_engine* e, int32_t chain, double* x, double* prior, double* like);
/* The crossover / gamma-level probabilities the Dream instance driving `chain` decides with (Dream.CR_probabilities, Dream.py:134, :375;
 * Dream.gamma_probabilities, :143, :383): under dz_step_range every chain keeps its own copy, refreshed by its own adaptation updates and
 * at the end of the burn-in (:409-415); before any single-chain step, and under dz_step, the shared vectors.  [ncr], [ngamma]; NULL skips. */
int     dz_get_chain_probs(dz_engine* e, int32_t chain, double* cr_probs, double* gamma_probs);
/* run_dream(restart=True) on the LIVE engine (core.py:46-62, 255-263; Dream.py:128-147 load the history and the adapted probabilities from
 * the files of the previous run -- here they are still in HBM): the archive as it stands becomes the new run's seed history, the
 * crossover / gamma-level probabilities stay, their accumulators (delta_m, ncr_updates; core.py:287-293) restart from zero, the generation
 * counter restarts at 0 (generation 0 appends, the crossover burn-in of `crossover_burnin` generations runs again), the random contract
 * takes `seed`.  The archive and the trace grow to the given capacities (device-to-device copy, no upload).  Follow with dz_set_state.
 * Identical, bit for bit, to a new engine given the downloaded archive and probabilities.  Not for sharded or tempering engines. */
int     dz_continue_run(dz_engine* e, int64_t history_capacity, int64_t trace_capacity, uint64_t seed, int32_t crossover_burnin);
int     dz_sync(dz_engine* e);
int     dz_trace_reset(dz_engine* e);
int64_t dz_generation(dz_engine* e);
/* Dream.py:281-289: a multi-try proposal set whose tries are all impossible (log density -inf or nan) is generated again, with the
 * same decisions, until one try is finite.  Redraw round r >= 1 takes its draws from the Philox key seed + r * DZ_REDRAW_KEY_STEP
 * (mod 2^64); the reference's loop is unbounded, the engine gives up after DZ_MAX_REDRAWS rounds (the step is then a rejection).
 * dz_redraw_rounds: number of redraw launches since dz_create (0 for targets and priors that cannot produce such a set). */
#define DZ_REDRAW_KEY_STEP 0x9E3779B97F4A7C15ull
#define DZ_MAX_REDRAWS 64
int64_t dz_redraw_rounds(dz_engine* e);
/* What the most recent generation(s) ran as: the template instantiation of the persistent kernel, e.g.
 * "k_generations<7,tri,xlds,16,1,lean>" (row tiles, matrix form, chain states in LDS or HBM, chains per block, waves per chain,
 * lean / full proposal code [, k1 = multitry off]), "k_generations_mix", or "multi-kernel path".  Lets a parity test assert that the
 * instantiation a benchmark times is the one it compared with the oracle.  The string lives until the next dz_step on the handle. */
const char* dz_last_kernel_variant(dz_engine* e);

int dz_get_state(dz_engine* e, double* X, double* prior, double* like);             /* [nl,d],[nl],[nl] */
/* trace of generations [g0,g0+ng) since the last reset: sampled_params / log_ps of
 * core.py:98-116 plus the decision sequences the parity contract is stated on */
int dz_get_trace(dz_engine* e, int64_t g0, int64_t ng, double* X, double* logp,
                 uint8_t* moved, int32_t* try_idx, int32_t* cr_idx, uint8_t* snooker);
/* the same samples chain by chain, the shape run_dream returns (core.py:98, :127: one (niterations, d) array per
 * chain): sample g0+i of local chain c goes to X[(c*chain_stride_rows + i)*d ...].  The device keeps the trace in
 * this order, so a whole-buffer request is a single strided copy. */
int dz_get_trace_chains(dz_engine* e, int64_t g0, int64_t ng, double* X, int64_t chain_stride_rows, double* logp /* [nl][chain_stride_rows] or NULL */);
/* the same for X without stopping the engine: the copy is queued behind the generations stepped so far, the call returns, and
 * generations stepped afterwards run while the rows leave (replaces nothing in the reference: its chains write their samples into
 * host arrays as they go, core.py:114).  X must be page-locked (dz_host_register) and untouched until dz_trace_download_wait. */
int dz_trace_download_begin(dz_engine* e, int64_t g0, int64_t ng, double* X, int64_t chain_stride_rows);
int dz_trace_download_wait(dz_engine* e);
/* optional: page-lock the destination beforehand (e.g. from a second thread while dz_step runs) */
int dz_host_register(void* ptr, int64_t bytes);
int dz_host_unregister(void* ptr);
int dz_get_history(dz_engine* e, double* Z, int64_t cap_rows, int64_t* rows);       /* Dream_shared_vars.history / count */
int dz_get_history_range(dz_engine* e, int64_t row0, int64_t nrows, double* Z /* [nrows, d] */);   /* rows [row0, row0 + nrows) only */
/* A 64-bit checksum of the archive as dz_get_history would return it ([rows, d], every appended row of every rank waited for), made on the
 * device: the sum modulo 2^64 over all elements of mix64(bits(Z[r][j]) + (r d + j + 1) 0x9E3779B97F4A7C15), mix64 = the splitmix64 finaliser.
 * What the ranks of a sharded run compare afterwards: the reference's chains all see ONE history (the shared array of core.py:281-283),
 * here every GPU holds a replica, and a replica that missed or misplaced a row would still produce a plausible chain. */
int dz_history_checksum(dz_engine* e, uint64_t* sum, int64_t* rows);
int dz_get_cr_state(dz_engine* e, double* probs, double* delta_m, double* n_updates);     /* cross_probs, delta_m, ncr_updates */
int dz_get_gamma_state(dz_engine* e, double* probs, double* delta_m, double* n_updates);  /* gamma_level_probs, ... */
/* Gelman_Rubin (convergence.py:3-20) over the traced generations of the local chains */
int dz_get_rhat(dz_engine* e, double* rhat);
/* per-chain second-half mean / population variance [nl,d] each (inputs of R-hat, for multi-rank reduction) */
int dz_get_chain_moments(dz_engine* e, double* mean, double* var);

/* evaluate the configured log densities on arbitrary points (testing / first-step logp) */
int dz_eval_logp(dz_engine* e, const double* X, int64_t n, double* prior, double* like);
/* one proposal set of one chain, for parity tests (generate_proposal_points, Dream.py:670-796) */
int dz_debug_propose(dz_engine* e, int32_t chain_local, int64_t gen, int32_t phase, const double* base,
                     int32_t run_snooker, int32_t cr_idx, int32_t delta, int32_t glev,
                     double* pts, double* slogp);

/* HIP-event timing of the engine's kernels over the launches since the last reset.
 * which: 0 propose, 1 logp, 2 accept, 3 adapt, 4 exchange, 5 generations (the persistent
 * multi-generation kernel), 6 empty (64 event pairs with no launch in between, recorded by
 * dz_profile_reset: the time one bracket adds to every measured launch).  Returns total ms and count. */
int dz_profile_enable(dz_engine* e, int32_t on);
int dz_profile_get(dz_engine* e, int32_t which, double* total_ms, int64_t* launches);
int dz_profile_reset(dz_engine* e);
int dz_profile_get_list(dz_engine* e, int32_t which, double* ms /* [cap] */, int64_t cap, int64_t* launches);   /* every launch of the class, in order */

#ifdef __cplusplus
}
#endif
#endif
