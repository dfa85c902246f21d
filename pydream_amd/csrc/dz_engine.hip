// dz_engine.hip -- host side of libdreamzs.so: device-resident state of the MT-DREAM(ZS)
// sampler (Z archive, chain states, adaptation accumulators, trace) and the per-generation
// launch sequence, behind the C ABI of include/dreamzs.h.
//
// Replaces, for the hot path: pydream/core.py:250-327 (_setup_mp_dream_pool / _mp_dream_init:
// shared arrays -> HBM buffers), core.py:89-129 (_sample_dream loop -> dz_step) and
// pydream/Dream.py:193-422 (astep -> the kernels of dz_kernels.h).
#include "../../include/dreamzs.h"
#include "dz_kernels.h"
#include "dz_megakernel.h"
#include "dz_mega_launch.h"

#include <hip/hip_ext.h>
#include <dlfcn.h>
#include <sys/mman.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <type_traits>
#include <thread>
#include <utility>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }

#define HIPCK(expr)                                                                                   \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)
#define DZCK(expr) do { int _r = (expr); if (_r) return _r; } while (0)

enum { LK_NONE = 0, LK_MVN = 1, LK_MIX = 2, LK_HOST = 3, LK_MODULE = 4 };
enum { PR_PROPOSE = 0, PR_LOGP = 1, PR_ACCEPT = 2, PR_ADAPT = 3, PR_EXCHANGE = 4, PR_GENERATIONS = 5, PR_EMPTY = 6, PR_COUNT = 7 };

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string path, hip_path;
    // RCCL must sit on the SAME HIP runtime as this library: device pointers and streams of one runtime instance mean nothing to
    // another.  A process that also imports PyTorch holds a second ROCm stack (torch/lib/librccl.so + its own libamdhip64), and a
    // bare dlopen("librccl.so.1") hands back whichever copy was loaded first.  So the library is opened by path, next to the
    // libamdhip64 this engine is linked against (DZ_RCCL_LIB overrides), and the HIP runtime it resolves is checked against ours.
    int load()
    {
        if (lib) return 0;
        Dl_info hi;
        if (!dladdr((void*)&hipGetDeviceCount, &hi) || !hi.dli_fname) return fail("cannot locate the HIP runtime this library is linked against");
        hip_path = hi.dli_fname;
        const std::string dir = hip_path.substr(0, hip_path.find_last_of('/'));
        std::vector<std::string> cand;
        if (const char* ov = getenv("DZ_RCCL_LIB")) cand.push_back(ov);
        cand.push_back(dir + "/librccl.so.1");
        cand.push_back(dir + "/librccl.so");
        cand.push_back("librccl.so.1");        // installs that keep RCCL apart from HIP (split packages, LD_LIBRARY_PATH): the runtime check below still applies
        std::string tried;
        for (const std::string& c : cand) {
            lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (lib) { path = c; break; }
            tried += c + " (" + dlerror() + "); ";
        }
        if (!lib) return fail("cannot load librccl (next to " + hip_path + " or by soname): " + tried);
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        CommCount = (decltype(CommCount))dlsym(lib, "ncclCommCount");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy) { dlclose(lib); lib = nullptr; return fail("librccl lacks required symbols: " + path); }
        Dl_info ri, rh;
        if (dladdr((void*)GetUniqueId, &ri) && ri.dli_fname) path = ri.dli_fname;
        void* their_hip = dlsym(lib, "hipGetDeviceCount");         // (searches librccl and its dependencies: the runtime it was built against)
        if (their_hip && dladdr(their_hip, &rh) && rh.dli_fname && hip_path != rh.dli_fname && !getenv("DZ_RCCL_ALLOW_MISMATCH")) {
            const std::string theirs = rh.dli_fname;
            dlclose(lib); lib = nullptr;
            return fail("librccl " + path + " uses the HIP runtime " + theirs + " but this engine uses " + hip_path +
                        ": two ROCm stacks in one process (set DZ_RCCL_LIB to the matching librccl)");
        }
        return 0;
    }
};
Rccl g_rccl;

template <class T> int dalloc(T** p, size_t n)
{
    HIPCK(hipMalloc((void**)p, sizeof(T) * (n ? n : 1)));
    HIPCK(hipMemset(*p, 0, sizeof(T) * (n ? n : 1)));
    // hipMemset on device memory may return before the fill has run, and the fill is queued on the NULL stream, which the engine's
    // non-blocking streams do not wait for: without this wait a late fill can wipe what the first kernels wrote (seen as zeroed
    // log_ps of a short run_dream call right after another engine's work)
    HIPCK(hipStreamSynchronize(nullptr));
    return 0;
}

}  // namespace

struct dz_engine {
    dz_config c{};
    dz::Params p{};
    hipStream_t stream = nullptr;
    // independent chain groups ("lanes") run their propose/logp/accept kernels on separate streams: between two
    // history appends the groups share nothing, so one group's likelihood (matrix pipe) overlaps another's
    // proposal generation (VALU) -- lane 0 is `stream`
    int nlanes = 1; hipStream_t lane_stream[8] = {nullptr}; hipEvent_t lane_ev[8] = {nullptr}; bool need_join = true;
    // bounded host run-ahead: a marker every ra_stride generations, the host never gets more than 3 markers ahead
    dz::Params p_shadow; bool params_uploaded = false;    // what d_params holds
    void* h_pin = nullptr; hipEvent_t pin_ev[2] = {nullptr, nullptr};          // page-locked bounce buffer (two halves) for downloads into pageable memory (d2h_2d)
    double* d_bar = nullptr;                // dz_comm_barrier's all-gather buffer (one element per rank)
    double* d_qpart = nullptr; size_t qpart_len = 0; bool force_big = false; bool logp_gemm = true; int logp_bm = 0;    // DZ_LOGP_GEMM=0: no LDS-tiled product; row-tile sums of the tiled large-d likelihood; DZ_LOGP_BIG=1: the one-wave-per-tile kernel
    int logp_waves = 0;      // DZ_LOGP_WAVES: force the block size of k_logp_mvn_lds (tuning)
    int ra_stride = 32; hipEvent_t ra_ev[4] = {nullptr}; bool ra_used[4] = {false, false, false, false}; int64_t ra_n = 0;
    int nch = 1;
    int64_t M = 0, gen = 0, ntrace = 0, draws_gen = -1;
    int64_t napp = 0;               // appends made since the archive was set (dz_config.history_lag: the last `lag` of them are not sampleable yet)
    std::vector<int64_t> gen_c;     // per-chain generation counters (differ only under single-chain stepping)
    uint4* d_draws[2] = {nullptr, nullptr};
    dz::ChainCtl* d_ctl[2] = {nullptr, nullptr};
    bool have_logp = false, adapt = false;
    int lk = LK_NONE;
    dz_logp_cb cb = nullptr; void* cb_user = nullptr;
    // a user-supplied DEVICE likelihood (dz_set_likelihood_module): a kernel of a code object the user built, launched where k_logp_* run
    hipModule_t lk_module = nullptr; hipFunction_t lk_fn = nullptr; void* d_lk_data = nullptr; int lk_lanes = 1; bool lk_finite = false;
    // ... and, when the code object has them, the persistent kernels with that density inlined (generations_wave_body<.., UserLike>): [0] lean, [1] full proposal code
    hipFunction_t lk_gen_fn[4] = {nullptr, nullptr, nullptr, nullptr};      // ([2], [3]: 128 < d <= 256)
    dz_exchange_cb xcb = nullptr; void* xcb_user = nullptr;
    ncclComm_t comm = nullptr; int rank = 0, world = 1;
    // peer transport (dz_peer_export / dz_peer_attach): the other ranks' archives, position buffers and flag words mapped into this
    // process; rows travel on copy streams (one per peer: xGMI is point to point), never through a kernel
    struct Peer { double* Z = nullptr; double* cp[3] = {nullptr, nullptr, nullptr}; double* gs[2] = {nullptr, nullptr}; unsigned long long* flags = nullptr; hipStream_t st = nullptr; };
    std::vector<Peer> peers; bool peer_on = false;
    unsigned long long* d_flags = nullptr;      // [4][world]: exchange number received from every rank -- [0] history appends, [1] published positions, [2] the attach-time self-test, [3] adaptation group sums
    unsigned long long* d_seq = nullptr; int64_t seq_cap = 0;     // seq[i] = i: the source of the flag pushes
    unsigned long long* h_gate = nullptr;       // host-mapped: ticks waited, gates passed, error (k_peer_gate)
    hipEvent_t push_ev[8] = {nullptr}; int push_n = 0;
    int64_t z_pushed = 0, z_gated = 0, pos_pushed = 0, hello_seq = 1;      // (hello 1 = the attach-time self-test)
    double* d_cp[3] = {nullptr, nullptr, nullptr}; int cp_idx = 1;      // published positions rotate through three buffers (a peer may run one generation ahead)
    // owned device buffers (also referenced from p)
    double *d_mins = nullptr, *d_maxs = nullptr, *d_gtab = nullptr, *d_mu = nullptr, *d_Mt = nullptr, *d_Mtp = nullptr, *d_mixF = nullptr;
    double *d_pa = nullptr, *d_pb = nullptr, *d_plogb = nullptr, *d_pc2 = nullptr; int32_t* d_pkind = nullptr;
    double *d_shared = nullptr;      // cr_probs|cr_delta|cr_n|g_probs|g_delta|g_n -- two copies: e->p points at the current one (sh_cur); a persistent launch
    int sh_cur = 0;                  // that applies pending adaptation totals in its prologue reads the current copy and leaves the new state in the other
    bool adapt_pending = false;      // d_TOT / d_CNT hold totals that no launch has applied yet (only between two launches inside dz_step)
    double *d_partial = nullptr, *d_mean = nullptr, *d_sd = nullptr, *d_sdc = nullptr, *d_dl = nullptr, *d_dlg = nullptr;
    int *d_binc = nullptr, *d_bing = nullptr;
    double* d_binsum = nullptr;     // k_adapt_update's scratch: [strips of 64 chains][ncr + ngamma] sums, then the same shape of counts
    // lockstep adaptation, contract v3 (dz_kernels.h adapt_unit_sums): the units' sums [units][nq][ld] and counts [units][ncr + ngamma], their totals
    double *d_PR = nullptr, *d_PC = nullptr, *d_TOT = nullptr, *d_CNT = nullptr;
    bool mega_mix_wide = true;      // the wave-per-chain kernels (mixture, a user's device function) also at 128 < d <= 256 (DZ_MEGA_MIX_WIDE=0: the multi-kernel path there)
    bool adapt_fused = true;        // the persistent kernels make their block's unit sums themselves (DZ_ADAPT_FUSED=0: k_adapt_partials does)
    // dz_config.adapt_lag = L >= 1 (round 6): d_TOT / d_CNT are rings of L + 1 slots (slot = generation mod (L + 1); d_CNT rows of ad_nbp), d_DOT
    // [L + 1][ad_nbp] the bins' dot products (k_adapt_dots), d_PR / d_PC rings as well when launches hold several burn-in generations (ad_multi).
    // Updates of generations (ad_applied, ad_made] are pending: made, not yet added to the shared state.  d_x0ring [2 (L + 1)][ld]: global chain
    // 0's published position after generation h at slot h mod 2 (L + 1); d_x0start: before generation 0.
    int ad_R1 = 1, ad_nbp = 0; bool ad_multi = false; int64_t ad_applied = -1, ad_made = -1; size_t ad_tot_stride = 0, ad_pr_stride = 0, ad_pc_stride = 0;
    double *d_DOT = nullptr, *d_x0ring = nullptr, *d_x0start = nullptr;
    // ... and for the kernels that do not make their blocks' unit sums (burnin_multi == 2): d_posring [L + 2][N][ld] the published positions of the last
    // L + 2 generations (slot = generation mod (L + 2); made on first use), posring_gen the last generation in it; d_PG [L + 1][ad_nbp] the probabilities
    // the last launch's generations decided with (k_adapt_partials_ring)
    double *d_posring = nullptr, *d_PG = nullptr; int64_t posring_gen = -2; bool ad_ring = true;
    // sharded crossover burn-in (round 5): a rank that owns whole groups of 256 chains exchanges its groups' sums (dz_kernels.h k_adapt_groups /
    // k_group_totals) instead of its positions.  d_GS[parity of the generation]: [world][gs_rec] records (two buffers: a peer may be one
    // generation ahead); d_shift[parity]: global chain 0's position after that generation (the shift of the next one's column sums)
    bool adapt_groups = false; double* d_GS[2] = {nullptr, nullptr}; double* d_shift[2] = {nullptr, nullptr}; size_t gs_rec = 0; int gs_nbp = 0;
    int64_t gs_pushed = 0, gs_last_gen = -1, gs_launches = 0;
    int64_t bytes_z = 0, bytes_pos = 0, bytes_sums = 0;      // bytes this rank has sent to EACH other rank, by kind (dz_exchange_bytes)
    dz::Params* d_params = nullptr;  // device copy of `p` for kernels that take it by pointer
    double *d_scratch = nullptr; size_t scratch_rows = 0;   // debug / eval staging [rows, ld]
    double *d_cmean = nullptr, *d_cvar = nullptr, *d_rhat = nullptr;
    std::vector<double> h_stage;     // host staging (callback likelihood / exchange)
    bool prof = false;
    int num_cu = 256;
    int waves_per_block = 0;        // DZ_WPB
    bool fuse = true;               // DZ_FUSE=0 disables the accept+propose fusion
    bool fuse_stream = true;        // streamed generations: the Metropolis step rides in front of the next generation's proposal set (k_accept_propose; DZ_FUSE_STREAM=0: off)
    int64_t stream_prop_gen = -1;   // the generation whose proposal set k_accept_propose has already made
    bool q_defer = true;            // streamed generations: no k_q_finish launches, the proposal / Metropolis kernels add the row-tile sums (DZ_QFIN=0: off)
    double* last_qpart = nullptr;   // the scratch slice the last tiled likelihood launch wrote
    bool stream_propose = true;     // ld > 256: the streaming proposal kernel (k_propose_stream); DZ_STREAM=0 keeps k_propose<4|8>
    bool mega = true;               // the persistent generation kernel serves every eligible configuration (mega_eligible); DZ_MEGA=0 forces the multi-kernel path
    bool mega_redo_on = true;       // redraw rounds (Dream.py:281-289) inside the persistent kernel; DZ_MEGA_REDO=0: such configurations take the multi-kernel path
    unsigned long long* d_redraw_count = nullptr;
    bool mega_mix_pb = true;        // the mixture kernel's full-code instantiation (priors, boundaries, several pairs); DZ_MEGA_MIX_PB=0: multi-kernel path there
    int mega_d2 = 1;                // 128 < d <= 256: k_generations_d2 from 1025 chains on (DZ_MEGA_D2=0: the multi-kernel path there; 2: at any chain count)
    bool mega_w4 = true;            // small populations (4 chains x 4 waves per block), lean, multitry 3..6: k_generations_w4 (DZ_MEGA_W4=0: k_generations<.., 4, 4, lean>)
    bool mega_split = true;         // a remainder of chains beyond whole rounds of 16-chain blocks goes in a second launch of smaller blocks; DZ_MEGA_SPLIT=0: off
    bool mega_burnin = true;        // ... the generations of the crossover burn-in too, one per launch (positions published by the kernel); DZ_MEGA_BURNIN=0: multi-kernel path there
    int mega_max_gen = 1 << 20;     // DZ_MEGA_MAXGEN: generations per launch cap (measurement)
    int mega_segs = 1 << 20;        // DZ_MEGA_SEGS: history appends per launch cap (1: a launch ends with its append, as without a lag)
    int mega_ch = 0;                // DZ_MEGA_CHAINS: force 16 / 8 / 4 chains per block (0: by chain count)
    bool tempering = false; double* d_Tc = nullptr; int32_t* d_tswap = nullptr;    // parallel tempering (dz_set_temperatures)
    // Dream.py:281-289: a proposal set whose tries are all impossible is drawn again (redo_possible / one_generation)
    uint8_t* d_redo = nullptr; int32_t* d_redo_list = nullptr; uint8_t* h_redo = nullptr; int32_t* h_redo_list = nullptr;     // (host side: page-locked)
    std::vector<int32_t> h_pkind; std::vector<double> h_pa, h_pb, h_mins, h_maxs;     // host copies: is the uniform priors' support covered by the hard boundaries?
    hipStream_t copy_stream = nullptr; hipEvent_t copy_ev[8] = {nullptr};   // dz_trace_download_begin / _wait: trace rows leave while later generations run
    int64_t redraw_rounds = 0;      // redraw launches so far (dz_redraw_rounds)
    double *d_own_cr = nullptr, *d_own_g = nullptr; std::vector<char> own_init;      // per-instance probabilities of single-chain stepping (Params::own_cr)
    std::string broken;             // non-empty: a failed dz_continue_run left the engine without some of its buffers -- dz_step refuses
    std::string last_variant;       // what the last dz_step launched for its generations (dz_last_kernel_variant)
    bool pending_accept = false;    // generation gen-1's Metropolis step has been deferred into the next proposal kernel
    const double* pending_qfin_r = nullptr;      // ... and with it the reference set's row-tile sums (Params::qfin_r), still in the scratch array
    int64_t pending_slot = -1;
    int propose_split = 0;          // waves per chain in k_propose (DZ_PROPOSE_SPLIT); 0 = by problem shape
    int force_pt = 0;               // measurement switch: DZ_MFMA_PT=1|2 forces the point tiles per wave
    std::vector<hipEvent_t> ev_pool;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[PR_COUNT];
    std::vector<void*> to_free;
};

namespace {

struct ProfScope {
    dz_engine* e; int which; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    bool on;
    ProfScope(dz_engine* e_, int w, hipStream_t st_ = nullptr, bool active = true) : e(e_), which(w), st(st_ ? st_ : e_->stream), on(active && e_->prof)
    {
        if (on) {
            for (hipEvent_t* x : {&a, &b}) {
                if (!e->ev_pool.empty()) { *x = e->ev_pool.back(); e->ev_pool.pop_back(); }
                else (void)hipEventCreate(x);
            }
            (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope() { if (on) { (void)hipEventRecord(b, st); e->ev[which].emplace_back(a, b); } }
};

hipEvent_t prof_event(dz_engine* e)
{
    hipEvent_t x = nullptr;
    if (!e->ev_pool.empty()) { x = e->ev_pool.back(); e->ev_pool.pop_back(); }
    else (void)hipEventCreate(&x);
    return x;
}
// One kernel launch of profile class CLS.  While profiling, the launch carries its own start/stop events
// (hipExtLaunchKernelGGL): they take the dispatch's begin/end timestamps -- the figures rocprofv3's kernel trace
// reports -- instead of bracketing the launch with two extra barrier packets that each add microseconds.
#define DZ_KLAUNCH(E, CLS, ST, KERN, GRID, BLOCK, LDS, ...)                                               \
    do {                                                                                                   \
        if ((E)->prof) {                                                                                   \
            hipEvent_t ka_ = prof_event(E), kb_ = prof_event(E);                                           \
            hipExtLaunchKernelGGL(KERN, GRID, BLOCK, LDS, ST, ka_, kb_, 0, __VA_ARGS__);                   \
            (E)->ev[CLS].emplace_back(ka_, kb_);                                                           \
        } else hipLaunchKernelGGL(KERN, GRID, BLOCK, LDS, ST, __VA_ARGS__);                                \
    } while (0)

// archive rows the next generation samples from: everything written, less the rows of the last `history_lag` appends
int64_t visible_rows(const dz_engine* e)
{
    const int64_t lag = std::min<int64_t>(e->napp, (int64_t)e->c.history_lag);
    return e->M - (int64_t)e->p.N * lag;
}

int peer_check(dz_engine* e);
// everything this engine has queued: the lane streams AND the per-peer copy streams -- a row or flag push still in flight when the
// caller goes on to a control-plane barrier and then frees or unmaps buffers would be a DMA write into memory that is gone
int sync_all(dz_engine* e)
{
    for (int s = 0; s < e->nlanes; ++s) HIPCK(hipStreamSynchronize(e->lane_stream[s]));
    for (auto& pr : e->peers) if (pr.st) HIPCK(hipStreamSynchronize(pr.st));
    return peer_check(e);
}

// Device -> pageable host, 2-D: through a page-locked bounce buffer owned by the engine (stream-ordered copy, wait, then
// plain memcpy).  Asynchronous and blocking 2-D copies straight into pageable memory both misbehaved on this stack
// (a rare late landing after the stream wait; zeros from the blocking form once several engines had lived in the process).
int d2h_2d(dz_engine* e, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height)
{
    if (!width || !height) return 0;
    constexpr size_t CAP = (size_t)64 << 20, HALF = CAP / 2;
    if (!e->h_pin) { if (hipHostMalloc(&e->h_pin, CAP, hipHostMallocDefault) != hipSuccess) { e->h_pin = nullptr; return fail("hipHostMalloc (bounce buffer) failed"); } }
    if (width > HALF) return fail("row too long for the bounce buffer");
    // two halves: the DMA of the next batch of rows runs while the host copies the batch before out of the other half
    const size_t rows_per = std::max<size_t>(1, HALF / width);
    auto issue = [&](size_t r0, int half) -> hipError_t {
        const size_t nr = std::min(rows_per, height - r0);
        return hipMemcpy2DAsync((char*)e->h_pin + (size_t)half * HALF, width, (const char*)src + r0 * spitch, spitch, width, nr, hipMemcpyDeviceToHost, e->stream);
    };
    if (!e->pin_ev[0]) for (int i = 0; i < 2; ++i) HIPCK(hipEventCreateWithFlags(&e->pin_ev[i], hipEventDisableTiming));
    HIPCK(issue(0, 0));
    HIPCK(hipEventRecord(e->pin_ev[0], e->stream));
    int half = 0;
    for (size_t r0 = 0; r0 < height; r0 += rows_per, half ^= 1) {
        const size_t nr = std::min(rows_per, height - r0);
        if (r0 + rows_per < height) { HIPCK(issue(r0 + rows_per, half ^ 1)); HIPCK(hipEventRecord(e->pin_ev[half ^ 1], e->stream)); }
        HIPCK(hipEventSynchronize(e->pin_ev[half]));
        const char* from = (const char*)e->h_pin + (size_t)half * HALF;
        if (dpitch == width) memcpy((char*)dst + r0 * dpitch, from, nr * width);
        else for (size_t r = 0; r < nr; ++r) memcpy((char*)dst + (r0 + r) * dpitch, from + r * width, width);
    }
    HIPCK(hipStreamSynchronize(e->stream));
    return 0;
}

template <class T> int ealloc(dz_engine* e, T** p, size_t n)
{
    DZCK(dalloc(p, n));
    e->to_free.push_back((void*)*p);
    return 0;
}

int upload_padded(dz_engine* e, double* dst, const double* src, int rows, double padval)
{   // src [rows,d] -> dst [rows,ld]
    const int d = e->p.d, ld = e->p.ld;
    std::vector<double> h((size_t)rows * ld, padval);
    for (int r = 0; r < rows; ++r) memcpy(&h[(size_t)r * ld], src + (size_t)r * d, sizeof(double) * d);
    HIPCK(hipMemcpyAsync(dst, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, e->stream));
    DZCK(sync_all(e));
    return 0;
}
int download_rows(dz_engine* e, double* dst, const double* src, size_t rows)
{   // src [rows,ld] device -> dst [rows,d] host
    DZCK(sync_all(e));
    return d2h_2d(e, dst, sizeof(double) * e->p.d, src, sizeof(double) * e->p.ld, sizeof(double) * e->p.d, rows);
}

#define NCH_DISPATCH(e, CALL)                      \
    switch ((e)->nch) {                            \
        case 1: { constexpr int NCH = 1; CALL; } break; \
        case 2: { constexpr int NCH = 2; CALL; } break; \
        case 4: { constexpr int NCH = 4; CALL; } break; \
        default: { constexpr int NCH = 8; CALL; } break; \
    }

int launch_check(const char* what)
{
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(std::string(what) + ": " + hipGetErrorString(err));
    return 0;
}

// Model.total_logp for n points stored [n,ld] on the device
// defer_finish (large-d MVN only): the row-tile sums stay in the scratch array (e->last_qpart) for the consuming kernel to add
// (Params::qfin_*) instead of a k_q_finish launch
int eval_logp(dz_engine* e, const double* pts, int n, double* prior, double* like, hipStream_t st = nullptr, bool defer_finish = false)
{
    if (n <= 0) return 0;
    if (!st) st = e->stream;
    const int nrt = e->p.ld / 16;
    const int ks4 = 4 * ((e->p.d + 3) / 4);
    // LDS kernel: matrix (packed triangle or square) + mean + at least 4 wave tiles must fit 160 KB
    // (the dense 128-D square does not: 131 KB + 66 KB; it runs on the register-operand MFMA kernel)
    const bool lds_fits = sizeof(double) * ((e->p.tri ? (size_t)e->p.mtp_len : (size_t)ks4 * e->p.ld) + e->p.ld + (size_t)4 * 16 * (e->p.ld + 1)) <= (size_t)160 * 1024;
    const bool one_kernel = e->lk == LK_MVN && !e->p.have_prior && nrt <= 8 && !e->force_pt && lds_fits;   // the LDS kernel alone: timed by the launch's own events
    ProfScope ps(e, PR_LOGP, st, !one_kernel);
    const dim3 grid((n + 3) / 4), block(256);
    if (e->lk == LK_MVN) {
        {
            if (nrt <= 8 && !e->force_pt && lds_fits) {
                const int ntiles = (n + 15) / 16;
                // one wave per point tile of the CU's share when LDS allows (4..8 waves): 1280 tiles on 256 CUs run as
                // 5-wave blocks instead of 4-wave blocks of which a quarter does a second tile
                const size_t lds_fixed = sizeof(double) * ((e->p.tri ? (size_t)e->p.mtp_len : (size_t)ks4 * e->p.ld) + e->p.ld), lds_wave = sizeof(double) * (size_t)16 * (e->p.ld + 1);
                int nwv = std::max(4, std::min(8, (ntiles + e->num_cu - 1) / e->num_cu));
                while (nwv > 4 && lds_fixed + nwv * lds_wave > (size_t)160 * 1024) --nwv;
                if (e->logp_waves) nwv = e->logp_waves;
                const size_t lds = lds_fixed + nwv * lds_wave;
                const dim3 gl((unsigned)std::min(e->num_cu, (ntiles + nwv - 1) / nwv)), bl(64 * nwv);
#define DZ_LDS_CASE(NRT_)                                                                                                      \
    case NRT_:                                                                                                                 \
        if (!one_kernel) {                                                                                                     \
            if (e->p.tri) hipLaunchKernelGGL((dz::k_logp_mvn_lds<NRT_, true>), gl, bl, lds, st, e->p, pts, n, prior, like);   \
            else hipLaunchKernelGGL((dz::k_logp_mvn_lds<NRT_, false>), gl, bl, lds, st, e->p, pts, n, prior, like);           \
        } else if (e->p.tri) DZ_KLAUNCH(e, PR_LOGP, st, (dz::k_logp_mvn_lds<NRT_, true>), gl, bl, lds, e->p, pts, n, prior, like); \
        else DZ_KLAUNCH(e, PR_LOGP, st, (dz::k_logp_mvn_lds<NRT_, false>), gl, bl, lds, e->p, pts, n, prior, like);           \
        break;
                switch (nrt) { DZ_LDS_CASE(1) DZ_LDS_CASE(2) DZ_LDS_CASE(3) DZ_LDS_CASE(4) DZ_LDS_CASE(5) DZ_LDS_CASE(6) DZ_LDS_CASE(7) DZ_LDS_CASE(8) }
#undef DZ_LDS_CASE
            } else if (nrt <= 8) {
                const int ntiles = (n + 15) / 16;
                const int pt = (e->force_pt ? e->force_pt : (ntiles > 1024 ? 2 : 1));   // 1024 SIMDs: share B operands between two point tiles once every SIMD has work
                const dim3 g2((n + 64 * pt - 1) / (64 * pt));
#define DZ_MFMA_CASE(NRT_)                                                                                                                     \
    case NRT_:                                                                                                                                 \
        if (pt == 2) { if (e->p.tri) hipLaunchKernelGGL((dz::k_logp_mvn_mfma<2, NRT_, true>), g2, block, 0, st, e->p, pts, n, prior, like);   \
                       else hipLaunchKernelGGL((dz::k_logp_mvn_mfma<2, NRT_, false>), g2, block, 0, st, e->p, pts, n, prior, like); }  \
        else { if (e->p.tri) hipLaunchKernelGGL((dz::k_logp_mvn_mfma<1, NRT_, true>), g2, block, 0, st, e->p, pts, n, prior, like);           \
               else hipLaunchKernelGGL((dz::k_logp_mvn_mfma<1, NRT_, false>), g2, block, 0, st, e->p, pts, n, prior, like); }          \
        break;
                switch (nrt) { DZ_MFMA_CASE(1) DZ_MFMA_CASE(2) DZ_MFMA_CASE(3) DZ_MFMA_CASE(4) DZ_MFMA_CASE(5) DZ_MFMA_CASE(6) DZ_MFMA_CASE(7) DZ_MFMA_CASE(8) }
#undef DZ_MFMA_CASE
            } else {
                // tiled product: waves = point-tile groups x row-tile groups; per-row-tile sums through a scratch array
                constexpr int PT = 2, RTC = 4;
                const int nrtb = (e->p.d + 15) / 16, nrg = (nrtb + RTC - 1) / RTC, npg = (n + 16 * PT - 1) / (16 * PT);
                // every chain-group stream has its own slice of the row-tile scratch array
                int sl = 0;
                for (int s2 = 1; s2 < e->nlanes; ++s2) if (e->lane_stream[s2] == st) sl = s2;
                const size_t need = (size_t)n * nrtb;
                if (need * e->nlanes > e->qpart_len) {
                    DZCK(sync_all(e));
                    if (e->d_qpart) (void)hipFree(e->d_qpart);
                    e->d_qpart = nullptr; e->qpart_len = 0;
                    DZCK(dalloc(&e->d_qpart, need * e->nlanes));
                    e->qpart_len = need * e->nlanes;
                }
                double* qpart = e->d_qpart + (size_t)sl * (e->qpart_len / e->nlanes);
                if (e->force_big) hipLaunchKernelGGL(dz::k_logp_mvn_mfma_big<8>, dim3((n + 63) / 64), block, 0, st, e->p, pts, n, prior, like);
                else {
                    if (e->logp_gemm && n >= 512) {            // enough points to fill the chip with 64 x 64 block tiles
                        const int nbn = (nrtb * 16 + 63) / 64;
                        const size_t ldsm = e->p.mu_zero ? 0 : sizeof(double) * (size_t)e->p.ld;
                        // Points per block (128 / 64 / 32): the kernel is bound by how many blocks a CU has in flight (each one is a
                        // chain of barrier -> LDS reads -> MFMAs -> LDS store), so the largest tile that still leaves five blocks per
                        // CU: 2560 points x 1000 rows: 32 points 65 us, 64 points 70 us, 128 points 98 us per launch.
                        int bmsel = e->logp_bm;
                        if (bmsel != 32 && bmsel != 64 && bmsel != 128)
                            bmsel = ((n + 127) / 128) * nbn >= 5 * e->num_cu ? 128 : ((n + 63) / 64) * nbn >= 5 * e->num_cu ? 64 : 32;
                        const dim3 gridg(((n + bmsel - 1) / bmsel) * nbn);
                        if (bmsel == 32) hipLaunchKernelGGL(dz::k_logp_mvn_gemm<1>, gridg, block, ldsm, st, e->p, pts, n, qpart, e->num_cu);
                        else if (bmsel == 64) hipLaunchKernelGGL(dz::k_logp_mvn_gemm<2>, gridg, block, ldsm, st, e->p, pts, n, qpart, e->num_cu);
                        else hipLaunchKernelGGL(dz::k_logp_mvn_gemm<4>, gridg, block, ldsm, st, e->p, pts, n, qpart, e->num_cu);
                    } else
                    hipLaunchKernelGGL((dz::k_logp_mvn_mfma_tiled<PT, RTC>), dim3((npg * nrg + 3) / 4), block, 0, st, e->p, pts, n, qpart);
                    e->last_qpart = qpart;
                    if (!defer_finish) hipLaunchKernelGGL(dz::k_q_finish, dim3((n + 63) / 64), dim3(64), 0, st, e->p, (const double*)qpart, n, nrtb, prior, like);
                }
            }
            if (e->p.have_prior) NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_prior_only<NCH>, grid, block, 0, st, e->p, pts, n, prior));
        }
    } else if (e->lk == LK_MIX) {
        NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_logp_mix<NCH>, grid, block, 0, st, e->p, pts, n, prior, like));
    } else if (e->lk == LK_MODULE) {
        // the user's kernel writes like[n]; the engine zeroes prior[n] first and then adds the device priors and maps nan -> -inf (k_prior_add),
        // exactly what follows a host callback
        struct { const double* X; long long n; int d; int ld; double* like; const void* data; } a = {pts, (long long)n, e->p.d, e->p.ld, like, e->d_lk_data};
        size_t asz = sizeof(a);
        void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
        HIPCK(hipMemsetAsync(prior, 0, sizeof(double) * (size_t)n, st));
        const unsigned per_block = e->lk_lanes == 64 ? 4u : 256u;      // one thread or one wave per point, 256 threads per block
        HIPCK(hipModuleLaunchKernel(e->lk_fn, ((unsigned)n + per_block - 1) / per_block, 1, 1, 256, 1, 1, 0, st, nullptr, extra));
        NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_prior_add<NCH>, grid, block, 0, st, e->p, pts, n, prior, like));
    } else if (e->lk == LK_HOST) {
        const int d = e->p.d;
        e->h_stage.resize((size_t)n * (d + 2));
        double* hx = e->h_stage.data(); double* hp = hx + (size_t)n * d; double* hl = hp + n;
        DZCK(download_rows(e, hx, pts, (size_t)n));
        for (int i = 0; i < n; ++i) { hp[i] = 0.0; hl[i] = 0.0; }
        if (e->cb(hx, n, d, hp, hl, e->cb_user)) return fail("host likelihood callback failed");
        HIPCK(hipMemcpyAsync(prior, hp, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
        HIPCK(hipMemcpyAsync(like, hl, sizeof(double) * n, hipMemcpyHostToDevice, e->stream));
        DZCK(sync_all(e));
        NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_prior_add<NCH>, grid, block, 0, st, e->p, pts, n, prior, like));
    } else return fail("no likelihood set");
    return launch_check("logp kernel");
}

enum { XK_Z = 0, XK_POS = 1, XK_HELLO = 2, XK_SUMS = 3, XK_KINDS = 4 };
double gate_timeout_s()
{
    // (liveness of the other ranks is the control plane's business; the bound only keeps a kernel from spinning for ever -- a rank with
    //  a host-callback likelihood or a slow start can legitimately be minutes late)
    if (const char* s = getenv("DZ_PEER_TIMEOUT_S")) return std::max(0.01, atof(s));
    return 600.0;
}
// peer transport: this rank's rows of `buf` (global layout [N,ld], rows [off, off+nl) at `row0`) go to every peer's copy of the
// buffer, each followed by this rank's flag word for exchange number `seq` of that kind
// (XK_SUMS: `buf` = d_GS[which_cp], this rank's record of gs_rec doubles at rank * gs_rec)
int peer_push(dz_engine* e, int kind, double* buf, int which_cp, size_t row0, unsigned long long seq, size_t sums_cnt = 0)
{
    if ((int64_t)seq >= e->seq_cap) return fail("peer exchange: sequence capacity exceeded");
    hipEvent_t ev = e->push_ev[e->push_n++ & 7];
    HIPCK(hipEventRecord(ev, e->stream));
    const size_t srec = sums_cnt ? sums_cnt : e->gs_rec;      // (a launch of several burn-in generations: that many records per rank, back to back)
    const size_t first = kind == XK_SUMS ? (size_t)e->rank * srec : (row0 + (size_t)e->p.off) * e->p.ld;
    const size_t cnt = kind == XK_SUMS ? srec : (size_t)e->p.nl * e->p.ld;
    for (int r = 0; r < e->world; ++r) {
        if (r == e->rank) continue;
        dz_engine::Peer& pr = e->peers[r];
        double* dst = kind == XK_Z ? pr.Z : (kind == XK_SUMS ? pr.gs[which_cp] : pr.cp[which_cp]);
        HIPCK(hipStreamWaitEvent(pr.st, ev, 0));
        if (kind != XK_HELLO) HIPCK(hipMemcpyAsync(dst + first, buf + first, sizeof(double) * cnt, hipMemcpyDeviceToDevice, pr.st));
        HIPCK(hipMemcpyAsync(pr.flags + (size_t)kind * e->world + e->rank, e->d_seq + seq, sizeof(unsigned long long), hipMemcpyDeviceToDevice, pr.st));
    }
    return 0;
}
// ... and the wait for everybody else's rows of exchange `need`: a one-wave kernel on the engine's stream (k_peer_gate)
int peer_gate(dz_engine* e, int kind, unsigned long long need, double timeout_s = 0.0)
{
    const unsigned long long ticks = (unsigned long long)((timeout_s > 0.0 ? timeout_s : gate_timeout_s()) * 1e8);
    // (rendezvous gates keep their own counters, [4..6]: what they wait for is the other ranks' lateness, not an exchange)
    hipLaunchKernelGGL(dz::k_peer_gate, dim3(1), dim3(64), 0, e->stream, (const unsigned long long*)(e->d_flags + (size_t)kind * e->world), e->world, e->rank, need, ticks,
                       e->h_gate + (kind == XK_HELLO ? 4 : 0));
    return launch_check("k_peer_gate");
}
// did a gate give up?  h_gate is host-mapped, so this costs nothing: dz_step asks before it queues each launch (a timed-out gate is
// fatal for the run: whatever was queued behind it sampled rows that never arrived, and no getter hands such results out), every
// synchronising entry point asks after its wait
int peer_check(dz_engine* e)
{
    if (e->peer_on && e->h_gate && (e->h_gate[2] || e->h_gate[6])) {
        const int r = (int)(e->h_gate[2] ? e->h_gate[2] : e->h_gate[6]) - 1;
        return fail("peer exchange: no rows from rank " + std::to_string(r) + " within " + std::to_string(gate_timeout_s()) + " s (DZ_PEER_TIMEOUT_S)");
    }
    return 0;
}
// before generations that sample the archive: the rows of every append that is visible now have arrived from every rank
// (ahead: history appends the launch makes itself in front of its last generation -- the generations behind the j-th of them sample the rows
//  of one more append)
int ensure_visible(dz_engine* e, int ahead = 0)
{
    if (!e->peer_on) return 0;
    const int64_t made = e->napp + ahead;
    const int64_t need = made - std::min<int64_t>(made, (int64_t)e->c.history_lag);
    if (need > e->z_gated) { DZCK(peer_gate(e, XK_Z, (unsigned long long)need)); e->z_gated = need; }
    return 0;
}

// Replicates this rank's rows of `buf` (global layout [N,ld]; the rank's nl rows sit at row0 + off) on every rank.
//   RCCL: in-place ncclAllGather on the engine's stream;  host: staged through the exchange callback (tests, gloo);
//   peer: pushed by the copy engines into the peers' mapped buffers -- history appends are waited for only when their rows become
//   sampleable (ensure_visible: with history_lag >= 1 a whole thin-cycle later), published positions at once.
// XK_SUMS (sharded crossover burn-in): buf = d_GS[which], [world][gs_rec]; this rank's record is replicated, waited for at once.
int exchange_rows(dz_engine* e, int kind, double* buf, size_t row0, int which = 0, size_t sums_cnt = 0)
{
    if (!e->comm && !e->peer_on && e->p.nl == e->p.N) return 0;     // single GPU, no communicator: the kernels wrote the rows in place
    ProfScope ps(e, PR_EXCHANGE);
    const size_t srec = sums_cnt ? sums_cnt : e->gs_rec;      // (XK_SUMS: doubles per rank -- one record, or the records of a launch's generations back to back)
    const size_t cnt = kind == XK_SUMS ? srec : (size_t)e->p.nl * e->p.ld;
    double* base = kind == XK_SUMS ? buf : buf + row0 * e->p.ld;
    double* mine = kind == XK_SUMS ? buf + (size_t)e->rank * srec : base + (size_t)e->p.off * e->p.ld;
    (kind == XK_Z ? e->bytes_z : kind == XK_SUMS ? e->bytes_sums : e->bytes_pos) += (int64_t)(sizeof(double) * cnt);
    if (e->peer_on) {
        if (kind == XK_Z) { DZCK(peer_push(e, XK_Z, buf, 0, row0, (unsigned long long)(e->z_pushed + 1))); e->z_pushed++; return 0; }
        if (kind == XK_SUMS) {
            DZCK(peer_push(e, XK_SUMS, buf, which, 0, (unsigned long long)(e->gs_pushed + 1), srec)); e->gs_pushed++;
            return peer_gate(e, XK_SUMS, (unsigned long long)e->gs_pushed);
        }
        DZCK(peer_push(e, XK_POS, buf, e->cp_idx, row0, (unsigned long long)(e->pos_pushed + 1))); e->pos_pushed++;
        return peer_gate(e, XK_POS, (unsigned long long)e->pos_pushed);
    }
    if (e->comm) {
        ncclResult_t r = g_rccl.AllGather(mine, base, cnt, ncclDouble, e->comm, e->stream);
        if (r != ncclSuccess) return fail(std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return 0;
    }
    if (!e->xcb) return fail("sharded run needs dz_comm_init_rccl, dz_peer_attach or dz_set_exchange");
    const size_t nranks = (size_t)e->p.N / e->p.nl;
    e->h_stage.resize(cnt * (nranks + 1));
    double* hs = e->h_stage.data(); double* hr = hs + cnt;
    HIPCK(hipMemcpyAsync(hs, mine, sizeof(double) * cnt, hipMemcpyDeviceToHost, e->stream));
    DZCK(sync_all(e));
    if (e->xcb(hs, hr, (int64_t)(sizeof(double) * cnt), e->xcb_user)) return fail("exchange callback failed");
    HIPCK(hipMemcpyAsync(base, hr, sizeof(double) * cnt * nranks, hipMemcpyHostToDevice, e->stream));
    DZCK(sync_all(e));
    return 0;
}

void point_shared(dz_engine* e)
{   // e->p's six pointers at the current copy of the shared adaptation state
    dz::Params& p = e->p; const int ncr = e->c.ncr, ng = e->c.ngamma;
    p.cr_probs = e->d_shared + (size_t)e->sh_cur * 3 * (ncr + ng); p.cr_delta = p.cr_probs + ncr; p.cr_n = p.cr_delta + ncr;
    p.g_probs = p.cr_n + ncr; p.g_delta = p.g_probs + ng; p.g_n = p.g_delta + ng;
}
// adapt_lag >= 1: the last generation whose update generation g decides with -- g - 1 - lag inside the burn-in, everything that was made from the
// hand-over on (g > crossover_burnin: Dream.py:385-415)
int64_t adapt_due(const dz_engine* e, int64_t g)
{
    return g > (int64_t)e->p.burnin ? e->ad_made : std::min<int64_t>(e->ad_made, g - 1 - (int64_t)e->c.adapt_lag);
}
// ... and the pending updates up to there added to the shared state, on their own (k_adapt_apply_pending: in place)
int adapt_apply_due(dz_engine* e, int64_t g)
{
    const int64_t due = adapt_due(e, g);
    if (due <= e->ad_applied) return 0;
    hipLaunchKernelGGL(dz::k_adapt_apply_pending, dim3(1), dim3(64), 0, e->stream, e->p, (const double*)e->d_DOT, (const double*)e->d_CNT, e->ad_nbp, e->ad_R1,
                       (long long)(e->ad_applied + 1), (long long)(e->ad_made + 1), (long long)g, e->c.adapt_lag);
    e->ad_applied = due;
    return launch_check("k_adapt_apply_pending");
}
// totals left by the last generation's adaptation that no launch has applied yet: the update on its own
int adapt_flush(dz_engine* e)
{
    if (e->c.adapt_lag > 0) return adapt_apply_due(e, e->gen);
    if (!e->adapt_pending) return 0;
    NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_adapt_apply<NCH>, dim3(1), dim3(64), 0, e->stream, e->p, (const double*)e->d_TOT, (const double*)e->d_CNT));
    e->adapt_pending = false;
    return launch_check("k_adapt_apply");
}
// adapt_lag >= 1: behind the totals of generations g .. g + n - 1 (ring slots), their bins' dot products; the updates are then "made"
int adapt_dots(dz_engine* e, uint32_t g, int n)
{
    NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_adapt_dots<NCH>, dim3(n), dim3(64), 0, e->stream, e->p, (const double*)e->d_TOT, (const double*)e->d_CNT, e->d_DOT,
                                       (long long)g, e->ad_R1, (long long)e->ad_tot_stride, e->ad_nbp));
    e->ad_made = (int64_t)g + n - 1;
    return launch_check("k_adapt_dots");
}
// the totals of the units' sums (k_adapt_totals) of generations g .. g + n - 1 (pr_ring: a launch that can hold several burn-in generations, adapt_lag
// >= 1, has left their units' sums in ring slots; else d_PR / d_PC hold one generation's); adapt_lag 0: the update itself (adapt_apply_wave) is left to the next persistent launch's prologue
// when `defer` says one follows, else made at once (k_adapt_apply); adapt_lag >= 1: the dot products follow, the update waits until it is due
int adapt_finish(dz_engine* e, bool defer, uint32_t g = 0, int n = 1, bool pr_ring = false)
{
    const dz::Params& p = e->p;
    const int nq = 2 + p.ncr + p.ngamma, units = (p.N + 15) / 16, nb = p.ncr + p.ngamma;
    const bool ring = e->c.adapt_lag > 0;
    hipLaunchKernelGGL(dz::k_adapt_totals, dim3((nq * p.d + 15) / 16 + 1, n), dim3(256), 0, e->stream, (const double*)e->d_PR, (const double*)e->d_PC, units, nq, p.d, p.ld, nb, e->d_TOT, e->d_CNT,
                       (long long)g, ring ? e->ad_R1 : 1, (long long)(pr_ring ? e->ad_pr_stride : 0), (long long)(pr_ring ? e->ad_pc_stride : 0), (long long)e->ad_tot_stride, ring ? e->ad_nbp : 0);
    DZCK(launch_check("k_adapt_totals"));
    if (ring) return adapt_dots(e, g, n);
    e->adapt_pending = true;
    return defer ? 0 : adapt_flush(e);
}

// the row the column sums of generation g are taken around: global chain 0's previous published position (contract v3).  Ranks that exchange
// group sums instead of positions get it with the records of generation g - 1 (k_group_totals), generation 0's from the start positions.
// adapt_lag = L >= 1: global chain 0's position after generation g - 1 - L (its start position before there is one)
const double* adapt_shift(const dz_engine* e, uint32_t g)
{
    if (e->c.adapt_lag > 0) {
        const int64_t hs = (int64_t)g - 1 - e->c.adapt_lag;
        return hs < 0 ? e->d_x0start : e->d_x0ring + (size_t)(hs % (2 * e->ad_R1)) * e->p.ld;
    }
    return (e->adapt_groups && g > 0) ? e->d_shift[(g - 1u) & 1u] : e->p.cp_prev;
}
// Sharded (whole groups per rank), adapt_lag >= 1, a launch of n burn-in generations g .. g + n - 1 whose units' sums (this rank's units) sit in the
// rings' slots: every generation's group sums into one record each, the n records of every rank everywhere in ONE exchange, then the totals of
// every generation (the same additions in the same order as adapt_generation makes them one generation at a time) and their dot products.
// Buffers by the parity of the LAUNCH (a peer may be one launch ahead); rank r's records of this launch at r * n * gs_rec.
int adapt_finish_groups(dz_engine* e, uint32_t g, int n)
{
    const dz::Params& p = e->p;
    const int units_l = p.nl / 16, gl = p.nl / 256, nq = 2 + p.ncr + p.ngamma, nb = p.ncr + p.ngamma, nbp = e->gs_nbp, R1 = e->ad_R1;
    const int par = (int)(e->gs_launches++ & 1);
    const size_t rstride = (size_t)n * e->gs_rec;
    hipLaunchKernelGGL(dz::k_adapt_groups, dim3((gl * nq * p.d + gl * nbp + p.ld + 255) / 256, n), dim3(256), 0, e->stream, (const double*)e->d_PR, (const double*)e->d_PC, units_l, nq, p.d, p.ld, nb, nbp,
                       (const double*)e->d_x0ring, e->d_GS[par] + (size_t)e->rank * rstride, (long long)g, R1, (long long)e->ad_pr_stride, (long long)e->ad_pc_stride, (long long)e->gs_rec, 2 * R1);
    DZCK(launch_check("k_adapt_groups"));
    DZCK(exchange_rows(e, XK_SUMS, e->d_GS[par], 0, par, rstride));
    hipLaunchKernelGGL(dz::k_group_totals, dim3((nq * p.d + nb + p.ld + 255) / 256, n), dim3(256), 0, e->stream, (const double*)e->d_GS[par], e->world, rstride, gl, nq, p.d, p.ld, nb, nbp,
                       e->d_TOT, e->d_CNT, e->d_x0ring, (long long)g, R1, (long long)e->gs_rec, (long long)e->ad_tot_stride, (long long)e->ad_nbp, 2 * R1);
    DZCK(launch_check("k_group_totals"));
    e->gs_last_gen = (int64_t)g + n - 1;
    return adapt_dots(e, g, n);
}
// fused = the generation's persistent launch has already left its units' sums in d_PR / d_PC (adapt_unit_sums in its epilogue)
int adapt_generation(dz_engine* e, uint32_t g, int gc0, int ngc, bool fused = false, bool defer = false)
{
    ProfScope ps(e, PR_ADAPT);
    const dz::Params& p = e->p;
    // lockstep generations (schedule S2): reduction contract v3 -- one pass over the positions makes, per unit of 16 chains, the column
    // sums the standard deviations need AND the per-bin column sums of squared jumps; their totals; the update.
    // A single chain's update (Dream.astep, schedule S1): the reference's own chain-by-chain form, all rows in row order (numpy's).
    const bool single = ngc != p.N;
    const bool ring = e->c.adapt_lag > 0;
    const int slot = ring ? (int)(g % (uint32_t)e->ad_R1) : 0;
    double* x0_out = ring ? e->d_x0ring + (size_t)(g % (uint32_t)(2 * e->ad_R1)) * p.ld : nullptr;      // global chain 0's position after this generation
    if (!single && e->adapt_groups) {
        // sharded, whole groups per rank: own units' sums -> own groups' sums -> every rank's record everywhere -> the totals (the same
        // additions in the same order as k_adapt_totals makes them from all units: dz_kernels.h)
        const int units_l = p.nl / 16, gl = p.nl / 256, nq = 2 + p.ncr + p.ngamma, nb = p.ncr + p.ngamma, nbp = e->gs_nbp, par = (int)(g & 1u);
        if (!ring && g > 0 && e->gs_last_gen != (int64_t)g - 1) return fail("sharded adaptation: the previous burn-in generation left no shift row");
        if (!fused) {
            hipLaunchKernelGGL(dz::k_adapt_partials, dim3(units_l), dim3(1024), 0, e->stream, p, g, e->d_PR, e->d_PC, p.off / 16, adapt_shift(e, g));
            DZCK(launch_check("k_adapt_partials"));
        }
        hipLaunchKernelGGL(dz::k_adapt_groups, dim3((gl * nq * p.d + gl * nbp + p.ld + 255) / 256), dim3(256), 0, e->stream, (const double*)e->d_PR, (const double*)e->d_PC, units_l, nq, p.d, p.ld, nb, nbp,
                           (const double*)(p.cp_new + (size_t)p.off * p.ld), e->d_GS[par] + (size_t)e->rank * e->gs_rec, 0ll, 1, 0ll, 0ll, 0ll, 0);
        DZCK(launch_check("k_adapt_groups"));
        DZCK(exchange_rows(e, XK_SUMS, e->d_GS[par], 0, par));
        hipLaunchKernelGGL(dz::k_group_totals, dim3((nq * p.d + nb + p.ld + 255) / 256), dim3(256), 0, e->stream, (const double*)e->d_GS[par], e->world, e->gs_rec, gl, nq, p.d, p.ld, nb, nbp,
                           e->d_TOT + (size_t)slot * e->ad_tot_stride, e->d_CNT + (size_t)slot * e->ad_nbp, ring ? x0_out : e->d_shift[par], 0ll, 1, 0ll, 0ll, 0ll, 0);
        DZCK(launch_check("k_group_totals"));
        e->gs_last_gen = (int64_t)g;
        if (ring) return adapt_dots(e, g, 1);
        e->adapt_pending = true;
        return defer ? 0 : adapt_flush(e);
    }
    if (!single) {
        if (ring) HIPCK(hipMemcpyAsync(x0_out, p.cp_new, sizeof(double) * p.ld, hipMemcpyDeviceToDevice, e->stream));
        if (!fused) {
            hipLaunchKernelGGL(dz::k_adapt_partials, dim3((p.N + 15) / 16), dim3(1024), 0, e->stream, p, g, e->d_PR, e->d_PC, 0, ring ? adapt_shift(e, g) : (const double*)nullptr);
            DZCK(launch_check("k_adapt_partials"));
        }
        return adapt_finish(e, defer, g, 1);
    }
    const int strip = p.N, nstrips = 1;
    const dim3 b(128), gcol((p.d + 127) / 128, nstrips);
    double* partial1 = e->d_partial + (size_t)((p.N + 63) / 64) * p.ld;
    hipLaunchKernelGGL(dz::k_strip_partial, gcol, b, 0, e->stream, p.cp_new, p.N, p.d, p.ld, e->d_mean, 0, e->d_partial, strip, 1);
    hipLaunchKernelGGL(dz::k_strip_dev, gcol, b, 0, e->stream, p.cp_new, p.N, p.d, p.ld, (const double*)e->d_partial, nstrips, partial1, e->d_mean, strip, 1);
    NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_jump<NCH>, dim3((p.N + dz::JUMP_CHAINS - 1) / dz::JUMP_CHAINS), dim3(64 * dz::JUMP_CHAINS), 0, e->stream, p, g, gc0, ngc,
                                       (const double*)partial1, nstrips, e->d_sd, e->d_sdc, e->d_dl, e->d_dlg, e->d_binc, e->d_bing));
    hipLaunchKernelGGL(dz::k_adapt_update, dim3(1), dim3(1024), (size_t)(dz::ADAPT_CHUNK + dz::ADAPT_CHUNK / 64) * 24, e->stream, p,
                       (const double*)e->d_dl, (const double*)e->d_dlg, (const int*)e->d_binc, (const int*)e->d_bing, e->d_binsum);
    if (e->d_own_cr)      // the chains that updated the shared probabilities adopt them (Dream.py:375, :383, :409-415)
        hipLaunchKernelGGL(dz::k_own_probs, dim3((ngc + 63) / 64), dim3(64), 0, e->stream, p, gc0 - p.off, ngc, (const int*)e->d_binc, (const int*)e->d_bing, 0, e->d_own_cr, e->d_own_g);
    return launch_check("adaptation kernels");
}

// Can every try of a multi-try proposal set be impossible (log density -inf or nan)?  Not with the built-in likelihoods under flat or
// normal priors; yes with a host likelihood, and with a uniform prior unless hard boundaries keep every proposal inside its support.
// Those configurations take the multi-kernel path, which checks after the proposal set's evaluation and draws again (Dream.py:281-289).
bool redo_possible(const dz_engine* e)
{
    const dz::Params& p = e->p;
    if (p.k <= 1) return false;
    if (e->lk == LK_HOST) return true;
    if (e->lk == LK_MODULE && !e->lk_finite) return true;      // (a user's density may be -inf anywhere unless the caller promises otherwise: DZ_LIKE_ALWAYS_FINITE)
    for (size_t j = 0; j < e->h_pkind.size(); ++j) {
        if (e->h_pkind[j] != 2) continue;
        const bool covered = p.hard && j < e->h_mins.size() && e->h_mins[j] >= e->h_pa[j] && e->h_maxs[j] <= e->h_pa[j] + e->h_pb[j];
        if (!covered) return true;
    }
    return false;
}
// every lane waits for everything queued so far on every other lane
int join_all(dz_engine* e)
{
    if (e->nlanes <= 1) return 0;
    for (int s = 0; s < e->nlanes; ++s) HIPCK(hipEventRecord(e->lane_ev[s], e->lane_stream[s]));
    for (int s = 0; s < e->nlanes; ++s)
        for (int t = 0; t < e->nlanes; ++t) if (t != s) HIPCK(hipStreamWaitEvent(e->lane_stream[s], e->lane_ev[t], 0));
    return 0;
}
// Dream.py:281-289: while every try of a chain's proposal set is impossible, the set is generated again with the same decisions
// (snooker, CR, DE pairs, gamma level) and evaluated again.  Redraw round r >= 1 takes its point and dimension streams from the Philox
// key seed + r * DZ_REDRAW_KEY_STEP, evaluated in place (the precomputed draw table holds round 0).  A round:
// k_redo_flags, k_propose over the listed chains whose flag is still set, the listed sets packed into the (not yet used)
// reference-set buffers, evaluated there -- a host likelihood sees only those points -- and their log densities scattered back.
// The list comes from one read-back of the flags; up to four rounds are queued per read-back (chains that succeed in between drop
// out by their flag: same result as a round-by-round loop, fewer host round trips).  The reference's loop is unbounded; after
// DZ_MAX_REDRAWS rounds the step is a rejection (k_accept: no finite try), deviation D1.
int redraw_impossible_sets(dz_engine* e, uint32_t g, int lc0, int lnc, int sp0, int wpb, hipStream_t st)
{
    dz::Params& p = e->p;
    const int k = p.k;
    if (!e->d_redo) DZCK(ealloc(e, &e->d_redo, (size_t)p.nl));
    if (!e->d_redo_list) DZCK(ealloc(e, &e->d_redo_list, (size_t)p.nl));
    if (!e->h_redo && hipHostMalloc((void**)&e->h_redo, (size_t)p.nl, hipHostMallocDefault) != hipSuccess) { e->h_redo = nullptr; return fail("hipHostMalloc (redraw flags) failed"); }
    if (!e->h_redo_list && hipHostMalloc((void**)&e->h_redo_list, sizeof(int32_t) * (size_t)p.nl, hipHostMallocDefault) != hipSuccess) { e->h_redo_list = nullptr; return fail("hipHostMalloc (redraw list) failed"); }
    for (int round = 1; round <= DZ_MAX_REDRAWS;) {
        hipLaunchKernelGGL(dz::k_redo_flags, dim3((lnc + 255) / 256), dim3(256), 0, st, p, lc0, lnc, e->d_redo);
        DZCK(launch_check("k_redo_flags"));
        HIPCK(hipMemcpyAsync(e->h_redo, e->d_redo + lc0, (size_t)lnc, hipMemcpyDeviceToHost, st));
        HIPCK(hipStreamSynchronize(st));
        int n = 0;
        for (int t = 0; t < lnc; ++t) if (e->h_redo[t]) e->h_redo_list[n++] = lc0 + t;
        if (n == 0) break;
        HIPCK(hipMemcpyAsync(e->d_redo_list, e->h_redo_list, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
        // (a host likelihood costs more than a round trip: one round per read-back, no point is evaluated that did not have to be)
        const int batch = e->lk == LK_HOST ? 1 : std::min(DZ_MAX_REDRAWS - round + 1, n >= 16 ? 4 : (n >= 4 ? 3 : 2));
        for (int b = 0; b < batch; ++b, ++round) {
            if (b) hipLaunchKernelGGL(dz::k_redo_flags, dim3((lnc + 255) / 256), dim3(256), 0, st, p, lc0, lnc, e->d_redo);
            dz::Params pr = p;
            const uint64_t key = e->c.seed + (uint64_t)round * DZ_REDRAW_KEY_STEP;
            pr.k0 = (uint32_t)key; pr.k1 = (uint32_t)(key >> 32);
            pr.draws = nullptr; pr.redo_list = e->d_redo_list; pr.redo = e->d_redo;
            NCH_DISPATCH(e, DZ_KLAUNCH(e, PR_PROPOSE, st, dz::k_propose<NCH>, dim3((n * sp0 + wpb - 1) / wpb), dim3(64 * wpb), 0, pr, 0, g, (uint32_t)visible_rows(e), 0, n, sp0, 0, (int64_t)-1));
            DZCK(launch_check("propose(redraw)"));
            const size_t nd2 = (size_t)n * k * p.ld / 2;
            hipLaunchKernelGGL(dz::k_gather_sets, dim3((unsigned)((nd2 + 255) / 256)), dim3(256), 0, st, p, (const int32_t*)e->d_redo_list, n, p.R);
            DZCK(eval_logp(e, p.R, n * k, p.r_prior, p.r_like, st));
            hipLaunchKernelGGL(dz::k_scatter_logp, dim3((n * k + 255) / 256), dim3(256), 0, st, p, (const int32_t*)e->d_redo_list, n, (const double*)p.r_prior, (const double*)p.r_like);
            DZCK(launch_check("redraw gather/scatter"));
            e->redraw_rounds++;
        }
    }
    return 0;
}

// one transition of local chains [c0, c0+nc) at generation g.  Full range = a lockstep generation
// (schedule S2); a sub-range is the single-chain view used by Dream.astep: its end-of-generation
// updates (append, publish, adaptation) take effect immediately, as when the reference is driven
// round-robin in one process.
int one_generation(dz_engine* e, int c0, int nc, uint32_t g, bool traced, bool more_follow)
{
    dz::Params& p = e->p;
    const int k = p.k;
    const bool full = (c0 == 0 && nc == p.nl);
    if (!full && e->world > 1) return fail("single-chain stepping is not available on a sharded engine");
    if (!full && e->c.history_lag) return fail("single-chain stepping (Dream.astep) appends with immediate effect: history_lag must be 0");
    if (!full && e->c.adapt_lag) return fail("single-chain stepping (Dream.astep) updates the crossover probabilities with immediate effect: adapt_lag must be 0");
    DZCK(adapt_flush(e));               // (these kernels read the shared probabilities in place)
    DZCK(ensure_visible(e));
    const uint32_t Mv = (uint32_t)visible_rows(e);
    const int L = (full && e->lk != LK_HOST && !redo_possible(e)) ? e->nlanes : 1;     // the host-callback likelihood is synchronous anyway; redraw rounds use shared buffers
    if (e->need_join || !full) { DZCK(join_all(e)); e->need_join = false; }
    p.own_cr = nullptr; p.own_g = nullptr;
    if (!full && e->adapt) {
        // Dream.astep chain by chain: every Dream instance decides with its OWN copy of the crossover / gamma-level probabilities
        // (Dream.py:134, :143; refreshed at :375, :383, :409-415) -- the shared vectors may have moved on through other chains' updates
        if (!e->d_own_cr) {
            DZCK(ealloc(e, &e->d_own_cr, (size_t)p.nl * p.ncr)); DZCK(ealloc(e, &e->d_own_g, (size_t)p.nl * p.ngamma));
            e->own_init.assign((size_t)p.nl, 0);
        }
        for (int c = c0; c < c0 + nc; ++c) if (!e->own_init[c]) {
            hipLaunchKernelGGL(dz::k_own_probs, dim3(1), dim3(64), 0, e->stream, p, c, 1, (const int*)nullptr, (const int*)nullptr, 1, e->d_own_cr, e->d_own_g);
            e->own_init[c] = 1;
        }
        p.own_cr = e->d_own_cr; p.own_g = e->d_own_g;
    }
    if (full && g == 0 && e->adapt) {   // publish the start positions (Dream_shared_vars.current_positions)
        const size_t n = (size_t)p.nl * p.ld;
        hipLaunchKernelGGL(dz::k_copy_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, p.X, p.cp_new + (size_t)p.off * p.ld, n);
        DZCK(exchange_rows(e, XK_POS, p.cp_new, 0));
        if (e->c.adapt_lag > 0) HIPCK(hipMemcpyAsync(e->d_x0start, p.cp_new, sizeof(double) * p.ld, hipMemcpyDeviceToDevice, e->stream));      // global chain 0's start position
        DZCK(join_all(e));
    }
    p.draws = e->d_draws[g & 1]; p.draws_next = e->d_draws[(g + 1) & 1];
    p.ctl = e->d_ctl[g & 1]; p.ctl_next = e->d_ctl[(g + 1) & 1];
    const bool fused_in = full && e->pending_accept;                  // generation g-1's accept rides in front of this phase-0 kernel
    const bool need_draws = !fused_in && (!full || e->draws_gen != (int64_t)g);     // not prepared by the accept of the previous generation
    const bool append = (g % (uint32_t)p.thin) == 0;                           // Dream.py:360
    const bool publish = e->adapt && (int64_t)g < (int64_t)p.burnin + 1;      // Dream.py:364
    const int nrows = full ? p.N : nc;
    if (append && e->M + nrows > e->c.history_capacity) return fail("history capacity exceeded");
    if (publish) {
        if (full) { e->cp_idx = (e->cp_idx + 1) % 3; p.cp_prev = p.cp_new; p.cp_new = e->d_cp[e->cp_idx]; }      // (three buffers: a peer may be one generation ahead)
        else {   // the jump of a single chain is measured from its state at the start of the step
            const size_t n = (size_t)nc * p.ld;
            hipLaunchKernelGGL(dz::k_copy_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->stream, p.X + (size_t)c0 * p.ld, p.cp_prev + (size_t)(p.off + c0) * p.ld, n);
        }
    }
    // waves per block: 16 (one block fills a CU's 4 SIMDs evenly) once there is at least one such block per CU
    const int wpb = e->nch >= 4 ? 4 : (e->waves_per_block ? e->waves_per_block : ((nc / L) >= 16 * e->num_cu ? 16 : 4));    // (k_propose<NCH >= 4> is built for 4-wave blocks)
    // one wave per (chain, try) pays when a try is long and the chains are few: d > 512 (measured at 512 x 1000-D: +17%;
    // at d <= 200 the per-chain wave with its fused Metropolis step is faster)
    const int split = e->propose_split > 0 ? e->propose_split : (e->nch >= 8 ? k : 1);
    const int sp0 = std::max(1, std::min(k, split)), sp1 = std::max(1, std::min(k - 1, split));
    const int64_t slot = (traced && e->c.trace_capacity) ? e->ntrace : -1;
    // the Metropolis step of this generation can ride in front of the next generation's proposal kernel when
    // nothing shared changes in between (no history append, no published positions) and a generation follows
    // ld > 256, one DE pair, multi-try: one wave per (chain, try) streaming over the dimension chunks (dz_kernels.h)
    const bool streamed = e->stream_propose && e->nch >= 4 && p.depairs == 1 && k >= 3 && !redo_possible(e);     // (redraw rounds go through k_propose)
    const bool redo = redo_possible(e);         // (then the proposal set's evaluation is followed by a check on the host: nothing is deferred)
    const bool have_prop = streamed && full && e->stream_prop_gen == (int64_t)g;
    e->stream_prop_gen = -1;
    // ... and the streamed form of the same: accept(g) + proposal set(g+1) in one launch (k_accept_propose)
    const bool fuse_next = streamed && e->fuse_stream && full && more_follow && !append && !publish && !e->tempering && k <= 16;
    const bool defer = full && e->fuse && more_follow && !append && !publish && split == 1 && e->lk != LK_HOST && !e->tempering && !streamed && !redo;
    const int64_t zbase = full ? e->M : e->M - (int64_t)(p.off + c0);
    // large d: the row-tile sums of the likelihood product are added by the kernels that use them (no k_q_finish launches)
    // (round 5: also the per-chain proposal kernels of 128 < d <= 256 -- the reference example's own d = 200 --: two launches of six fewer)
    const bool qdefer = e->q_defer && e->lk == LK_MVN && p.ld / 16 > 8 && !e->force_big && (streamed || (full && k >= 3 && !redo_possible(e) && e->nch < 4 && e->nlanes == 1));
    p.qfin_p = nullptr; p.qfin_r = nullptr; p.qfin_nrt = (p.d + 15) / 16;
    for (int s = 0; s < L; ++s) {
        const int lc0 = c0 + (int)((int64_t)nc * s / L), lc1 = c0 + (int)((int64_t)nc * (s + 1) / L), lnc = lc1 - lc0;
        if (lnc <= 0) continue;
        hipStream_t st = e->lane_stream[s];
        if (need_draws)
            hipLaunchKernelGGL(dz::k_draws, dim3((lnc * p.nslots + 255) / 256), dim3(256), 0, st, p, g, lc0, lnc, e->d_draws[g & 1], e->d_ctl[g & 1]);
        if (streamed && have_prop) { }                      // made by the k_accept_propose launch of the generation before
        else if (streamed && !fused_in) DZ_KLAUNCH(e, PR_PROPOSE, st, dz::k_propose_stream, dim3((lnc * k + 3) / 4), dim3(256), 0, p, 0, g, Mv, lc0, lnc);
        else {
            if (fused_in) {      // (the Metropolis step of generation g - 1 in front: its reference set's row-tile sums may still be in the scratch array)
                p.qfin_r = e->pending_qfin_r; p.qfin_c0 = lc0; p.qfin_nc = lnc;
                NCH_DISPATCH(e, DZ_KLAUNCH(e, PR_PROPOSE, st, dz::k_propose<NCH>, dim3((lnc + wpb - 1) / wpb), dim3(64 * wpb), 0, p, 0, g, Mv, lc0, lnc, 1, 1, e->pending_slot));
                p.qfin_r = nullptr;
            }
            else { NCH_DISPATCH(e, DZ_KLAUNCH(e, PR_PROPOSE, st, dz::k_propose<NCH>, dim3((lnc * sp0 + wpb - 1) / wpb), dim3(64 * wpb), 0, p, 0, g, Mv, lc0, lnc, sp0, 0, (int64_t)-1)); }
        }
        DZCK(launch_check("propose"));
        DZCK(eval_logp(e, p.P + (size_t)lc0 * k * p.ld, lnc * k, p.p_prior + (size_t)lc0 * k, p.p_like + (size_t)lc0 * k, st, qdefer));
        if (redo) DZCK(redraw_impossible_sets(e, g, lc0, lnc, sp0, wpb, st));
        if (k > 1) {
            if (qdefer) { p.qfin_p = e->last_qpart; p.qfin_c0 = lc0; p.qfin_nc = lnc; }
            if (streamed) DZ_KLAUNCH(e, PR_PROPOSE, st, dz::k_propose_stream, dim3((lnc * (k - 1) + 3) / 4), dim3(256), 0, p, 1, g, Mv, lc0, lnc);
            else {
                NCH_DISPATCH(e, DZ_KLAUNCH(e, PR_PROPOSE, st, dz::k_propose<NCH>, dim3((lnc * sp1 + wpb - 1) / wpb), dim3(64 * wpb), 0, p, 1, g, Mv, lc0, lnc, sp1, 0, (int64_t)-1));
            }
            DZCK(launch_check("propose(ref)"));
            p.qfin_p = nullptr;
            DZCK(eval_logp(e, p.R + (size_t)lc0 * (k - 1) * p.ld, lnc * (k - 1), p.r_prior + (size_t)lc0 * (k - 1), p.r_like + (size_t)lc0 * (k - 1), st, qdefer));
            if (qdefer) { p.qfin_r = e->last_qpart; p.qfin_c0 = lc0; p.qfin_nc = lnc; }
        }
        if (fuse_next) {
            const size_t lds = sizeof(double) * (size_t)p.ld + 64 * sizeof(uint4) + sizeof(dz::ChainCtl);
            NCH_DISPATCH(e, DZ_KLAUNCH(e, PR_ACCEPT, st, dz::k_accept_propose<NCH>, dim3(lnc), dim3(64 * k), lds, p, g, zbase, lc0, lnc, slot, Mv));
            DZCK(launch_check("accept + propose"));
        } else if (!defer) {
            NCH_DISPATCH(e, DZ_KLAUNCH(e, PR_ACCEPT, st, dz::k_accept<NCH>, dim3((lnc + wpb - 1) / wpb), dim3(64 * wpb), 0, p, g, zbase, lc0, lnc, slot, append ? 1 : 0, publish ? 1 : 0, (full && !publish) ? 1 : 0));
            DZCK(launch_check("accept"));
        }
        p.qfin_r = nullptr;
    }
    if (fuse_next) e->stream_prop_gen = (int64_t)g + 1;
    e->pending_accept = defer; e->pending_slot = defer ? slot : -1;
    e->pending_qfin_r = (defer && qdefer && k > 1) ? e->last_qpart : nullptr;      // (one chain group when deferred sums are in play: see L below)
    e->draws_gen = (full && !publish) ? (int64_t)g + 1 : -1;   // while adapting, next generation's decisions must wait for the new probabilities
    if (publish || append) {         // shared state changed: the lanes meet before anything reads it
        if (L > 1) DZCK(join_all(e));
        if (publish) {
            if (full && !e->adapt_groups) DZCK(exchange_rows(e, XK_POS, p.cp_new, 0));      // (ranks that own whole groups exchange their groups' sums: adapt_generation)
            DZCK(adapt_generation(e, g, full ? 0 : p.off + c0, full ? p.N : nc));
        }
        if (append) { if (full) DZCK(exchange_rows(e, XK_Z, p.Z, (size_t)e->M)); e->M += nrows; e->napp += 1; }
        e->need_join = true;
    }
    if (e->tempering && full) {      // temperature swap (core.py:185-221): after every chain's step and the updates above
        if (L > 1) { DZCK(join_all(e)); e->need_join = true; }
        NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_pt_swap<NCH>, dim3(1), dim3(64), 0, e->stream, p, g, slot, publish ? 1 : 0));
        DZCK(launch_check("k_pt_swap"));
    }
    e->last_variant = "multi-kernel path";
    for (int c = c0; c < c0 + nc; ++c) e->gen_c[c] = (int64_t)g + 1;
    if (full) e->gen = (int64_t)g + 1;
    if (slot >= 0) e->ntrace++;
    return 0;
}

// ---- persistent generation kernel (dz_megakernel.h) -------------------------------------------------
// chain states, gamma table and decisions ride in LDS next to the matrix whenever that fits (the packed triangle at
// d=100 leaves room; the dense square does not)
// chains (= waves) per block: 16 once that still gives every CU a block, else 8 or 4 (C2: 1024 chains -> 256 blocks of 4)
// -> chains per block of the (first) launch; split_c / ch_b: a generation as TWO launches -- whole rounds of ch-chain blocks for the local chains
// [0, split_c), the remainder in blocks of ch_b chains (split_c == nl: one launch)
struct MegaPlan { int ch, split_c, ch_b; };
MegaPlan mega_plan(const dz_engine* e)
{
    const dz::Params& p = e->p;
    if (e->mega_ch) return MegaPlan{e->mega_ch, p.nl, 0};
    // One block per CU is resident (LDS), so a launch takes ceil(blocks / CUs) rounds of a block's time; measured at 100-D a block of 12
    // chains needs 0.82 (round 6), one of 8 chains 0.67 and one of 4 chains (four waves per chain) 0.49 of the time of a block of 16.  The
    // smallest product wins, the larger block on a tie: 4096 chains -> 16, 3072 -> 12 (round 6: 256 blocks of 12 -- 677 M proposals/s; as
    // 192 blocks of 16, a quarter of the chip idle, 561), 2048 -> 8, 1024 -> 4 ... among the block sizes whose LDS layout fits at all (round
    // 4): at 113..128 dimensions the point tiles of 16 chains no longer fit next to the matrix, those of 8 do.
    // A chain count just above whole rounds -- 5000 chains: 313 blocks of 16 -- pays a whole round for its remainder; the remainder can go in
    // a SECOND launch of smaller blocks (whole rounds of the large blocks first).  The second launch's blocks start while the first one's
    // last blocks drain, worth about 0.12 of a 16-chain block's time (5000 chains: 16 + 8 measured 39.8 us per generation, modelled without
    // the overlap 46.4; 12 + 8: 40.0; 417 blocks of 12 in two rounds: 43.1).
    const int ncu = e->num_cu > 0 ? e->num_cu : 1;
    static const double c12 = getenv("DZ_COST12") ? atof(getenv("DZ_COST12")) : 0.82;
    const int cand[4] = {16, 12, 8, 4};
    const double cost[4] = {1.0, c12, 0.67, 0.49};
    const bool need_x = p.hard || p.have_prior || p.depairs > 1;          // (the full-code instantiations keep the states in LDS)
    const bool k1 = p.k == 1;
    bool fits[4];
    for (int i = 0; i < 4; ++i)
        fits[i] = !(cand[i] == 12 && k1) &&
                  sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, p.tri != 0, need_x, cand[i], need_x && p.pb_lds != 0).total <= (size_t)160 * 1024;
    MegaPlan best{dz::MEGA_CHAINS, p.nl, 0}; double tb = -1.0;
    for (int i = 0; i < 4; ++i) {
        if (!fits[i]) continue;
        const int ch = cand[i], blocks = (p.nl + ch - 1) / ch, rounds = (blocks + ncu - 1) / ncu;
        const double t = rounds * cost[i];
        if (tb < 0.0 || t < tb - 1e-9) { best = MegaPlan{ch, p.nl, 0}; tb = t; }
        if (!e->mega_split || k1 || ch < 12 || blocks <= ncu || blocks % ncu == 0) continue;
        const int full = (blocks / ncu) * ncu * ch, r = p.nl - full;
        for (int j = i + 1; j < 4; ++j) {
            if (!fits[j]) continue;
            const int rb = ((r + cand[j] - 1) / cand[j] + ncu - 1) / ncu;
            const double t2 = (blocks / ncu) * cost[i] + rb * cost[j] - 0.12;
            if (t2 < tb - 1e-9) { best = MegaPlan{ch, full, cand[j]}; tb = t2; }
        }
    }
    return best;
}
int mega_chains(const dz_engine* e) { return mega_plan(e).ch; }
size_t mega_lds_bytes(const dz_engine* e, bool xlds)
{
    return sizeof(double) * (size_t)dz::mega_layout(e->p.d, e->p.k, e->p.ld / 16, e->p.ncr, e->p.ngamma, e->p.tri != 0, xlds, mega_chains(e), e->p.pb_lds != 0).total;
}
// the full-code instantiations stage the per-dimension prior / boundary constants in LDS when that still fits (Params::pb_lds)
void mega_set_pb_lds(dz_engine* e)
{
    const bool pb = e->p.hard || e->p.have_prior || e->p.depairs > 1 || (redo_possible(e) && e->lk == LK_MVN && e->p.k > 1 && e->mega_redo_on);
    e->p.pb_lds = 0;
    {   // uniform / flat priors whose supports contain the hard boundaries' box: the log prior of every proposal is one constant
        bool cst = e->p.have_prior && e->p.prior_nonormal && e->p.hard && !getenv("DZ_PRIOR_CONST_OFF");
        for (size_t j = 0; cst && j < e->h_pkind.size(); ++j)
            if (e->h_pkind[j] == 2 && !(j < e->h_mins.size() && e->h_mins[j] >= e->h_pa[j] && e->h_maxs[j] <= e->h_pa[j] + e->h_pb[j])) cst = false;
        e->p.prior_const = cst ? 1 : 0;
    }
    if (!pb) return;
    e->p.pb_lds = 1;
    if (mega_lds_bytes(e, true) > (size_t)160 * 1024) e->p.pb_lds = 0;
}
bool mega_xlds(const dz_engine* e) { return mega_lds_bytes(e, true) <= (size_t)160 * 1024; }
// the barrier-free variant for the mixture likelihood (k_generations_mix)
bool mega_mix_eligible(const dz_engine* e)
{
    const dz::Params& p = e->p;
    const bool pbm = p.hard || p.have_prior || p.depairs > 1;
    return e->mega && (e->lk == LK_MIX || (e->lk == LK_MODULE && e->lk_gen_fn[(p.ld > 128 ? 2 : 0) + (pbm ? 1 : 0)])) && !redo_possible(e) && (!pbm || e->mega_mix_pb) &&
           p.ld <= (e->mega_mix_wide ? 256 : 128) && (p.k == 1 || p.k >= 3) && p.k <= dz::MAXK &&
           p.nslots <= 64 && p.J <= 32;
}
// redraw rounds inside the persistent kernel: the instantiations with the full proposal code, multi-try, device MVN likelihood
bool mega_redo(const dz_engine* e) { return redo_possible(e) && e->lk == LK_MVN && e->p.k > 1 && e->mega_redo_on; }
// 128 < d <= 256 (the reference example's d = 200): k_generations_d2 -- configurations with the triangular factor (what
// MVNormalLogLike builds) whose point tiles of 16 chains fit LDS without the matrix (it is read from L2): d <= ~230 at 5 tries.  Measured and
// left on the multi-kernel path: 8 chains per block (d = 256: 131 against 123 us per generation) and the dense matrix (its 16-chain
// instantiation spills a hundred registers; at 8 chains per block 156 against 148 us at d = 200).
// (round 6) 16 chains per block whose k tries' point tiles do not fit but k - 1 do (229..256 dimensions at 5 tries): the proposal set in two passes
// (k_generations_d2<.., SP>); 3..15 tries, 128 < d
bool mega_d2_two_pass(const dz_engine* e, int ch = 16)
{
    const dz::Params& p = e->p;
    if (p.ld <= 128 || p.k < 3 || p.nslots > 64 || (getenv("DZ_MEGA_D2_SP") && atoi(getenv("DZ_MEGA_D2_SP")) == 0)) return false;
    if (sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, false, ch, false, false, true).total <= (size_t)160 * 1024) return false;
    return sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, false, ch, false, false, true, 0, true).total <= (size_t)160 * 1024;
}
int mega_d2_chains(const dz_engine* e)
{
    const dz::Params& p = e->p;
    if (!e->mega || !e->mega_d2 || e->lk != LK_MVN || p.ld > 256) return 0;
    if (p.ld < 80 && p.nslots <= 64) return 0;
    // (round 6) more than 15 tries: a generation's draw slots exceed a wave's lanes, the classic kernels do not take them; this one keeps one phase's
    // slots at a time (k npt <= 64: up to 32 tries with one or two DE pairs)
    const bool bigk = p.nslots > 64;
    if (bigk && (p.k * p.npt > 64 || p.k > dz::MAXK || (getenv("DZ_MEGA_BIGK") && atoi(getenv("DZ_MEGA_BIGK")) == 0))) return 0;
    if (p.ld <= 128 && !bigk) {      // d <= 128: only where the classic kernel's 16-chain layout (matrix in LDS) does not fit -- it then runs 8 chains per block (113..128
                            // dimensions at 5 tries, 100 dimensions at 8 or more)
        const bool pbx = p.hard || p.have_prior || p.depairs > 1;
        const size_t classic = sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, pbx, 16, pbx && p.pb_lds != 0).total;
        if (classic <= (size_t)160 * 1024 || p.k == 1) return 0;
    }
    if (e->mega_d2 == 1 && p.nl <= 1024) return 0;      // (64 blocks or fewer leave three CUs in four idle: 57 against 52 us per generation at 1024 x 200-D; DZ_MEGA_D2=2 forces it)
    if (redo_possible(e)) return 0;      // (whole proposal sets that can be impossible: the multi-kernel path's redraw rounds)
    if (p.k != 1 && p.k < 3) return 0;
    if (!p.tri || !p.Mtp) return 0;
    const size_t lds = sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, false, 16, false, false, true).total;
    if (lds <= (size_t)160 * 1024) return 16;
    if (mega_d2_two_pass(e)) return 16;
    // round 6: 8 chains x 2 waves per block where the point tiles of 16 chains do not fit (128 < d: 229..256 dimensions at 5 tries) -- multi-try only
    if (p.ld <= 128 && !bigk) {      // (13..15 tries at 100 dimensions: the classic kernel would run 4 chains x 4 waves -- 603 M proposals/s at k = 15 against 693 for 8 x 2 here at k = 16)
        const bool pbx = p.hard || p.have_prior || p.depairs > 1;
        if (sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, pbx, 8, pbx && p.pb_lds != 0).total <= (size_t)160 * 1024) return 0;
    }
    if (p.k >= 3 && !(getenv("DZ_MEGA_D2_W2") && atoi(getenv("DZ_MEGA_D2_W2")) == 0) &&
        sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, false, 8, false, false, true).total <= (size_t)160 * 1024) return 8;
    if (p.k >= 3 && !(getenv("DZ_MEGA_D2_W2") && atoi(getenv("DZ_MEGA_D2_W2")) == 0) && mega_d2_two_pass(e, 8)) return 8;      // 8 x 2 with the two-pass set (256 dimensions at 8 tries)
    // ... and 4 chains x 4 waves where not even those fit (24..32 tries at 100 dimensions, 20..32 at 128, 8 at 256): four rounds of 1024 blocks at 4096
    // chains, still ahead of the multi-kernel path
    if (p.k >= 4 && !(getenv("DZ_MEGA_D2_W4") && atoi(getenv("DZ_MEGA_D2_W4")) == 0) &&
        sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, true, false, 4, false, false, true).total <= (size_t)160 * 1024) return 4;
    return 0;
}
bool mega_eligible(dz_engine* e)
{
    mega_set_pb_lds(e);
    const dz::Params& p = e->p;
    if (mega_d2_chains(e) > 0) return true;
    if (redo_possible(e) && !mega_redo(e)) return false;
    if (mega_mix_eligible(e)) return true;
    if ((p.hard || p.have_prior || p.depairs > 1 || mega_redo(e)) && !mega_xlds(e)) return false;      // (the full-code instantiations keep the states in LDS)
    return e->mega && e->lk == LK_MVN && p.ld <= 128 && (p.k == 1 || p.k >= 3) && p.k <= dz::MAXK &&
           p.nslots <= 64 && (!p.tri || p.Mtp) && mega_lds_bytes(e, false) <= (size_t)160 * 1024;
}
// number of generations, starting at g, that one launch may cover: none of them publishes positions
// (crossover burn-in), and only the last one may append to the history
// The persistent kernels read Params through a pointer: the device copy is refreshed when a field THEY read has changed (the buffers
// that rotate from generation to generation -- published positions, the draw tables of the multi-kernel path -- are not among them).
int upload_params(dz_engine* e)
{
    dz::Params q = e->p;
    q.cp_prev = nullptr; q.cp_new = nullptr; q.draws = nullptr; q.draws_next = nullptr; q.ctl = nullptr; q.ctl_next = nullptr;
    q.own_cr = nullptr; q.own_g = nullptr; q.redo = nullptr; q.redo_list = nullptr;
    q.cr_probs = nullptr; q.cr_delta = nullptr; q.cr_n = nullptr; q.g_probs = nullptr; q.g_delta = nullptr; q.g_n = nullptr;     // (the persistent kernels get the state through Publish::sh)
    if (e->params_uploaded && memcmp(&e->p_shadow, &q, sizeof(dz::Params)) == 0) return 0;
    // (through a staging copy that outlives the call: the source of an asynchronous copy from pageable memory must not be a local)
    memcpy(&e->p_shadow, &q, sizeof(dz::Params));
    HIPCK(hipMemcpyAsync(e->d_params, &e->p_shadow, sizeof(dz::Params), hipMemcpyHostToDevice, e->stream));
    e->params_uploaded = true;
    return 0;
}
bool publishing(const dz_engine* e, uint32_t g) { return e->adapt && (int64_t)g < (int64_t)e->p.burnin + 1; }      // Dream.py:364
// History appends one launch of the persistent kernels may hold (the last one at its end), a: a generation behind i appends samples the rows
// of the first i - lag of them, and no generation of a launch may sample rows the launch itself writes (blocks do not meet): a <= lag + 1.
// Several GPUs (round 6): the rows the launch's own chains append travel AFTER the launch -- by an all-gather or through the host at once
// (then a <= lag + 1 as well: the exchange stays exposed, between two launches), or by the copy engines while the next launch computes: that
// launch must not sample them either, so 2 a - 1 <= lag (lag 1: one append per launch; lag 3: two, the 20 generations per launch one GPU runs).
int mega_appends_per_launch(const dz_engine* e)
{
    const int lag = e->c.history_lag;
    const int a = (e->world > 1 && e->peer_on) ? (lag + 1) / 2 : lag + 1;
    return std::max(1, std::min(a, e->mega_segs));
}
// adapt_lag >= 1: can a launch hold several burn-in generations?  One GPU, and a kernel instantiation that makes its block's unit sums
// generation by generation (blocks of 16 chains = one unit): the mixture kernel's MG instantiations
size_t mix_multi_lds(const dz_engine* e)
{
    const dz::Params& p = e->p;
    const size_t LDP = (size_t)4 * ((p.d + 3) / 4) + 1;
    return sizeof(double) * ((size_t)16 * dz::mega_mix_wave_doubles(p.d, p.k, p.J) + (size_t)e->ad_R1 * e->ad_nbp + 64 * LDP + 32);
}
size_t mvn_multi_lds(const dz_engine* e)
{
    const dz::Params& p = e->p;
    return sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, p.ld / 16, p.ncr, p.ngamma, p.tri != 0, true, 16, p.pb_lds != 0, true, false, e->ad_R1).total;
}
// -> 1: the kernel makes its blocks' unit sums generation by generation (the MG instantiations: blocks of 16 chains = one unit);  2: any other persistent
// kernel (blocks of 12 / 8 / 4 chains, split generations, 128 < d <= 256, multitry off, redraw rounds) -- every generation's positions go to a ring of
// published positions and ONE k_adapt_partials_ring launch behind it makes the sums of all its generations;  0: one burn-in generation per launch
int burnin_multi(const dz_engine* e)
{
    if (!e->ad_multi || !e->mega_burnin || e->tempering) return 0;
    const dz::Params& p = e->p;
    const size_t tab = sizeof(double) * (size_t)e->ad_R1 * e->ad_nbp, cap = (size_t)160 * 1024;
    const bool ring_ok = e->ad_ring && sizeof(double) * (size_t)(e->c.adapt_lag + 2) * p.N * p.ld <= ((size_t)8 << 30);
    if (e->lk == LK_MIX || e->lk == LK_MODULE) {
        if (!mega_mix_eligible(e)) return 0;
        if (e->lk == LK_MIX && e->adapt_fused && p.k >= 3 && p.ld <= 128 && mix_multi_lds(e) <= cap) return 1;
        return (ring_ok && sizeof(double) * (size_t)dz::MIXW * dz::mega_mix_wave_doubles(p.d, p.k, p.J) + tab <= cap) ? 2 : 0;
    }
    if (e->lk != LK_MVN || !e->mega) return 0;
    const int nrt = p.ld / 16;
    if (const int chd = mega_d2_chains(e)) {
        const bool sp = (chd == 16 && mega_d2_two_pass(e)) || (chd == 8 && mega_d2_two_pass(e, 8));
        return (ring_ok && sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, nrt, p.ncr, p.ngamma, p.tri != 0, false, chd, false, false, true, e->ad_R1, sp).total <= cap) ? 2 : 0;
    }
    if (p.ld > 128 || (p.k != 1 && p.k < 3) || p.k > dz::MAXK || p.nslots > 64 || (p.tri && !p.Mtp)) return 0;
    const MegaPlan plan = mega_plan(e);
    const bool pb = p.hard || p.have_prior || p.depairs > 1 || mega_redo(e);
    if (e->adapt_fused && p.k >= 3 && !redo_possible(e) && plan.ch == 16 && plan.split_c == p.nl && (pb || mega_xlds(e)) && mvn_multi_lds(e) <= cap) return 1;
    if (!ring_ok || (redo_possible(e) && !mega_redo(e))) return 0;
    const bool xl = pb ? true : mega_xlds(e);
    for (int part = 0; part < 2; ++part) {
        const int chp = part ? plan.ch_b : plan.ch;
        if (part && plan.split_c == p.nl) break;
        if (sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, nrt, p.ncr, p.ngamma, p.tri != 0, xl, chp, p.pb_lds != 0, false, false, e->ad_R1).total > cap) return 0;
    }
    return 2;
}
int mega_segment(const dz_engine* e, uint32_t g, int64_t remaining)
{
    // crossover burn-in: the positions are published and the probabilities adapted after every generation -- one generation per launch, or
    // (adapt_lag = L >= 1, one GPU, a kernel that makes its units' sums generation by generation: burnin_multi) up to L + 1 of them
    const bool pub0 = publishing(e, g);
    if (e->tempering) return remaining > 0 ? 1 : 0;       // parallel tempering: a temperature swap follows every generation (core.py:185-221)
    if (pub0 && !burnin_multi(e)) return (e->mega_burnin && remaining > 0) ? 1 : 0;
    if (pub0 && !e->mega_burnin) return 0;
    const int maxg = pub0 ? std::min(e->c.adapt_lag + 1, e->mega_max_gen) : e->mega_max_gen;
    // A launch ends with a history append -- unless the rows it writes are not sampleable yet anyway (history_lag >= 1, every lagged append
    // already made): the kernel then makes the append itself and runs on, up to mega_appends_per_launch of them (the generations behind the
    // j-th one sample j * N more rows: all of them written -- and, sharded, received -- before the launch).
    const int lag = e->c.history_lag;
    int segs = (lag >= 1 && e->napp >= lag) ? mega_appends_per_launch(e) : 1;
    segs = (int)std::max<int64_t>(1, std::min<int64_t>(segs, (e->c.history_capacity - e->M) / std::max(1, e->p.N)));      // (an archive sized to the last append: no launch asks for more rows than one append at a time would)
    int n = 0, apps = 0;
    for (uint32_t gg = g; n < remaining && n < maxg; ++gg) {
        if (publishing(e, gg) != pub0) break;
        ++n;
        if (gg % (uint32_t)e->p.thin == 0 && ++apps >= segs) break;
    }
    return n;
}
int run_mega_segment(dz_engine* e, uint32_t g, int n, bool mega_follows)
{
    dz::Params& p = e->p;
    // the history appends of generations g .. g + n - 1 (one at the end, or -- mega_segment -- up to history_lag + 1, the last one at the end)
    int napps = 0, seg0 = 0;
    for (int i = 0; i < n; ++i) if ((g + (uint32_t)i) % (uint32_t)p.thin == 0) { if (!napps) seg0 = i + 1; ++napps; }
    const bool append_last = napps > 0;
    {   // no generation of the launch may sample rows the launch itself writes: at most history_lag appends in front of any generation
        const bool ends_with_one = ((g + (uint32_t)n - 1) % (uint32_t)p.thin) == 0;
        const int allowed = mega_appends_per_launch(e);
        if (napps > 1 && (e->napp < e->c.history_lag || napps > allowed - (ends_with_one ? 0 : 1))) return fail("internal: too many history appends in one launch");
    }
    if (append_last && e->M + (int64_t)napps * p.N > e->c.history_capacity) return fail("history capacity exceeded");
    DZCK(join_all(e));
    {   // the rows the launch samples have arrived: its last generation may sit behind all but the last of its own appends
        const bool ends_with_one = ((g + (uint32_t)n - 1) % (uint32_t)p.thin) == 0;
        DZCK(ensure_visible(e, std::max(0, napps - (ends_with_one ? 1 : 0))));
    }
    const bool publish = publishing(e, g);            // (then n == 1: mega_segment -- or, adapt_lag >= 1 and burnin_multi, up to adapt_lag + 1)
    const bool ring = e->c.adapt_lag > 0;
    const int mmode = (publish && ring) ? burnin_multi(e) : 0;
    const bool multi = mmode == 1;      // the launch applies the pending updates itself and makes its units' sums generation by generation
    const bool rmulti = mmode == 2;     // ... or leaves every generation's positions in the ring; the sums follow (k_adapt_partials_ring)
    if (ring && !mmode) DZCK(adapt_apply_due(e, (int64_t)g));   // (every generation of such a launch decides with the same state: one burn-in generation, or none)
    if (publish && n > 1 && !mmode) return fail("internal: several burn-in generations in a launch that cannot hold them");
    if (g == 0 && e->adapt) {   // publish the start positions (Dream_shared_vars.current_positions)
        const size_t nn = (size_t)p.nl * p.ld;
        hipLaunchKernelGGL(dz::k_copy_rows, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, e->stream, p.X, p.cp_new + (size_t)p.off * p.ld, nn);
        DZCK(exchange_rows(e, XK_POS, p.cp_new, 0));
        if (e->c.adapt_lag > 0) HIPCK(hipMemcpyAsync(e->d_x0start, p.cp_new, sizeof(double) * p.ld, hipMemcpyDeviceToDevice, e->stream));      // global chain 0's start position
    }
    if (publish && !rmulti) { e->cp_idx = (e->cp_idx + 1) % 3; p.cp_prev = p.cp_new; p.cp_new = e->d_cp[e->cp_idx]; }
    const size_t pos_stride = (size_t)p.N * p.ld; const int RP = e->c.adapt_lag + 2;
    if (rmulti) {      // the ring of published positions; the positions before this launch's first generation are the chains' states now
        if (!e->d_posring) { DZCK(ealloc(e, &e->d_posring, (size_t)RP * pos_stride)); DZCK(ealloc(e, &e->d_PG, (size_t)e->ad_R1 * e->ad_nbp)); }
        if (e->posring_gen != (int64_t)g - 1)
            HIPCK(hipMemcpyAsync(e->d_posring + (size_t)(((int64_t)g - 1 + RP) % RP) * pos_stride + (size_t)p.off * p.ld, p.X, sizeof(double) * (size_t)p.nl * p.ld, hipMemcpyDeviceToDevice, e->stream));
    }
    const int64_t slot0 = e->c.trace_capacity ? e->ntrace : -1;
    dz::Publish pub; pub.to = publish ? p.cp_new : nullptr; pub.shift = nullptr; pub.PR = nullptr; pub.PC = nullptr;
    pub.sh = p.cr_probs; pub.sh_out = nullptr; pub.TOT = nullptr; pub.CNT = nullptr; pub.c0 = 0; pub.c1 = p.nl;
    const bool applies = e->adapt_pending;      // the previous generation's adaptation totals: applied by this launch's prologue, new state into the other copy
    if (applies) { pub.TOT = e->d_TOT; pub.CNT = e->d_CNT; pub.sh_out = e->d_shared + (size_t)(e->sh_cur ^ 1) * 3 * (p.ncr + p.ngamma); }
    if (multi) {      // adapt_lag >= 1: the pending updates, the rings, the table of probabilities per generation (Publish, dz_kernels.h)
        pub.multi = 1; pub.lag = e->c.adapt_lag; pub.burnin = p.burnin; pub.nbp = e->ad_nbp; pub.pend0 = e->ad_applied + 1; pub.pend1 = e->ad_made + 1;
        pub.DOT = e->d_DOT; pub.CNTR = e->d_CNT; pub.x0ring = e->d_x0ring; pub.x0start = e->d_x0start; pub.pr_stride = (long long)e->ad_pr_stride; pub.pc_stride = (long long)e->ad_pc_stride;
        pub.PR = e->d_PR; pub.PC = e->d_PC;
        pub.sh_out = e->d_shared + (size_t)(e->sh_cur ^ 1) * 3 * (p.ncr + p.ngamma);
    }
    if (rmulti) {
        pub.multi = 2; pub.lag = e->c.adapt_lag; pub.burnin = p.burnin; pub.nbp = e->ad_nbp; pub.pend0 = e->ad_applied + 1; pub.pend1 = e->ad_made + 1;
        pub.DOT = e->d_DOT; pub.CNTR = e->d_CNT; pub.to = e->d_posring; pub.pos_stride = (long long)pos_stride; pub.PG = e->d_PG;
        pub.sh_out = e->d_shared + (size_t)(e->sh_cur ^ 1) * 3 * (p.ncr + p.ngamma);
    }
    auto launched = [&]() {
        if (applies) { e->sh_cur ^= 1; point_shared(e); e->adapt_pending = false; }
        if (multi || rmulti) { e->sh_cur ^= 1; point_shared(e); e->ad_applied = std::max(e->ad_applied, adapt_due(e, (int64_t)g + n - 1)); }
    };
    // crossover burn-in on one GPU: a block of 16 chains is one unit of the adaptation's column sums (contract v3) and makes them itself
    bool fused = false;
    auto fuse_adapt = [&]() { fused = true; pub.shift = adapt_shift(e, g); pub.PR = e->d_PR; pub.PC = e->d_PC; };
    auto after_launch = [&]() -> int {      // end of the generation(s): positions -> adaptation -> history append (schedule S2)
        const bool sharded = e->world > 1;      // (then the ranks own whole groups: their groups' sums travel, one exchange per launch -- adapt_finish_groups)
        if (multi) { ProfScope ps(e, PR_ADAPT); DZCK(sharded ? adapt_finish_groups(e, g, n) : adapt_finish(e, false, g, n, true)); }      // the totals and dot products of the launch's n generations
        else if (rmulti) {
            ProfScope ps(e, PR_ADAPT);
            hipLaunchKernelGGL(dz::k_adapt_partials_ring, dim3(sharded ? p.nl / 16 : (p.N + 15) / 16, n), dim3(1024), 0, e->stream, p, g, e->ad_R1, (const double*)e->d_posring, (long long)pos_stride,
                               (const double*)e->d_PG, e->ad_nbp, e->d_PR, e->d_PC, (long long)e->ad_pr_stride, (long long)e->ad_pc_stride, e->d_x0ring, (const double*)e->d_x0start, p.off / 16);
            DZCK(launch_check("k_adapt_partials_ring"));
            DZCK(sharded ? adapt_finish_groups(e, g, n) : adapt_finish(e, false, g, n, true));
            e->posring_gen = (int64_t)g + n - 1;
            p.cp_new = e->d_posring + (size_t)(e->posring_gen % RP) * pos_stride;      // (what a later launch that is not of this kind measures its jumps from)
        }
        else if (publish) {
            if (!e->adapt_groups) DZCK(exchange_rows(e, XK_POS, p.cp_new, 0));      // (ranks that own whole groups exchange their groups' sums instead: adapt_generation)
            DZCK(adapt_generation(e, g, 0, p.N, fused, mega_follows));
        }
        for (int a = 0; a < napps; ++a) { DZCK(exchange_rows(e, XK_Z, p.Z, (size_t)e->M)); e->M += p.N; e->napp += 1; }
        if (e->tempering) {          // (then n == 1) the temperature swap: after every chain's step and the updates above, as on the multi-kernel path
            NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_pt_swap<NCH>, dim3(1), dim3(64), 0, e->stream, p, g, slot0, publish ? 1 : 0));
            DZCK(launch_check("k_pt_swap"));
        }
        e->need_join = true;
        e->draws_gen = -1;
        e->gen = (int64_t)g + n;
        for (auto& gcv : e->gen_c) gcv = e->gen;
        return 0;
    };
    if (e->lk == LK_MIX || e->lk == LK_MODULE) {
        DZCK(upload_params(e));
        int mw = dz::MIXW;
        const size_t lds_probs = sizeof(double) * (size_t)((p.ncr + p.ngamma + 1) & ~1);
        const size_t lds_xo = sizeof(double) * (size_t)16 * (4 * ((p.d + 3) / 4) + 1);
        if (multi) mw = 16;
        else if (publish && !ring && e->adapt_fused && (e->world == 1 || e->adapt_groups) && p.k >= 3 && p.ld <= 128 && sizeof(double) * (size_t)16 * dz::mega_mix_wave_doubles(p.d, p.k, p.J) + lds_probs + lds_xo <= (size_t)160 * 1024) { mw = 16; fuse_adapt(); }
        const dim3 gridm((p.nl + mw - 1) / mw), blockm(64 * mw);
        const size_t ldsm = multi ? mix_multi_lds(e) : sizeof(double) * (size_t)mw * dz::mega_mix_wave_doubles(p.d, p.k, p.J) + (rmulti ? sizeof(double) * (size_t)e->ad_R1 * e->ad_nbp : lds_probs) + (fused ? lds_xo : 0);
        const bool pbm = p.hard || p.have_prior || p.depairs > 1;
        if (e->lk == LK_MODULE) {      // the user's density inside the same kernel: the code object's own instantiation (dz_set_likelihood_module)
            const dz::Params* pp_ = (const dz::Params*)e->d_params; uint32_t g_ = g; int n_ = n; uint32_t M_ = (uint32_t)visible_rows(e); int64_t s_ = slot0, z_ = append_last ? e->M : (int64_t)-1; int seg_ = seg0;
            void* args[] = {&pp_, &g_, &n_, &M_, &s_, &z_, &seg_, &pub};
            if (e->prof) {
                hipEvent_t ka = prof_event(e), kb = prof_event(e);
                e->ev[PR_GENERATIONS].emplace_back(ka, kb);
                HIPCK(hipExtModuleLaunchKernel(e->lk_gen_fn[(p.ld > 128 ? 2 : 0) + (pbm ? 1 : 0)], gridm.x * blockm.x, 1, 1, blockm.x, 1, 1, ldsm, e->stream, args, nullptr, ka, kb, 0));
            } else HIPCK(hipModuleLaunchKernel(e->lk_gen_fn[(p.ld > 128 ? 2 : 0) + (pbm ? 1 : 0)], gridm.x, 1, 1, blockm.x, 1, 1, (unsigned)ldsm, e->stream, args, nullptr));
            DZCK(launch_check("dz_user_generations"));
            launched();
            e->last_variant = std::string(p.ld > 128 ? (pbm ? "k_generations_user<full,wide>" : "k_generations_user<wide>") : (pbm ? "k_generations_user<full>" : "k_generations_user")) + (rmulti ? " +ring" : "");
            DZCK(after_launch());
            if (slot0 >= 0) e->ntrace += n;
            return 0;
        }
        if (p.ld > 128) {      // 128 < d <= 256: a lane owns four dimensions
            if (pbm) DZ_KLAUNCH(e, PR_GENERATIONS, e->stream, (dz::k_generations_mix<true, false, 2>), gridm, blockm, ldsm, (const dz::Params*)e->d_params, g, n, (uint32_t)visible_rows(e), slot0, append_last ? e->M : (int64_t)-1, seg0, pub);
            else DZ_KLAUNCH(e, PR_GENERATIONS, e->stream, (dz::k_generations_mix<false, false, 2>), gridm, blockm, ldsm, (const dz::Params*)e->d_params, g, n, (uint32_t)visible_rows(e), slot0, append_last ? e->M : (int64_t)-1, seg0, pub);
        } else
        if (multi) {
            if (pbm) DZ_KLAUNCH(e, PR_GENERATIONS, e->stream, (dz::k_generations_mix<true, true>), gridm, blockm, ldsm, (const dz::Params*)e->d_params, g, n, (uint32_t)visible_rows(e), slot0, append_last ? e->M : (int64_t)-1, seg0, pub);
            else DZ_KLAUNCH(e, PR_GENERATIONS, e->stream, (dz::k_generations_mix<false, true>), gridm, blockm, ldsm, (const dz::Params*)e->d_params, g, n, (uint32_t)visible_rows(e), slot0, append_last ? e->M : (int64_t)-1, seg0, pub);
        } else
        if (pbm) DZ_KLAUNCH(e, PR_GENERATIONS, e->stream, dz::k_generations_mix<true>, gridm, blockm, ldsm, (const dz::Params*)e->d_params, g, n, (uint32_t)visible_rows(e), slot0, append_last ? e->M : (int64_t)-1, seg0, pub);
        else DZ_KLAUNCH(e, PR_GENERATIONS, e->stream, dz::k_generations_mix<false>, gridm, blockm, ldsm, (const dz::Params*)e->d_params, g, n, (uint32_t)visible_rows(e), slot0, append_last ? e->M : (int64_t)-1, seg0, pub);
        DZCK(launch_check("k_generations_mix"));
        launched();
        e->last_variant = multi ? (pbm ? "k_generations_mix<full,multi>" : "k_generations_mix<multi>") : (pbm ? "k_generations_mix<full>" : "k_generations_mix");
        if (p.ld > 128) e->last_variant = pbm ? "k_generations_mix<full,wide>" : "k_generations_mix<wide>";
        if (rmulti) e->last_variant += " +ring";
        DZCK(after_launch());
        if (slot0 >= 0) e->ntrace += n;
        return 0;
    }
    const int nrt = p.ld / 16;
    if (const int chd = mega_d2_chains(e)) {      // 128 < d <= 256
        DZCK(upload_params(e));
        dz::MegaLaunch ml;
        const int wpcd = chd == 8 ? 2 : (chd == 4 ? 4 : 1);
        const bool sp = (chd == 16 && mega_d2_two_pass(e)) || (chd == 8 && mega_d2_two_pass(e, 8));
        ml.tri = p.tri != 0; ml.xlds = false; ml.pb = p.hard || p.have_prior || p.depairs > 1; ml.k1 = p.k == 1; ml.ch = chd; ml.wpc = wpcd; ml.redo = false; ml.sp = sp;
        ml.grid = dim3((p.nl + chd - 1) / chd); ml.block = dim3(64 * chd * wpcd);
        ml.lds = sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, nrt, p.ncr, p.ngamma, p.tri != 0, false, chd, false, false, true, rmulti ? e->ad_R1 : 0, sp).total;
        ml.st = e->stream; ml.ka = nullptr; ml.kb = nullptr;
        ml.pp = (const dz::Params*)e->d_params; ml.g = g; ml.n = n; ml.M = (uint32_t)visible_rows(e); ml.slot0 = slot0; ml.zappend = append_last ? e->M : (int64_t)-1; ml.seg0 = seg0; ml.publish = &pub;
        if (e->prof) { ml.ka = prof_event(e); ml.kb = prof_event(e); e->ev[PR_GENERATIONS].emplace_back(ml.ka, ml.kb); }
        const char* name = nullptr;
        switch (nrt) {
            case 1: name = dz::mega_launch_d2_nrt1(ml); break; case 2: name = dz::mega_launch_d2_nrt2(ml); break;
            case 3: name = dz::mega_launch_d2_nrt3(ml); break; case 4: name = dz::mega_launch_d2_nrt4(ml); break;
            case 5: name = dz::mega_launch_d2_nrt5(ml); break; case 6: name = dz::mega_launch_d2_nrt6(ml); break;
            case 7: name = dz::mega_launch_d2_nrt7(ml); break; case 8: name = dz::mega_launch_d2_nrt8(ml); break;
            case 9: name = dz::mega_launch_nrt9(ml); break; case 10: name = dz::mega_launch_nrt10(ml); break;
            case 11: name = dz::mega_launch_nrt11(ml); break; case 12: name = dz::mega_launch_nrt12(ml); break;
            case 13: name = dz::mega_launch_nrt13(ml); break; case 14: name = dz::mega_launch_nrt14(ml); break;
            case 15: name = dz::mega_launch_nrt15(ml); break; case 16: name = dz::mega_launch_nrt16(ml); break;
            default: return fail("k_generations_d2: ld out of range");
        }
        char buf[96]; snprintf(buf, sizeof buf, name, chd, wpcd);
        e->last_variant = buf;
        if (rmulti) e->last_variant += " +ring";
        DZCK(launch_check("k_generations_d2"));
        launched();
        DZCK(after_launch());
        if (slot0 >= 0) e->ntrace += n;
        return 0;
    }
    const int ch = mega_chains(e);
    // waves per chain: 4 when a block holds only 4 chains (fewer than 8 chains per CU) -- the tries of a phase then run side by
    // side (1024 chains: 238 -> 249 M proposals/s, 512: 120 -> 131); at 8 chains per block two waves per chain lose (405 -> 368).
    // multitry off: one try, one wave per chain whatever the block size
    const bool k1 = p.k == 1;
    const bool xlds = mega_xlds(e);
    const bool pb = p.hard || p.have_prior || p.depairs > 1 || mega_redo(e);      // the instantiations with the full proposal code
    DZCK(upload_params(e));
    // (a chain count just above whole rounds of blocks: the remainder in a second launch of smaller blocks -- mega_plan)
    const MegaPlan plan = mega_plan(e);
    const int split_c = plan.split_c, ch_b = plan.split_c != p.nl ? plan.ch_b : 0;
    std::string variant;
    auto launch_part = [&](int c0, int c1, int chp) -> int {
        const int wpcp = (chp == 4 && !k1) ? 4 : 1;
        size_t ldsp = sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, nrt, p.ncr, p.ngamma, p.tri != 0, pb ? true : xlds, chp, p.pb_lds != 0, false, false, rmulti ? e->ad_R1 : 0).total;
        dz::Publish pp = pub; pp.c0 = c0; pp.c1 = c1;
        if (multi) ldsp = mvn_multi_lds(e);      // (burnin_multi: one launch of 16-chain blocks, the states in LDS; + the states before the generation, + the table of probabilities)
        else if (rmulti) { }
        else
        if (split_c != p.nl) { pp.PR = nullptr; pp.PC = nullptr; pp.shift = nullptr; }      // (a split generation's unit sums come from k_adapt_partials)
        else if (publish && !ring && e->adapt_fused && (e->world == 1 || e->adapt_groups) && chp == 16 && wpcp == 1 && !k1 && p.k >= 3 && (pb || xlds)) {      // (the new and old states are read from LDS)
            const size_t with_xo = sizeof(double) * (size_t)dz::mega_layout(p.d, p.k, nrt, p.ncr, p.ngamma, p.tri != 0, true, chp, p.pb_lds != 0, true).total;
            if (with_xo <= (size_t)160 * 1024) { fuse_adapt(); pp.shift = pub.shift; pp.PR = pub.PR; pp.PC = pub.PC; ldsp = with_xo; }
        }
        // the instantiations live in one translation unit per row-tile count (dz_mega_tu.hip)
        dz::MegaLaunch ml;
        ml.tri = p.tri != 0; ml.xlds = pb ? true : xlds; ml.pb = pb; ml.k1 = k1; ml.ch = chp; ml.wpc = wpcp; ml.redo = mega_redo(e);
        ml.ahead = e->mega_w4 && chp == 4 && wpcp == 4 && !pb && xlds && !k1 && p.k >= 3 && p.k <= 6;
        ml.multi = multi;
        ml.grid = dim3((c1 - c0 + chp - 1) / chp); ml.block = dim3(64 * chp * wpcp); ml.lds = ldsp; ml.st = e->stream; ml.ka = nullptr; ml.kb = nullptr;
        ml.pp = (const dz::Params*)e->d_params; ml.g = g; ml.n = n; ml.M = (uint32_t)visible_rows(e); ml.slot0 = slot0; ml.zappend = append_last ? e->M : (int64_t)-1; ml.seg0 = seg0; ml.publish = &pp;
        if (e->prof) { ml.ka = prof_event(e); ml.kb = prof_event(e); e->ev[PR_GENERATIONS].emplace_back(ml.ka, ml.kb); }
        const char* name = nullptr;
        switch (nrt) {
            case 1: name = dz::mega_launch_nrt1(ml); break; case 2: name = dz::mega_launch_nrt2(ml); break;
            case 3: name = dz::mega_launch_nrt3(ml); break; case 4: name = dz::mega_launch_nrt4(ml); break;
            case 5: name = dz::mega_launch_nrt5(ml); break; case 6: name = dz::mega_launch_nrt6(ml); break;
            case 7: name = dz::mega_launch_nrt7(ml); break; case 8: name = dz::mega_launch_nrt8(ml); break;
            default: return fail("persistent kernel: ld > 128");
        }
        char buf[96]; snprintf(buf, sizeof buf, name, chp, wpcp);
        variant += (variant.empty() ? "" : " + ") + std::string(buf);
        return 0;
    };
    DZCK(launch_part(0, split_c, ch));
    if (ch_b) DZCK(launch_part(split_c, p.nl, ch_b));
    e->last_variant = variant + (rmulti ? " +ring" : "");
    DZCK(launch_check("k_generations"));
    launched();
    DZCK(after_launch());
    if (slot0 >= 0) e->ntrace += n;
    return 0;
}

}  // namespace

extern "C" {

int dz_version(void) { return DZ_VERSION; }
const char* dz_last_error(void) { return g_err.c_str(); }
int dz_device_count(int32_t* count)
{
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess) { *count = 0; return fail(std::string("hipGetDeviceCount: ") + hipGetErrorString(err)); }
    *count = n; return 0;
}

int dz_create(const dz_config* cfg, dz_engine** out)
{
    if (!cfg || !out) return fail("null argument");
    if (cfg->ndim < 1 || cfg->nchains < 1 || cfg->nchains_local < 1 || cfg->multitry < 1) return fail("bad sizes");
    if (cfg->ndim > 1024) return fail("ndim > 1024 not supported by this build");
    if (cfg->multitry == 2) return fail("multitry=2 is broken in the reference (Dream.py:867-868); rejected");
    if (cfg->multitry > dz::MAXK) return fail("multitry too large");
    if (cfg->depairs < 1 || cfg->depairs > dz::MAXPAIR) return fail("DEpairs must be 1..8");
    if (cfg->ncr < 1 || cfg->ngamma < 1 || cfg->ncr > 32 || cfg->ngamma > 32) return fail("bad nCR/gamma_levels");
    if (cfg->schedule != 2) return fail("the device engine runs schedule 2 (lockstep generations) only");
    if (cfg->chain_offset < 0 || cfg->chain_offset + cfg->nchains_local > cfg->nchains) return fail("bad shard");
    if (cfg->nchains % cfg->nchains_local) return fail("nchains must be a multiple of nchains_local");
    if (cfg->history_thin < 1) return fail("history_thin must be >= 1");
    if (cfg->history_lag < 0 || cfg->history_lag > 64) return fail("history_lag must be 0..64");
    if (cfg->adapt_lag < 0 || cfg->adapt_lag > 1023 || cfg->reserved0 != 0) return fail("adapt_lag must be 0..1023 (and dz_config.reserved0 zero)");
    int ndev = 0;
    hipError_t derr = hipGetDeviceCount(&ndev);
    if (derr != hipSuccess || ndev < 1) return fail("no HIP device available: libdreamzs has no CPU fallback");
    HIPCK(hipSetDevice(cfg->device));
    dz_engine* e = new dz_engine();
    e->c = *cfg;
    e->gen_c.assign((size_t)cfg->nchains_local, 0);
    if (const char* kv = getenv("DZ_FUSE")) e->fuse = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_STREAM")) e->stream_propose = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA")) e->mega = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_MAXGEN")) e->mega_max_gen = std::max(1, atoi(kv));
    if (const char* kv = getenv("DZ_MEGA_SEGS")) e->mega_segs = std::max(1, atoi(kv));
    if (const char* kv = getenv("DZ_MEGA_BURNIN")) e->mega_burnin = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_REDO")) e->mega_redo_on = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_ADAPT_FUSED")) e->adapt_fused = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_MIX_PB")) e->mega_mix_pb = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_SPLIT")) e->mega_split = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_ADAPT_RING")) e->ad_ring = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_MIX_WIDE")) e->mega_mix_wide = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_W4")) e->mega_w4 = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_MEGA_D2")) e->mega_d2 = atoi(kv);
    static_assert(dz::DZ_MAX_REDRAWS_DEV == DZ_MAX_REDRAWS && dz::DZ_REDRAW_KEY_STEP_DEV == DZ_REDRAW_KEY_STEP, "redraw constants");
    if (const char* kv = getenv("DZ_QFIN")) e->q_defer = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_FUSE_STREAM")) e->fuse_stream = atoi(kv) != 0;
    if (const char* kv = getenv("DZ_LOGP_BM")) e->logp_bm = atoi(kv);          // 64 / 128: points per block of k_logp_mvn_gemm (default: by size)
    if (const char* kv = getenv("DZ_MEGA_CHAINS")) { const int v = atoi(kv); e->mega_ch = (v == 16 || v == 12 || v == 8 || v == 4) ? v : 0; }
    if (const char* kv = getenv("DZ_WPB")) e->waves_per_block = atoi(kv);
    if (const char* kv = getenv("DZ_PROPOSE_SPLIT")) e->propose_split = atoi(kv);
    if (const char* kv = getenv("DZ_MFMA_PT")) e->force_pt = atoi(kv) == 2 ? 2 : atoi(kv) == 1 ? 1 : 0;
    dz::Params& p = e->p;
    p.N = cfg->nchains; p.nl = cfg->nchains_local; p.off = cfg->chain_offset; p.d = cfg->ndim;
    p.ld = (cfg->ndim + 15) / 16 * 16;
    p.k = cfg->multitry; p.depairs = cfg->depairs; p.ncr = cfg->ncr; p.ngamma = cfg->ngamma; p.thin = cfg->history_thin;
    p.burnin = cfg->crossover_burnin; p.adapt_cr = cfg->adapt_crossover; p.adapt_g = cfg->adapt_gamma; p.hard = 0;   // set by dz_set_bounds
    p.k0 = (uint32_t)cfg->seed; p.k1 = (uint32_t)(cfg->seed >> 32);
    p.lamb = cfg->lamb; p.zeta = cfg->zeta; p.snooker = cfg->snooker; p.pgu = cfg->p_gamma_unity; p.T = cfg->temperature;
    {   // contract constants made once (dz_device.h uniform16; dz_kernels.h crossover_threshold)
        const double low = -cfg->lamb, high = cfg->lamb;
        p.ec1 = (high - low) * (1.0 / 65536.0); p.ec0 = low + (high - low) * (1.0 / 131072.0);
        for (int m = 0; m < 32; ++m) p.crthr[m] = m < cfg->ncr ? dz::crossover_threshold((double)(m + 1) / (double)cfg->ncr) : 0u;
        const double q = std::min(1.0, std::max(0.0, cfg->p_gamma_unity));
        p.pgu_thr = (unsigned long long)std::ceil(std::ldexp(q, 53));
        const double sn = std::min(1.0, std::max(0.0, cfg->snooker));
        p.snk_thr = (unsigned long long)std::ceil(std::ldexp(sn, 53));
    }
    const int chunks = (p.ld + 127) / 128;
    e->nch = chunks <= 1 ? 1 : chunks <= 2 ? 2 : chunks <= 4 ? 4 : 8;
    e->adapt = cfg->adapt_crossover || cfg->adapt_gamma;
    e->world = cfg->nchains / cfg->nchains_local; e->rank = cfg->chain_offset / cfg->nchains_local;
    if (const char* cm = (getenv("DZ_CUMASK") && *getenv("DZ_CUMASK")) ? getenv("DZ_CUMASK") : nullptr) {      // measurement switch: every kernel of the engine on the first DZ_CUMASK compute units of each XCD-interleaved numbering
        const int ncu = std::max(1, atoi(cm));
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < ncu && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
        HIPCK(hipExtStreamCreateWithCUMask(&e->stream, 8, mask));
    } else
    HIPCK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    {
        int nl_req = 1;     // chain groups on separate streams; 2 gains ~3% at 4096 chains but shows occasional 2x-slow passes
        if (const char* ev = getenv("DZ_STREAMS")) nl_req = atoi(ev);
        e->nlanes = std::max(1, std::min(8, nl_req));
        if (cfg->nchains_local < 64 * e->nlanes) e->nlanes = 1;
        e->lane_stream[0] = e->stream;
        for (int s = 1; s < e->nlanes; ++s) HIPCK(hipStreamCreateWithFlags(&e->lane_stream[s], hipStreamNonBlocking));
        for (int s = 0; s < e->nlanes; ++s) HIPCK(hipEventCreateWithFlags(&e->lane_ev[s], hipEventDisableTiming));
        if (const char* ra = getenv("DZ_RUNAHEAD")) e->ra_stride = std::max(0, atoi(ra));
        if (const char* fb = getenv("DZ_LOGP_BIG")) e->force_big = atoi(fb) != 0;
        if (const char* fg = getenv("DZ_LOGP_GEMM")) e->logp_gemm = atoi(fg) != 0;
        if (const char* lw = getenv("DZ_LOGP_WAVES")) e->logp_waves = std::max(4, std::min(8, atoi(lw)));
        for (int s = 0; s < 4; ++s) HIPCK(hipEventCreateWithFlags(&e->ra_ev[s], hipEventDisableTiming));
    }
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cu = prop.multiProcessorCount; }
    const size_t ld = p.ld, nl = p.nl, N = p.N, k = p.k, tc = (size_t)cfg->trace_capacity;
    int rc = 0;
    rc |= ealloc(e, &p.Z, (size_t)cfg->history_capacity * ld);
    rc |= ealloc(e, &p.X, nl * ld); rc |= ealloc(e, &p.lprior, nl); rc |= ealloc(e, &p.llike, nl);
    rc |= ealloc(e, &p.P, nl * k * ld); rc |= ealloc(e, &p.R, nl * k * ld);
    rc |= ealloc(e, &p.p_prior, nl * k); rc |= ealloc(e, &p.p_like, nl * k); rc |= ealloc(e, &p.p_slogp, nl * k);
    rc |= ealloc(e, &p.r_prior, nl * k); rc |= ealloc(e, &p.r_like, nl * k); rc |= ealloc(e, &p.r_slogp, nl * k);
    rc |= ealloc(e, &p.cur_snk, nl);
    p.npt = 1 + (2 * cfg->depairs + 3) / 4; p.nslots = 3 + (2 * cfg->multitry - 1) * p.npt;
    rc |= ealloc(e, &e->d_draws[0], nl * (size_t)p.nslots); rc |= ealloc(e, &e->d_draws[1], nl * (size_t)p.nslots);
    rc |= ealloc(e, &e->d_ctl[0], nl); rc |= ealloc(e, &e->d_ctl[1], nl);
    rc |= ealloc(e, &p.sel, nl);
#ifdef DZ_EXPERIMENTS
    { void* hp = nullptr; if (hipHostMalloc(&hp, (size_t)4 * nl * 16 * 8, hipHostMallocDefault) == hipSuccess) { memset(hp, 0, (size_t)4 * nl * 16 * 8); p.dbg = (unsigned long long*)hp; } }
#endif
    rc |= ealloc(e, &e->d_params, 1);
    rc |= ealloc(e, &e->d_redraw_count, 1); p.redraw_count = e->d_redraw_count;
    rc |= ealloc(e, &e->d_mins, ld); rc |= ealloc(e, &e->d_maxs, ld);
    rc |= ealloc(e, &e->d_gtab, (size_t)cfg->ngamma * cfg->depairs * p.d);
    rc |= ealloc(e, &e->d_shared, (size_t)2 * 3 * (cfg->ncr + cfg->ngamma));
    rc |= ealloc(e, &e->d_pkind, ld); rc |= ealloc(e, &e->d_pa, ld); rc |= ealloc(e, &e->d_pb, ld); rc |= ealloc(e, &e->d_plogb, ld); rc |= ealloc(e, &e->d_pc2, ld);
    if (e->adapt) {
        for (int i = 0; i < 3; ++i) rc |= ealloc(e, &e->d_cp[i], N * ld);
        p.cp_prev = e->d_cp[0]; p.cp_new = e->d_cp[1]; e->cp_idx = 1;
        rc |= ealloc(e, &e->d_partial, (size_t)2 * ((N + 63) / 64) * ld);      // strip sums of pass 0 | of pass 1
        rc |= ealloc(e, &e->d_mean, ld); rc |= ealloc(e, &e->d_sd, ld); rc |= ealloc(e, &e->d_sdc, ld);
        rc |= ealloc(e, &e->d_dl, N); rc |= ealloc(e, &e->d_dlg, N); rc |= ealloc(e, &e->d_binc, N); rc |= ealloc(e, &e->d_bing, N);
        rc |= ealloc(e, &e->d_binsum, (size_t)2 * ((N + 63) / 64) * (cfg->ncr + cfg->ngamma));
        const size_t nq = 2 + (size_t)cfg->ncr + cfg->ngamma, units = (N + 15) / 16;
        // adapt_lag = L >= 1: rings of L + 1 generations (engine fields ad_*); several burn-in generations per launch on one GPU (ad_multi)
        e->ad_R1 = cfg->adapt_lag + 1; e->ad_nbp = (cfg->ncr + cfg->ngamma + 1) & ~1;
        // sharded, and this rank owns whole groups of 256 chains (then every rank does: equal shards): the burn-in exchanges group sums
        const bool groups_on = !(getenv("DZ_ADAPT_GROUPS") && atoi(getenv("DZ_ADAPT_GROUPS")) == 0);
        const bool groups = e->world > 1 && groups_on && p.off % 256 == 0 && p.nl % 256 == 0;
        // (several GPUs: only ranks that exchange group sums -- the records of a launch's generations travel in one exchange, adapt_finish_groups)
        e->ad_multi = cfg->adapt_lag > 0 && (e->world == 1 || groups) && !(getenv("DZ_ADAPT_MULTI") && atoi(getenv("DZ_ADAPT_MULTI")) == 0);
        e->ad_tot_stride = cfg->adapt_lag > 0 ? nq * ld : 0;
        e->ad_pr_stride = units * nq * ld; e->ad_pc_stride = units * (size_t)(cfg->ncr + cfg->ngamma);
        const size_t prs = e->ad_multi ? (size_t)e->ad_R1 : 1;
        rc |= ealloc(e, &e->d_PR, prs * e->ad_pr_stride); rc |= ealloc(e, &e->d_PC, prs * e->ad_pc_stride);
        rc |= ealloc(e, &e->d_TOT, (size_t)e->ad_R1 * nq * ld); rc |= ealloc(e, &e->d_CNT, (size_t)e->ad_R1 * e->ad_nbp);
        if (cfg->adapt_lag > 0) {
            rc |= ealloc(e, &e->d_DOT, (size_t)e->ad_R1 * e->ad_nbp);
            rc |= ealloc(e, &e->d_x0ring, (size_t)2 * e->ad_R1 * ld); rc |= ealloc(e, &e->d_x0start, ld);
        }
        if (groups) {
            e->adapt_groups = true;
            e->gs_nbp = (cfg->ncr + cfg->ngamma + 15) / 16 * 16;
            e->gs_rec = (size_t)(p.nl / 256) * (nq * ld + (size_t)e->gs_nbp) + ld;
            // (adapt_lag = L >= 1: room for the records of a launch of L + 1 generations from every rank)
            for (int i = 0; i < 2; ++i) { rc |= ealloc(e, &e->d_GS[i], (size_t)e->world * e->gs_rec * (e->ad_multi ? (size_t)e->ad_R1 : 1)); rc |= ealloc(e, &e->d_shift[i], ld); }
        }
    }
    if (tc) {
        p.tcap = (long long)tc;
        rc |= ealloc(e, &p.tX, tc * nl * ld); rc |= ealloc(e, &p.tlogp, tc * nl);
        rc |= ealloc(e, &p.tmoved, tc * nl); rc |= ealloc(e, &p.tsnk, tc * nl); rc |= ealloc(e, &p.ttry, tc * nl); rc |= ealloc(e, &p.tcr, tc * nl);
    }
    rc |= ealloc(e, &e->d_cmean, nl * (size_t)p.d); rc |= ealloc(e, &e->d_cvar, nl * (size_t)p.d); rc |= ealloc(e, &e->d_rhat, (size_t)p.d);
    if (rc) { dz_destroy(e); return -1; }
    p.mins = e->d_mins; p.maxs = e->d_maxs; p.gtab = e->d_gtab;
    point_shared(e);
    p.pkind = e->d_pkind; p.pa = e->d_pa; p.pb = e->d_pb; p.plogb = e->d_plogb; p.pc2 = e->d_pc2; p.have_prior = 0; p.prior_nonormal = 1;
    // defaults: unbounded, uniform CR / gamma-level probabilities (Dream.py:134, :143), computed gamma table
    {
        std::vector<double> lo(ld, -HUGE_VAL), hi(ld, HUGE_VAL);
        HIPCK(hipMemcpy(e->d_mins, lo.data(), sizeof(double) * ld, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(e->d_maxs, hi.data(), sizeof(double) * ld, hipMemcpyHostToDevice));
        std::vector<double> sh((size_t)3 * (cfg->ncr + cfg->ngamma), 0.0);
        for (int m = 0; m < cfg->ncr; ++m) sh[m] = 1.0 / (double)cfg->ncr;
        for (int m = 0; m < cfg->ngamma; ++m) sh[(size_t)3 * cfg->ncr + m] = 1.0 / (double)cfg->ngamma;
        HIPCK(hipMemcpy(e->d_shared, sh.data(), sizeof(double) * sh.size(), hipMemcpyHostToDevice));
    }
    *out = e;
    return dz_set_gamma_table(e, nullptr);
}

int dz_destroy(dz_engine* e)
{
#ifdef DZ_EXPERIMENTS
    if (e && e->p.dbg) {
        hipDeviceSynchronize();
        if (FILE* f = fopen("gpurun_out/stamps.bin", "wb")) { fwrite(e->p.dbg, 8, (size_t)4 * e->p.nl * 16, f); fclose(f); }
    }
#endif
    if (!e) return 0;
    (void)hipSetDevice(e->c.device);
    for (int s2 = 0; s2 < e->nlanes; ++s2) if (e->lane_stream[s2]) (void)hipStreamSynchronize(e->lane_stream[s2]);
    for (int s2 = 1; s2 < e->nlanes; ++s2) if (e->lane_stream[s2]) (void)hipStreamDestroy(e->lane_stream[s2]);
    for (int s2 = 0; s2 < e->nlanes; ++s2) if (e->lane_ev[s2]) (void)hipEventDestroy(e->lane_ev[s2]);
    for (int s2 = 0; s2 < 4; ++s2) if (e->ra_ev[s2]) (void)hipEventDestroy(e->ra_ev[s2]);
    for (auto& v : e->ev) for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t x : e->ev_pool) (void)hipEventDestroy(x);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    if (e->lk_module) (void)hipModuleUnload(e->lk_module);
    if (e->d_lk_data) (void)hipFree(e->d_lk_data);
    for (auto& pr : e->peers) {
        if (pr.st) { (void)hipStreamSynchronize(pr.st); (void)hipStreamDestroy(pr.st); }
        if (pr.Z) (void)hipIpcCloseMemHandle(pr.Z);
        if (pr.flags) (void)hipIpcCloseMemHandle(pr.flags);
        for (double* q : pr.cp) if (q) (void)hipIpcCloseMemHandle(q);
        for (double* q : pr.gs) if (q) (void)hipIpcCloseMemHandle(q);
    }
    for (hipEvent_t x : e->push_ev) if (x) (void)hipEventDestroy(x);
    if (e->h_gate) (void)hipHostFree(e->h_gate);
    for (void* q : e->to_free) (void)hipFree(q);
    if (e->d_scratch) (void)hipFree(e->d_scratch);
    if (e->d_qpart) (void)hipFree(e->d_qpart);
    if (e->h_pin) (void)hipHostFree(e->h_pin);
    for (hipEvent_t x : e->pin_ev) if (x) (void)hipEventDestroy(x);
    if (e->h_redo) (void)hipHostFree(e->h_redo);
    if (e->h_redo_list) (void)hipHostFree(e->h_redo_list);
    if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
    for (hipEvent_t ev : e->copy_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return 0;
}

int dz_set_bounds(dz_engine* e, const double* mins, const double* maxs)
{
    HIPCK(hipSetDevice(e->c.device));
    bool any_finite = false;
    for (int j = 0; j < e->p.d; ++j) any_finite = any_finite || std::isfinite(mins[j]) || std::isfinite(maxs[j]);
    e->p.hard = (e->c.hardboundaries && any_finite) ? 1 : 0;     // with all bounds infinite the reflection code can never trigger
    e->h_mins.assign(mins, mins + e->p.d); e->h_maxs.assign(maxs, maxs + e->p.d);
    DZCK(upload_padded(e, e->d_mins, mins, 1, -HUGE_VAL));
    return upload_padded(e, e->d_maxs, maxs, 1, HUGE_VAL);
}

int dz_set_gamma_table(dz_engine* e, const double* table)
{   // Dream.py:172-179
    HIPCK(hipSetDevice(e->c.device));
    const int d = e->p.d, ng = e->c.ngamma, np = e->c.depairs;
    std::vector<double> t((size_t)ng * np * d);
    if (table) memcpy(t.data(), table, sizeof(double) * t.size());
    else {
        double dec = 1.0;
        for (int lev = 1; lev <= ng; ++lev) {
            for (int delta = 1; delta <= np; ++delta)
                for (int dp = 1; dp <= d; ++dp)
                    t[((size_t)(lev - 1) * np + (delta - 1)) * d + (dp - 1)] = (2.38 / std::sqrt((double)(2 * delta) * (double)dp)) / dec;
            dec = dec * 2.0;
        }
    }
    HIPCK(hipMemcpy(e->d_gtab, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
    return 0;
}

int dz_set_history(dz_engine* e, const double* Z, int64_t rows)
{
    HIPCK(hipSetDevice(e->c.device));
    if (rows > e->c.history_capacity) return fail("history exceeds capacity");
    if (rows > 0xffffffffll) return fail("history too long");
    // (the peers' flag words count this rank's appends from the attach on: an archive reset underneath them would leave the gates
    //  comparing exchange numbers of two different histories)
    if (e->peer_on) return fail("dz_set_history: the archive of an engine attached to the peer transport cannot be reset (set it before dz_peer_attach)");
    DZCK(upload_padded(e, e->p.Z, Z, (int)rows, 0.0));
    e->M = rows; e->napp = 0;
    return 0;
}

int dz_set_state(dz_engine* e, const double* X, const double* prior, const double* like)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(upload_padded(e, e->p.X, X, e->p.nl, 0.0));
    if (prior && like) {
        HIPCK(hipMemcpy(e->p.lprior, prior, sizeof(double) * e->p.nl, hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(e->p.llike, like, sizeof(double) * e->p.nl, hipMemcpyHostToDevice));
        e->have_logp = true;
    } else e->have_logp = false;
    return 0;
}

int dz_set_cr_probs(dz_engine* e, const double* pr, int32_t n)
{
    if (n != e->c.ncr) return fail("nCR mismatch");
    HIPCK(hipSetDevice(e->c.device));
    DZCK(adapt_flush(e));
    DZCK(sync_all(e));
    HIPCK(hipMemcpy(e->p.cr_probs, pr, sizeof(double) * n, hipMemcpyHostToDevice));
    std::fill(e->own_init.begin(), e->own_init.end(), 0);
    e->draws_gen = -1;
    return 0;
}
int dz_set_gamma_probs(dz_engine* e, const double* pr, int32_t n)
{
    if (n != e->c.ngamma) return fail("ngamma mismatch");
    HIPCK(hipSetDevice(e->c.device));
    DZCK(adapt_flush(e));
    DZCK(sync_all(e));
    HIPCK(hipMemcpy(e->p.g_probs, pr, sizeof(double) * n, hipMemcpyHostToDevice));
    std::fill(e->own_init.begin(), e->own_init.end(), 0);
    e->draws_gen = -1;
    return 0;
}

int dz_set_prior(dz_engine* e, const int32_t* kind, const double* a, const double* b)
{
    HIPCK(hipSetDevice(e->c.device));
    const int d = e->p.d;
    bool any = false;
    for (int j = 0; j < d; ++j) { if (kind[j] < 0 || kind[j] > 2) return fail("prior kind must be 0,1,2"); any = any || kind[j] != 0; }
    HIPCK(hipMemcpy(e->d_pkind, kind, sizeof(int32_t) * d, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(e->d_pa, a, sizeof(double) * d, hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(e->d_pb, b, sizeof(double) * d, hipMemcpyHostToDevice));
    {   // Params::pc2: normal: the reciprocal of the scale; uniform: the upper end of the support
        std::vector<double> c2((size_t)d, 0.0);
        bool normal = false;
        for (int j = 0; j < d; ++j) { c2[j] = kind[j] == 1 ? 1.0 / b[j] : (kind[j] == 2 ? a[j] + b[j] : 0.0); normal = normal || kind[j] == 1; }
        HIPCK(hipMemcpy(e->d_pc2, c2.data(), sizeof(double) * d, hipMemcpyHostToDevice));
        e->p.prior_nonormal = normal ? 0 : 1;
        if (const char* kv = getenv("DZ_PRIOR_NONORMAL")) if (!atoi(kv)) e->p.prior_nonormal = 0;      // (measurement: the general butterfly form)
    }
    hipLaunchKernelGGL(dz::k_prior_consts, dim3((d + 127) / 128), dim3(128), 0, e->stream, (const double*)e->d_pb, d, e->d_plogb);
    DZCK(launch_check("k_prior_consts"));
    e->p.have_prior = any ? 1 : 0;
    e->h_pkind.assign(kind, kind + d); e->h_pa.assign(a, a + d); e->h_pb.assign(b, b + d);
    return 0;
}

int dz_set_likelihood_mvn(dz_engine* e, const double* mu, const double* M, int32_t kind, double log_F)
{
    HIPCK(hipSetDevice(e->c.device));
    const int d = e->p.d, ld = e->p.ld;
    if (!e->d_mu) DZCK(ealloc(e, &e->d_mu, (size_t)32 * ld));
    if (!e->d_Mt) DZCK(ealloc(e, &e->d_Mt, (size_t)ld * ld));
    std::vector<double> mt((size_t)ld * ld, 0.0), m(ld, 0.0);
    for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) mt[(size_t)c * ld + r] = M[(size_t)r * d + c];
    memcpy(m.data(), mu, sizeof(double) * d);
    HIPCK(hipMemcpy(e->d_Mt, mt.data(), sizeof(double) * mt.size(), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(e->d_mu, m.data(), sizeof(double) * ld, hipMemcpyHostToDevice));
    e->p.mu = e->d_mu; e->p.Mt = e->d_Mt; e->p.logF = log_F; e->p.tri = kind != 0; e->lk = LK_MVN;
    e->p.Mtp = nullptr; e->p.mtp_len = 0;
    e->p.mu_zero = 1;
    for (int j = 0; j < d; ++j) if (!(mu[j] == 0.0) || std::signbit(mu[j])) e->p.mu_zero = 0;
    if (kind != 0 && ld <= 256) {   // packed triangle for k_logp_mvn_lds and the persistent kernels (dz_kernels.h tri_row_offset); 128 < ld <= 256: k_generations_d2 reads it from L2
        const int rows = 4 * ((d + 3) / 4);
        std::vector<double> pk((size_t)dz::tri_row_offset(rows), 0.0);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < 16 * (r / 16 + 1); ++c) pk[(size_t)dz::tri_row_offset(r) + c] = mt[(size_t)r * ld + c];
        if (!e->d_Mtp) DZCK(ealloc(e, &e->d_Mtp, (size_t)dz::tri_row_offset(256)));
        HIPCK(hipMemcpy(e->d_Mtp, pk.data(), sizeof(double) * pk.size(), hipMemcpyHostToDevice));
        e->p.Mtp = e->d_Mtp; e->p.mtp_len = (int)pk.size();
    }
    return 0;
}

int dz_set_likelihood_mixture(dz_engine* e, int32_t J, const double* mu, const double* log_F)
{
    HIPCK(hipSetDevice(e->c.device));
    if (J < 1 || J > 32) return fail("mixture components must be 1..32");
    const int d = e->p.d, ld = e->p.ld;
    if (!e->d_mu) DZCK(ealloc(e, &e->d_mu, (size_t)32 * ld));
    if (!e->d_mixF) DZCK(ealloc(e, &e->d_mixF, 32));
    std::vector<double> m((size_t)J * ld, 0.0);
    for (int j = 0; j < J; ++j) memcpy(&m[(size_t)j * ld], mu + (size_t)j * d, sizeof(double) * d);
    HIPCK(hipMemcpy(e->d_mu, m.data(), sizeof(double) * m.size(), hipMemcpyHostToDevice));
    HIPCK(hipMemcpy(e->d_mixF, log_F, sizeof(double) * J, hipMemcpyHostToDevice));
    e->p.mu = e->d_mu; e->p.mixF = e->d_mixF; e->p.J = J; e->lk = LK_MIX;
    return 0;
}

int dz_set_likelihood_host(dz_engine* e, dz_logp_cb cb, void* user)
{
    if (!cb) return fail("null callback");
    e->cb = cb; e->cb_user = user; e->lk = LK_HOST;
    return 0;
}

int dz_set_likelihood_module(dz_engine* e, const char* code_object_path, const char* kernel_name, int32_t lanes_per_point, int32_t flags,
                             const void* data, int64_t data_bytes)
{
    if (!e || !code_object_path || !kernel_name) return fail("null argument");
    if (lanes_per_point != 1 && lanes_per_point != 64) return fail("dz_set_likelihood_module: lanes_per_point must be 1 (a thread per point) or 64 (a wave per point)");
    if (data_bytes < 0 || (data_bytes > 0 && !data)) return fail("dz_set_likelihood_module: bad data block");
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
    hipError_t err = hipModuleLoad(&mod, code_object_path);
    if (err != hipSuccess) { (void)hipGetLastError(); return fail(std::string("hipModuleLoad(") + code_object_path + "): " + hipGetErrorString(err) + " (a gfx950 code object is expected: hipcc --offload-arch=gfx950 --genco)"); }
    err = hipModuleGetFunction(&fn, mod, kernel_name);
    if (err != hipSuccess) { (void)hipGetLastError(); (void)hipModuleUnload(mod); return fail(std::string("hipModuleGetFunction(") + kernel_name + "): " + hipGetErrorString(err) + " (the kernel must be extern \"C\")"); }
    hipFunction_t gen[4] = {nullptr, nullptr, nullptr, nullptr};      // optional: the persistent kernels around the same density (a code object built by DeviceFunctionLogLike has them)
    if (hipModuleGetFunction(&gen[0], mod, "dz_user_generations_v" DZ_USER_STR(DZ_USER_ABI)) != hipSuccess) { (void)hipGetLastError(); gen[0] = nullptr; }
    if (hipModuleGetFunction(&gen[1], mod, "dz_user_generations_full_v" DZ_USER_STR(DZ_USER_ABI)) != hipSuccess) { (void)hipGetLastError(); gen[1] = nullptr; }
    if (hipModuleGetFunction(&gen[2], mod, "dz_user_generations_wide_v" DZ_USER_STR(DZ_USER_ABI)) != hipSuccess) { (void)hipGetLastError(); gen[2] = nullptr; }
    if (hipModuleGetFunction(&gen[3], mod, "dz_user_generations_wide_full_v" DZ_USER_STR(DZ_USER_ABI)) != hipSuccess) { (void)hipGetLastError(); gen[3] = nullptr; }
    if (getenv("DZ_MEGA_USER") && atoi(getenv("DZ_MEGA_USER")) == 0) for (auto& gfn : gen) gfn = nullptr;
    // the new data block first, into a temporary: the engine's module, function and data change together, and only once every step has
    // succeeded -- a failure leaves the engine as it was (advisor, round 5)
    void* d_new = nullptr;
    if (data_bytes > 0) {
        hipError_t me = hipMalloc(&d_new, (size_t)data_bytes);
        if (me == hipSuccess) me = hipMemcpy(d_new, data, (size_t)data_bytes, hipMemcpyHostToDevice);
        if (me != hipSuccess) {
            (void)hipGetLastError();
            if (d_new) (void)hipFree(d_new);
            (void)hipModuleUnload(mod);
            return fail(std::string("dz_set_likelihood_module: data block: ") + hipGetErrorString(me));
        }
    }
    if (e->lk_module) { (void)hipModuleUnload(e->lk_module); e->lk_module = nullptr; e->lk_fn = nullptr; }
    if (e->d_lk_data) { (void)hipFree(e->d_lk_data); e->d_lk_data = nullptr; }
    e->d_lk_data = d_new;
    for (int i = 0; i < 4; ++i) e->lk_gen_fn[i] = gen[i];
    e->p.udata = d_new; e->p.J = 1;      // (J: the wave's scratch row in the persistent kernel's layout)
    e->lk_module = mod; e->lk_fn = fn; e->lk_lanes = lanes_per_point; e->lk_finite = (flags & DZ_LIKE_ALWAYS_FINITE) != 0;
    e->lk = LK_MODULE; e->have_logp = false;
    return 0;
}

const char* dz_comm_library(void)
{   // path of the librccl the engine uses (loads it if necessary); NULL with dz_last_error set if it cannot be loaded
    if (g_rccl.load()) return nullptr;
    return g_rccl.path.c_str();
}

const char* dz_hip_library(void)
{   // path of the HIP runtime this library's calls are bound to (in a process that imported PyTorch first that is torch's copy)
    static std::string path;
    Dl_info hi;
    if (!dladdr((void*)&hipGetDeviceCount, &hi) || !hi.dli_fname) { fail("cannot locate the HIP runtime"); return nullptr; }
    path = hi.dli_fname;
    return path.c_str();
}

int dz_comm_unique_id(void* id128)
{
    DZCK(g_rccl.load());
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return fail("ncclGetUniqueId failed");
    static_assert(sizeof(ncclUniqueId) == 128, "unexpected ncclUniqueId size");
    memcpy(id128, &id, 128);
    return 0;
}

int dz_comm_init_rccl(dz_engine* e, int32_t rank, int32_t world, const void* id128)
{
    HIPCK(hipSetDevice(e->c.device));
    if (world != e->world || rank != e->rank) return fail("rank/world do not match the chain shard in dz_config");
    DZCK(g_rccl.load());
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclResult_t r = g_rccl.CommInitRank(&e->comm, world, id, rank);
    if (r != ncclSuccess) return fail(std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    return 0;
}

int dz_comm_count(dz_engine* e, int32_t* ranks)
{   // ncclCommCount of the engine's communicator: how many ranks RCCL itself says take part (0: no communicator)
    if (!e || !ranks) return fail("null argument");
    *ranks = 0;
    if (!e->comm) return 0;
    if (!g_rccl.CommCount) return fail("librccl has no ncclCommCount");
    int n = 0;
    ncclResult_t r = g_rccl.CommCount(e->comm, &n);
    if (r != ncclSuccess) return fail(std::string("ncclCommCount: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    *ranks = n;
    return 0;
}

int dz_set_exchange(dz_engine* e, dz_exchange_cb cb, void* user) { e->xcb = cb; e->xcb_user = user; return 0; }

// ---- peer transport: rows pushed by the copy engines into the other ranks' buffers, mapped through HIP IPC ----
namespace {
struct PeerBlob {          // what a rank publishes about itself (DZ_PEER_BLOB_BYTES)
    uint32_t magic, nchains, nchains_local, ld; int64_t capacity; int32_t has_cp, rank;
    hipIpcMemHandle_t z, flags, cp[3], gs[2];      // (has_cp: 1 = position buffers, 3 = ... and the group-sum records of the sharded burn-in)
};
static_assert(sizeof(PeerBlob) <= DZ_PEER_BLOB_BYTES, "PeerBlob does not fit DZ_PEER_BLOB_BYTES");
}  // namespace

int dz_peer_export(dz_engine* e, void* blob)
{
    HIPCK(hipSetDevice(e->c.device));
    if (e->world < 2) return fail("dz_peer_export: the engine is not sharded");
    if (e->world > 64) return fail("dz_peer_export: more than 64 ranks (the gate kernel polls one flag word per lane)");
    if (!e->d_flags) {
        // The flag words are written by OTHER GPUs' copy engines while a kernel of this GPU polls them: ordinary (coarse-grained) device
        // memory may sit in this GPU's L2, which a write arriving over the fabric does not update -- the poll would spin on its own cached
        // copy.  Uncached device memory (what RCCL uses for its peer-to-peer flags) is read from memory by every system-scope load.
        {
            void* q = nullptr;
            if (hipExtMallocWithFlags(&q, sizeof(unsigned long long) * XK_KINDS * (size_t)e->world, hipDeviceMallocUncached) != hipSuccess) {
                (void)hipGetLastError();
                return fail("hipExtMallocWithFlags(hipDeviceMallocUncached) for the peer flags failed");
            }
            HIPCK(hipMemset(q, 0, sizeof(unsigned long long) * XK_KINDS * (size_t)e->world));
            HIPCK(hipStreamSynchronize(nullptr));
            e->d_flags = (unsigned long long*)q; e->to_free.push_back(q);
        }
        // one flag push per history append and per published generation: bounded by the archive's capacity and the burn-in
        e->seq_cap = (e->c.history_capacity / std::max(1, e->p.N)) + (int64_t)e->c.crossover_burnin + 65536;      // (+ rendezvous numbers: dz_comm_barrier)
        if (e->seq_cap > ((int64_t)1 << 24)) e->seq_cap = (int64_t)1 << 24;
        DZCK(ealloc(e, &e->d_seq, (size_t)e->seq_cap));
        std::vector<unsigned long long> h((size_t)e->seq_cap);
        for (size_t i = 0; i < h.size(); ++i) h[i] = i;
        HIPCK(hipMemcpy(e->d_seq, h.data(), sizeof(unsigned long long) * h.size(), hipMemcpyHostToDevice));
        HIPCK(hipHostMalloc((void**)&e->h_gate, 8 * sizeof(unsigned long long), hipHostMallocMapped));
        memset(e->h_gate, 0, 8 * sizeof(unsigned long long));
    }
    PeerBlob b; memset(&b, 0, sizeof b);
    b.magic = 0x445a5058u; b.nchains = (uint32_t)e->p.N; b.nchains_local = (uint32_t)e->p.nl; b.ld = (uint32_t)e->p.ld; b.capacity = e->c.history_capacity;
    b.has_cp = e->adapt ? (e->adapt_groups ? 3 : 1) : 0; b.rank = e->rank;
    HIPCK(hipIpcGetMemHandle(&b.z, e->p.Z));
    HIPCK(hipIpcGetMemHandle(&b.flags, e->d_flags));
    if (e->adapt) for (int i = 0; i < 3; ++i) HIPCK(hipIpcGetMemHandle(&b.cp[i], e->d_cp[i]));
    if (e->adapt_groups) for (int i = 0; i < 2; ++i) HIPCK(hipIpcGetMemHandle(&b.gs[i], e->d_GS[i]));
    memset(blob, 0, DZ_PEER_BLOB_BYTES);
    memcpy(blob, &b, sizeof b);
    return 0;
}

int dz_peer_attach(dz_engine* e, int32_t rank, int32_t world, const void* blobs)
{
    HIPCK(hipSetDevice(e->c.device));
    if (world != e->world || rank != e->rank) return fail("rank/world do not match the chain shard in dz_config");
    if (!e->d_flags) return fail("dz_peer_attach: call dz_peer_export first");
    if (e->peer_on) return fail("dz_peer_attach: already attached");
    e->peers.assign((size_t)world, dz_engine::Peer());
    for (int r = 0; r < world; ++r) {
        if (r == rank) continue;
        PeerBlob b; memcpy(&b, (const char*)blobs + (size_t)r * DZ_PEER_BLOB_BYTES, sizeof b);
        if (b.magic != 0x445a5058u || b.rank != r) return fail("dz_peer_attach: blob " + std::to_string(r) + " is not rank " + std::to_string(r) + "'s export");
        if ((int)b.nchains != e->p.N || (int)b.nchains_local != e->p.nl || (int)b.ld != e->p.ld || b.capacity != e->c.history_capacity || b.has_cp != (e->adapt ? (e->adapt_groups ? 3 : 1) : 0))
            return fail("dz_peer_attach: rank " + std::to_string(r) + " was created with a different configuration");
        dz_engine::Peer& pr = e->peers[r];
        HIPCK(hipIpcOpenMemHandle((void**)&pr.Z, b.z, hipIpcMemLazyEnablePeerAccess));
        HIPCK(hipIpcOpenMemHandle((void**)&pr.flags, b.flags, hipIpcMemLazyEnablePeerAccess));
        if (e->adapt) for (int i = 0; i < 3; ++i) HIPCK(hipIpcOpenMemHandle((void**)&pr.cp[i], b.cp[i], hipIpcMemLazyEnablePeerAccess));
        if (e->adapt_groups) for (int i = 0; i < 2; ++i) HIPCK(hipIpcOpenMemHandle((void**)&pr.gs[i], b.gs[i], hipIpcMemLazyEnablePeerAccess));
        HIPCK(hipStreamCreateWithFlags(&pr.st, hipStreamNonBlocking));
    }
    for (int i = 0; i < 8; ++i) HIPCK(hipEventCreateWithFlags(&e->push_ev[i], hipEventDisableTiming));
    e->peer_on = true;
    // Self-test before anything depends on it: every rank pushes one flag word into every peer and waits for theirs (a mapping that
    // cannot be written, a write this GPU's poll does not see: better a refusal here -- the caller then falls back to another
    // transport -- than a timeout in the middle of a run).
    {
        const double t_self = getenv("DZ_PEER_SELFTEST_S") ? std::max(0.5, atof(getenv("DZ_PEER_SELFTEST_S"))) : 30.0;
        int rc = peer_push(e, XK_HELLO, nullptr, 0, 0, 1ull);
        if (!rc) rc = peer_gate(e, XK_HELLO, 1ull, t_self);
        if (!rc && hipStreamSynchronize(e->stream) != hipSuccess) rc = fail("peer self-test: stream error");
        if (!rc) for (int r = 0; r < world; ++r) if (r != rank && hipStreamSynchronize(e->peers[r].st) != hipSuccess) rc = fail("peer self-test: copy stream error");
        if (!rc) rc = peer_check(e);
        if (rc) {
            const std::string why = g_err;
            e->peer_on = false; e->h_gate[2] = 0; e->h_gate[6] = 0;
            return fail("peer transport self-test failed: " + why);
        }
    }
    return 0;
}

int dz_peer_detach(dz_engine* e)
{   // stop using the peer transport (another rank could not attach: every rank moves on to the same fallback); the mappings stay until dz_destroy
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    for (auto& pr : e->peers) if (pr.st) HIPCK(hipStreamSynchronize(pr.st));
    e->peer_on = false;
    return 0;
}

int dz_exchange_stats(dz_engine* e, int64_t* exchanges, int64_t* gates, double* gate_wait_us)
{   // (k_peer_gate's counters; call after dz_sync)
    if (exchanges) *exchanges = e->z_pushed + e->pos_pushed + e->gs_pushed;
    if (gates) *gates = e->h_gate ? (int64_t)e->h_gate[1] : 0;
    if (gate_wait_us) *gate_wait_us = e->h_gate ? (double)e->h_gate[0] * 0.01 : 0.0;      // 100 MHz ticks
    return 0;
}

int dz_exchange_bytes(dz_engine* e, int64_t* history_bytes, int64_t* position_bytes, int64_t* sums_bytes)
{   // bytes this rank has handed to the transport FOR EACH other rank so far, by what they were
    if (!e) return fail("null engine");
    if (history_bytes) *history_bytes = e->bytes_z;
    if (position_bytes) *position_bytes = e->bytes_pos;
    if (sums_bytes) *sums_bytes = e->bytes_sums;
    return 0;
}

// A rendezvous of the ranks on the device: a one-element all-gather on the engine's stream, then a stream synchronise.  Ranks leave
// an RCCL collective within microseconds of each other (a host barrier's exits are spread by tens of microseconds), which is what a
// timed region over a few hundred microseconds wants in front of it.  No communicator (one GPU, or the host transport): just the sync.
int dz_comm_barrier(dz_engine* e)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    if (e->peer_on) {   // peer transport: every rank pushes its next "hello" number to every peer and waits for theirs
        e->hello_seq++;
        DZCK(peer_push(e, XK_HELLO, nullptr, 0, 0, (unsigned long long)e->hello_seq));
        DZCK(peer_gate(e, XK_HELLO, (unsigned long long)e->hello_seq));
        HIPCK(hipStreamSynchronize(e->stream));
        return peer_check(e);
    }
    if (!e->comm) return 0;
    if (!e->d_bar) DZCK(ealloc(e, &e->d_bar, (size_t)std::max(1, e->world)));
    ncclResult_t r = g_rccl.AllGather(e->d_bar + e->rank, e->d_bar, 1, ncclDouble, e->comm, e->stream);
    if (r != ncclSuccess) return fail(std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    HIPCK(hipStreamSynchronize(e->stream));
    return 0;
}

int dz_set_temperatures(dz_engine* e, const double* T, int32_t swaps)
{   // core.py:133-136 (the ladder is the host's), :185-221 (swaps != 0: one swap attempt per generation)
    HIPCK(hipSetDevice(e->c.device));
    if (swaps && e->world > 1) return fail("temperature swaps need all chains on one GPU");
    if (swaps && e->c.adapt_lag > 0) return fail("temperature swaps are not available with adapt_lag > 0");
    DZCK(sync_all(e));
    if (!e->d_Tc) DZCK(ealloc(e, &e->d_Tc, (size_t)e->p.N));
    HIPCK(hipMemcpy(e->d_Tc, T, sizeof(double) * e->p.N, hipMemcpyHostToDevice));
    e->p.Tc = e->d_Tc;
    e->tempering = swaps != 0;
    if (e->tempering && !e->d_tswap) { DZCK(ealloc(e, &e->d_tswap, (size_t)3 * std::max<int64_t>(1, e->c.trace_capacity))); e->p.tswap = e->d_tswap; }
    e->draws_gen = -1;
    return 0;
}
int dz_get_swaps(dz_engine* e, int64_t g0, int64_t ng, int32_t* out)
{
    HIPCK(hipSetDevice(e->c.device));
    if (!e->d_tswap || g0 < 0 || ng < 0 || g0 + ng > e->ntrace) return fail("swap log range");
    DZCK(sync_all(e));
    HIPCK(hipMemcpy(out, e->d_tswap + 3 * g0, sizeof(int32_t) * 3 * (size_t)ng, hipMemcpyDeviceToHost));
    return 0;
}

// A new run on the live engine, as run_dream(restart=True) starts one from the files the previous run left (core.py:46-62, 255-263;
// Dream.py:128-147): the archive as it stands is the new run's seed history (it is what the history file holds), the crossover /
// gamma-level probabilities stay (the two probability files), their accumulators start from zero (core.py:287-293), the generation
// counter restarts (Dream.iter = 0: generation 0 appends, the crossover burn-in runs again), the random contract takes the new key.
// Chain states come from dz_set_state afterwards.  Nothing is uploaded: a grown archive is copied device to device.
int dz_continue_run(dz_engine* e, int64_t history_capacity, int64_t trace_capacity, uint64_t seed, int32_t crossover_burnin)
{
    if (!e) return fail("null engine");
    HIPCK(hipSetDevice(e->c.device));
    if (e->peer_on || e->comm || e->world > 1) return fail("dz_continue_run: not on a sharded engine");
    if (e->tempering || e->p.Tc) return fail("dz_continue_run: not on a tempering engine");
    DZCK(adapt_flush(e));           // (the probabilities the new run starts from: everything that was due is in)
    DZCK(sync_all(e));
    if (e->copy_stream) HIPCK(hipStreamSynchronize(e->copy_stream));
    dz::Params& p = e->p;
    if (history_capacity < e->M) return fail("dz_continue_run: capacity below the rows already in the archive");
    // keep == 0 (the trace buffers: nothing of the old run's trace is used again): the old buffer is freed BEFORE the new one is allocated,
    // so the peak is max(old, new), not old + new (advisor, round 4); a failed allocation leaves the slot null and the engine says so
    auto swap_buffer = [&](auto** slot, size_t count, size_t keep) -> int {
        using T = std::remove_pointer_t<std::remove_pointer_t<decltype(slot)>>;
        auto drop_old = [&]() {
            auto it = std::find(e->to_free.begin(), e->to_free.end(), (void*)*slot);
            if (it != e->to_free.end()) e->to_free.erase(it);
            (void)hipFree((void*)*slot);
            *slot = nullptr;
        };
        if (!keep) drop_old();
        T* fresh = nullptr;
        if (dalloc(&fresh, count)) {
            if (!keep) { e->c.trace_capacity = 0; e->p.tcap = 0; e->broken = "dz_continue_run could not allocate the new trace buffers; the engine has none now: destroy it"; }
            return -1;
        }
        if (keep && hipMemcpy(fresh, *slot, sizeof(T) * keep, hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(fresh); return fail("dz_continue_run: device copy failed"); }
        if (keep) drop_old();
        *slot = fresh; e->to_free.push_back((void*)fresh);
        return 0;
    };
    if (history_capacity > e->c.history_capacity) {
        DZCK(swap_buffer(&p.Z, (size_t)history_capacity * p.ld, (size_t)e->M * p.ld));
        e->c.history_capacity = history_capacity;
    }
    if (trace_capacity > e->c.trace_capacity) {
        const size_t tc = (size_t)trace_capacity, nl = (size_t)p.nl;
        if (!e->c.trace_capacity) {
            DZCK(ealloc(e, &p.tX, tc * nl * p.ld)); DZCK(ealloc(e, &p.tlogp, tc * nl)); DZCK(ealloc(e, &p.tmoved, tc * nl));
            DZCK(ealloc(e, &p.tsnk, tc * nl)); DZCK(ealloc(e, &p.ttry, tc * nl)); DZCK(ealloc(e, &p.tcr, tc * nl));
        } else {
            DZCK(swap_buffer(&p.tX, tc * nl * p.ld, 0)); DZCK(swap_buffer(&p.tlogp, tc * nl, 0)); DZCK(swap_buffer(&p.tmoved, tc * nl, 0));
            DZCK(swap_buffer(&p.tsnk, tc * nl, 0)); DZCK(swap_buffer(&p.ttry, tc * nl, 0)); DZCK(swap_buffer(&p.tcr, tc * nl, 0));
        }
        p.tcap = (long long)tc; e->c.trace_capacity = trace_capacity;
    }
    e->c.seed = seed; p.k0 = (uint32_t)seed; p.k1 = (uint32_t)(seed >> 32);
    e->c.crossover_burnin = crossover_burnin; p.burnin = crossover_burnin;
    // delta_m / ncr_updates and the gamma analogue start from zero; the probabilities stay (layout: cr_probs|cr_delta|cr_n|g_probs|g_delta|g_n)
    HIPCK(hipMemset(p.cr_delta, 0, sizeof(double) * 2 * (size_t)p.ncr));
    HIPCK(hipMemset(p.g_delta, 0, sizeof(double) * 2 * (size_t)p.ngamma));
    if (e->d_redraw_count) HIPCK(hipMemset(e->d_redraw_count, 0, sizeof(unsigned long long)));
    HIPCK(hipStreamSynchronize(nullptr));
    e->gen = 0; std::fill(e->gen_c.begin(), e->gen_c.end(), (int64_t)0);
    e->napp = 0; e->ntrace = 0; e->draws_gen = -1; e->pending_accept = false; e->pending_slot = -1; e->stream_prop_gen = -1;
    e->have_logp = false; e->need_join = true; e->redraw_rounds = 0;
    e->ad_applied = -1; e->ad_made = -1;
    std::fill(e->own_init.begin(), e->own_init.end(), 0);
    if (e->adapt) { p.cp_prev = e->d_cp[0]; p.cp_new = e->d_cp[1]; e->cp_idx = 1; }
    e->params_uploaded = false;
    return 0;
}

int dz_step(dz_engine* e, int64_t generations)
{
    if (!e) return fail("null engine");
    HIPCK(hipSetDevice(e->c.device));
    if (!e->broken.empty()) return fail(e->broken);
    if (e->lk == LK_NONE) return fail("no likelihood set");
    if (e->M < 2 * e->c.depairs) return fail("history not seeded");
    if (e->c.trace_capacity && e->ntrace + generations > e->c.trace_capacity) return fail("trace capacity exceeded");
    if (!e->have_logp) {   // Dream.py:266-268
        DZCK(eval_logp(e, e->p.X, e->p.nl, e->p.lprior, e->p.llike));
        e->have_logp = true;
    }
    for (int c = 1; c < e->p.nl; ++c) if (e->gen_c[c] != e->gen_c[0]) return fail("chains are out of lockstep (single-chain stepping in progress)");
    e->gen = e->gen_c[0];
    e->p.own_cr = nullptr; e->p.own_g = nullptr;                   // lockstep generations read the shared probabilities;
    std::fill(e->own_init.begin(), e->own_init.end(), 0);          // a later single-chain step starts from them again
    const bool mega = mega_eligible(e);
    e->stream_prop_gen = -1;
    for (int64_t i = 0; i < generations;) {
        DZCK(peer_check(e));            // a gate that gave up is fatal: nothing more is queued behind it
        const int n = mega ? mega_segment(e, (uint32_t)e->gen, generations - i) : 0;
        if (n > 0) { DZCK(run_mega_segment(e, (uint32_t)e->gen, n, i + n < generations && mega_segment(e, (uint32_t)e->gen + (uint32_t)n, 1) > 0)); i += n; }
        else { DZCK(one_generation(e, 0, e->p.nl, (uint32_t)e->gen, true, i + 1 < generations && !(mega && mega_segment(e, (uint32_t)e->gen + 1, 1) > 0))); i += 1; }
        if (e->ra_stride > 0 && (++e->ra_n % e->ra_stride) == 0) {
            // keep the launch queue short: thousands of queued dispatches exhaust the runtime's kernarg/signal pools
            // and the whole pass then runs several times slower (measured; DESIGN.md "Host run-ahead")
            const int slot = (int)((e->ra_n / e->ra_stride) & 3), oldest = (slot + 1) & 3;
            HIPCK(hipEventRecord(e->ra_ev[slot], e->lane_stream[0]));
            e->ra_used[slot] = true;
            if (e->ra_used[oldest]) HIPCK(hipEventSynchronize(e->ra_ev[oldest]));
        }
    }
    // (adapt_lag 0: adaptation totals are left pending only between two launches of this loop; adapt_lag >= 1: updates that are due stay pending
    //  until a launch or a getter needs the state -- adapt_flush in dz_get_cr_state & co. -- instead of a launch of their own behind every dz_step)
    return e->c.adapt_lag > 0 ? 0 : adapt_flush(e);
}

int dz_step_range(dz_engine* e, int32_t c0, int32_t nc)
{
    if (!e) return fail("null engine");
    HIPCK(hipSetDevice(e->c.device));
    if (c0 < 0 || nc < 1 || c0 + nc > e->p.nl) return fail("bad chain range");
    if (e->lk == LK_NONE) return fail("no likelihood set");
    if (e->M < 2 * e->c.depairs) return fail("history not seeded");
    if (!e->have_logp) { DZCK(eval_logp(e, e->p.X, e->p.nl, e->p.lprior, e->p.llike)); e->have_logp = true; }
    for (int c = c0 + 1; c < c0 + nc; ++c) if (e->gen_c[c] != e->gen_c[c0]) return fail("chains of the range are at different generations");
    return one_generation(e, c0, nc, (uint32_t)e->gen_c[c0], false, false);
}

int dz_set_chain_state(dz_engine* e, int32_t c, const double* x, const double* prior, const double* like)
{
    HIPCK(hipSetDevice(e->c.device));
    if (c < 0 || c >= e->p.nl) return fail("bad chain index");
    if (!e->have_logp) { DZCK(eval_logp(e, e->p.X, e->p.nl, e->p.lprior, e->p.llike)); e->have_logp = true; }
    DZCK(upload_padded(e, e->p.X + (size_t)c * e->p.ld, x, 1, 0.0));
    if (prior && like) {
        HIPCK(hipMemcpy(e->p.lprior + c, prior, sizeof(double), hipMemcpyHostToDevice));
        HIPCK(hipMemcpy(e->p.llike + c, like, sizeof(double), hipMemcpyHostToDevice));
    } else DZCK(eval_logp(e, e->p.X + (size_t)c * e->p.ld, 1, e->p.lprior + c, e->p.llike + c));   // Dream.py:266-268
    return 0;
}

int dz_get_chain_state(dz_engine* e, int32_t c, double* x, double* prior, double* like)
{
    HIPCK(hipSetDevice(e->c.device));
    if (c < 0 || c >= e->p.nl) return fail("bad chain index");
    if (x) DZCK(download_rows(e, x, e->p.X + (size_t)c * e->p.ld, 1));
    DZCK(sync_all(e));
    if (prior) HIPCK(hipMemcpy(prior, e->p.lprior + c, sizeof(double), hipMemcpyDeviceToHost));
    if (like) HIPCK(hipMemcpy(like, e->p.llike + c, sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int dz_get_chain_probs(dz_engine* e, int32_t c, double* cr_probs, double* gamma_probs)
{   // Dream.CR_probabilities / Dream.gamma_probabilities of the instance that drives chain c (Dream.py:375, :383)
    HIPCK(hipSetDevice(e->c.device));
    if (c < 0 || c >= e->p.nl) return fail("bad chain index");
    DZCK(adapt_flush(e));
    DZCK(sync_all(e));
    const bool own = e->d_own_cr && c < (int)e->own_init.size() && e->own_init[c];
    if (cr_probs) HIPCK(hipMemcpy(cr_probs, own ? e->d_own_cr + (size_t)c * e->p.ncr : e->p.cr_probs, sizeof(double) * e->p.ncr, hipMemcpyDeviceToHost));
    if (gamma_probs) HIPCK(hipMemcpy(gamma_probs, own ? e->d_own_g + (size_t)c * e->p.ngamma : e->p.g_probs, sizeof(double) * e->p.ngamma, hipMemcpyDeviceToHost));
    return 0;
}

int dz_sync(dz_engine* e)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    return peer_check(e);
}
int dz_trace_reset(dz_engine* e) { e->ntrace = 0; return 0; }
int64_t dz_generation(dz_engine* e) { return e->gen; }
int64_t dz_redraw_rounds(dz_engine* e)
{   // launches of the multi-kernel path's redraw rounds + (block, round) pairs of the persistent kernel's
    unsigned long long dev = 0;
    if (e->d_redraw_count && hipSetDevice(e->c.device) == hipSuccess && sync_all(e) == 0)
        (void)hipMemcpy(&dev, e->d_redraw_count, sizeof dev, hipMemcpyDeviceToHost);
    return e->redraw_rounds + (int64_t)dev;
}
const char* dz_last_kernel_variant(dz_engine* e) { return e->last_variant.c_str(); }

int dz_get_state(dz_engine* e, double* X, double* prior, double* like)
{
    HIPCK(hipSetDevice(e->c.device));
    if (X) DZCK(download_rows(e, X, e->p.X, (size_t)e->p.nl));
    DZCK(sync_all(e));
    if (prior) HIPCK(hipMemcpy(prior, e->p.lprior, sizeof(double) * e->p.nl, hipMemcpyDeviceToHost));
    if (like) HIPCK(hipMemcpy(like, e->p.llike, sizeof(double) * e->p.nl, hipMemcpyDeviceToHost));
    return 0;
}

// device trace rows are chain-major [nl][tcap][ld]; X_out[c][i][:] = sample g0+i of chain c at X_out + (c*chain_stride + i)*d
static int download_trace_rows(dz_engine* e, double* X, int64_t g0, int64_t ng, int64_t chain_stride_rows)
{
    const size_t nl = e->p.nl, d = e->p.d, ld = e->p.ld, tcap = (size_t)e->p.tcap;
    DZCK(sync_all(e));
    if (ng == 0) return 0;
    // pinning the destination lets the DMA engines write it directly (several times the pageable rate); not fatal if refused
    const size_t span = sizeof(double) * d * ((nl - 1) * (size_t)chain_stride_rows + (size_t)ng);
    const bool pinned = span >= ((size_t)64 << 20) && hipHostRegister(X, span, hipHostRegisterDefault) == hipSuccess;
    if (!pinned) (void)hipGetLastError();
    int rc = 0;
    auto copy2d = [&](double* dst, size_t dpitch, const double* src, size_t spitch, size_t width, size_t height) {
        if (pinned) return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, e->stream);
        return d2h_2d(e, dst, dpitch, src, spitch, width, height) ? hipErrorUnknown : hipSuccess;
    };
    if ((size_t)ng == tcap && (size_t)chain_stride_rows == (size_t)ng) {
        // whole buffer: the chains' blocks are back to back on both sides -- one strided copy
        if (copy2d(X, sizeof(double) * d, e->p.tX, sizeof(double) * ld, sizeof(double) * d, nl * (size_t)ng) != hipSuccess) rc = -1;
    } else {
        for (size_t c = 0; c < nl && !rc; ++c)
            if (copy2d(X + c * (size_t)chain_stride_rows * d, sizeof(double) * d, e->p.tX + (c * tcap + (size_t)g0) * ld, sizeof(double) * ld,
                       sizeof(double) * d, (size_t)ng) != hipSuccess) rc = -1;
    }
    if (hipStreamSynchronize(e->stream) != hipSuccess) rc = -1;
    if (pinned) (void)hipHostUnregister(X);
    return rc ? fail(std::string("trace download: ") + hipGetErrorString(hipGetLastError())) : 0;
}

int dz_get_trace(dz_engine* e, int64_t g0, int64_t ng, double* X, double* logp, uint8_t* moved, int32_t* try_idx, int32_t* cr_idx, uint8_t* snooker)
{
    HIPCK(hipSetDevice(e->c.device));
    if (g0 < 0 || ng < 0 || g0 + ng > e->ntrace) return fail("trace range");
    const size_t nl = e->p.nl, o = (size_t)g0 * nl, n = (size_t)ng * nl;
    if (X && ng > 0) {   // generation-major output [ng][nl][d]: one strided copy per chain (rows nl*d apart on the host side)
        const size_t d = e->p.d, ld = e->p.ld, tcap = (size_t)e->p.tcap;
        DZCK(sync_all(e));
        for (size_t c = 0; c < nl; ++c)
            DZCK(d2h_2d(e, X + c * d, sizeof(double) * nl * d, e->p.tX + (c * tcap + (size_t)g0) * ld, sizeof(double) * ld, sizeof(double) * d, (size_t)ng));
    }
    DZCK(sync_all(e));
    if (logp) HIPCK(hipMemcpy(logp, e->p.tlogp + o, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (moved) HIPCK(hipMemcpy(moved, e->p.tmoved + o, n, hipMemcpyDeviceToHost));
    if (try_idx) HIPCK(hipMemcpy(try_idx, e->p.ttry + o, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (cr_idx) HIPCK(hipMemcpy(cr_idx, e->p.tcr + o, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    if (snooker) HIPCK(hipMemcpy(snooker, e->p.tsnk + o, n, hipMemcpyDeviceToHost));
    return 0;
}

int dz_get_trace_chains(dz_engine* e, int64_t g0, int64_t ng, double* X, int64_t chain_stride_rows, double* logp)
{
    HIPCK(hipSetDevice(e->c.device));
    if (g0 < 0 || ng < 0 || g0 + ng > e->ntrace) return fail("trace range");
    if (chain_stride_rows < ng) return fail("bad destination");
    if (X) DZCK(download_trace_rows(e, X, g0, ng, chain_stride_rows));
    if (logp && ng > 0) {   // log_ps chain by chain: transposed on the device, then one strided copy
        const size_t nl = e->p.nl, need = nl * (size_t)ng;
        if (need > e->qpart_len) {
            DZCK(sync_all(e));
            if (e->d_qpart) (void)hipFree(e->d_qpart);
            e->d_qpart = nullptr; e->qpart_len = 0;
            DZCK(dalloc(&e->d_qpart, need));
            e->qpart_len = need;
        }
        hipLaunchKernelGGL(dz::k_transpose_logp, dim3((unsigned)((need + 255) / 256)), dim3(256), 0, e->stream, (const double*)e->p.tlogp, (int)nl, g0, (int)ng, e->d_qpart);
        DZCK(launch_check("k_transpose_logp"));
        DZCK(d2h_2d(e, logp, sizeof(double) * (size_t)chain_stride_rows, e->d_qpart, sizeof(double) * (size_t)ng, sizeof(double) * (size_t)ng, nl));
    }
    return 0;
}

// Samples [g0, g0+ng) of every local chain into X (as dz_get_trace_chains) WITHOUT stopping the engine: the copy is queued on a
// stream of its own behind everything queued so far on the engine's streams and the call returns; generations stepped afterwards
// run while the DMA engines move the rows.  X must be page-locked (dz_host_register) -- a copy into pageable memory would block --
// and stay untouched until dz_trace_download_wait.  One 3-D copy (row = d of ld doubles, rows = generations, slices = chains).
int dz_trace_download_begin(dz_engine* e, int64_t g0, int64_t ng, double* X, int64_t chain_stride_rows)
{
    HIPCK(hipSetDevice(e->c.device));
    if (g0 < 0 || ng < 0 || g0 + ng > e->ntrace) return fail("trace range");
    if (chain_stride_rows < ng || !X) return fail("bad destination");
    if (ng == 0) return 0;
    if (!e->copy_stream) {
        HIPCK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        for (int s = 0; s < e->nlanes; ++s) HIPCK(hipEventCreateWithFlags(&e->copy_ev[s], hipEventDisableTiming));
    }
    for (int s = 0; s < e->nlanes; ++s) {
        HIPCK(hipEventRecord(e->copy_ev[s], e->lane_stream[s]));
        HIPCK(hipStreamWaitEvent(e->copy_stream, e->copy_ev[s], 0));
    }
    const size_t nl = e->p.nl, d = e->p.d, ld = e->p.ld, tcap = (size_t)e->p.tcap;
    hipMemcpy3DParms c;
    memset(&c, 0, sizeof(c));
    c.srcPtr = make_hipPitchedPtr((void*)(e->p.tX + (size_t)g0 * ld), sizeof(double) * ld, sizeof(double) * ld, tcap);
    c.dstPtr = make_hipPitchedPtr((void*)X, sizeof(double) * d, sizeof(double) * d, (size_t)chain_stride_rows);
    c.extent = make_hipExtent(sizeof(double) * d, (size_t)ng, nl);
    c.kind = hipMemcpyDeviceToHost;
    HIPCK(hipMemcpy3DAsync(&c, e->copy_stream));
    return 0;
}
int dz_trace_download_wait(dz_engine* e)
{
    HIPCK(hipSetDevice(e->c.device));
    if (e->copy_stream) HIPCK(hipStreamSynchronize(e->copy_stream));
    return 0;
}

// page-lock a host array ahead of a download (first-touch faults and pinning then overlap the run instead of the copy)
int dz_host_register(void* ptr, int64_t bytes)
{
    // Page-locking a freshly allocated array is mostly the kernel faulting its pages in, one at a time, on the calling thread (6.5 GB:
    // 0.27 s).  Large ranges are first populated by several threads after asking for huge pages -- madvise(MADV_POPULATE_WRITE) where the
    // kernel has it (5.14+), else an atomic add of zero to a byte of every page: either way nothing another thread writes meanwhile is
    // lost --; hipHostRegister then finds the pages present.  DZ_PIN_THREADS (default 8, 1 = off).
    int nt = 8;
    if (const char* s = getenv("DZ_PIN_THREADS")) nt = std::max(1, std::min(64, atoi(s)));
    if (bytes >= ((int64_t)64 << 20) && nt > 1) {
        const uintptr_t page = 4096, lo = ((uintptr_t)ptr + page - 1) & ~(page - 1), hi = ((uintptr_t)ptr + (uintptr_t)bytes) & ~(page - 1);
        if (hi > lo) {
            (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
            const uintptr_t npages = (hi - lo) / page;
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([=] {
                    const uintptr_t i0 = npages * t / nt, i1 = npages * (t + 1) / nt;
                    if (i1 <= i0) return;
#ifdef MADV_POPULATE_WRITE
                    if (madvise((void*)(lo + i0 * page), (i1 - i0) * page, MADV_POPULATE_WRITE) == 0) return;
#endif
                    unsigned char* q = (unsigned char*)lo;
                    for (uintptr_t i = i0; i < i1; ++i) __atomic_fetch_add(q + i * page, (unsigned char)0, __ATOMIC_RELAXED);
                });
            for (auto& x : th) x.join();
        }
    }
    if (hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return fail("hipHostRegister failed"); }
    return 0;
}
int dz_host_unregister(void* ptr)
{
    if (hipHostUnregister(ptr) != hipSuccess) { (void)hipGetLastError(); return fail("hipHostUnregister failed"); }
    return 0;
}

int dz_get_history(dz_engine* e, double* Z, int64_t cap_rows, int64_t* rows)
{
    HIPCK(hipSetDevice(e->c.device));
    if (rows) *rows = e->M;
    if (Z) {
        if (cap_rows < e->M) return fail("buffer too small");
        // peer transport with history_lag: the rows of the last appends may still be on their way
        if (e->peer_on && e->z_gated < e->napp) { DZCK(peer_gate(e, XK_Z, (unsigned long long)e->napp)); e->z_gated = e->napp; }
        DZCK(download_rows(e, Z, e->p.Z, (size_t)e->M));
        DZCK(peer_check(e));
    }
    return 0;
}

int dz_get_history_range(dz_engine* e, int64_t row0, int64_t nrows, double* Z)
{   // rows [row0, row0 + nrows) of the archive (what a continued run appends to the history file of the run before)
    HIPCK(hipSetDevice(e->c.device));
    if (row0 < 0 || nrows < 0 || row0 + nrows > e->M || !Z) return fail("history range");
    if (e->peer_on && e->z_gated < e->napp) { DZCK(peer_gate(e, XK_Z, (unsigned long long)e->napp)); e->z_gated = e->napp; }
    return download_rows(e, Z, e->p.Z + (size_t)row0 * e->p.ld, (size_t)nrows);
}

int dz_history_checksum(dz_engine* e, uint64_t* sum, int64_t* rows)
{   // every appended row of every rank has arrived (as dz_get_history), then one pass over the archive on the device
    HIPCK(hipSetDevice(e->c.device));
    if (!sum) return fail("null argument");
    if (e->peer_on && e->z_gated < e->napp) { DZCK(peer_gate(e, XK_Z, (unsigned long long)e->napp)); e->z_gated = e->napp; }
    DZCK(join_all(e));
    if (!e->d_bar) DZCK(ealloc(e, &e->d_bar, (size_t)std::max(1, e->world)));
    unsigned long long* acc = (unsigned long long*)e->d_bar;         // (the rendezvous buffer: idle between barriers)
    HIPCK(hipMemsetAsync(acc, 0, sizeof(unsigned long long), e->stream));
    if (e->M > 0) {
        const unsigned blocks = (unsigned)std::min<int64_t>((e->M + 3) / 4, (int64_t)e->num_cu * 16);
        hipLaunchKernelGGL(dz::k_checksum, dim3(blocks), dim3(256), 0, e->stream, (const double*)e->p.Z, (long long)e->M, e->p.d, e->p.ld, acc);
        DZCK(launch_check("k_checksum"));
    }
    unsigned long long h = 0;
    HIPCK(hipMemcpyAsync(&h, acc, sizeof h, hipMemcpyDeviceToHost, e->stream));
    DZCK(sync_all(e));
    *sum = (uint64_t)h;
    if (rows) *rows = e->M;
    return 0;
}

static int get_shared(dz_engine* e, const double* probs, const double* delta, const double* n, int nb, double* o_probs, double* o_delta, double* o_n)
{
    DZCK(sync_all(e));
    if (o_probs) HIPCK(hipMemcpy(o_probs, probs, sizeof(double) * nb, hipMemcpyDeviceToHost));
    if (o_delta) HIPCK(hipMemcpy(o_delta, delta, sizeof(double) * nb, hipMemcpyDeviceToHost));
    if (o_n) HIPCK(hipMemcpy(o_n, n, sizeof(double) * nb, hipMemcpyDeviceToHost));
    return 0;
}
int dz_get_cr_state(dz_engine* e, double* probs, double* delta_m, double* n_updates)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(adapt_flush(e));           // (adapt_lag >= 1: the updates the next generation decides with are in; later ones stay pending)
    return get_shared(e, e->p.cr_probs, e->p.cr_delta, e->p.cr_n, e->c.ncr, probs, delta_m, n_updates);
}
int dz_get_gamma_state(dz_engine* e, double* probs, double* delta_m, double* n_updates)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(adapt_flush(e));
    return get_shared(e, e->p.g_probs, e->p.g_delta, e->p.g_n, e->c.ngamma, probs, delta_m, n_updates);
}

static int chain_moments(dz_engine* e)
{
    if (e->ntrace < 2) return fail("need at least 2 traced generations");
    const dz::Params& p = e->p;
    hipLaunchKernelGGL(dz::k_chain_moments, dim3((p.d + 127) / 128, p.nl), dim3(128), 0, e->stream, p.tX, p.nl, p.d, p.ld, p.tcap, (int)e->ntrace, e->d_cmean, e->d_cvar);
    return launch_check("chain moments");
}
int dz_get_chain_moments(dz_engine* e, double* mean, double* var)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(chain_moments(e));
    DZCK(sync_all(e));
    const size_t n = (size_t)e->p.nl * e->p.d;
    HIPCK(hipMemcpy(mean, e->d_cmean, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(var, e->d_cvar, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}
int dz_get_rhat(dz_engine* e, double* rhat)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(chain_moments(e));
    const dz::Params& p = e->p;
    hipLaunchKernelGGL(dz::k_rhat, dim3((p.d + 15) / 16), dim3(1024), sizeof(double) * 16 * (size_t)((p.nl + 63) / 64), e->stream, e->d_cmean, e->d_cvar, p.nl, p.d, (int)e->ntrace, e->d_rhat);
    DZCK(launch_check("rhat"));
    DZCK(sync_all(e));
    HIPCK(hipMemcpy(rhat, e->d_rhat, sizeof(double) * p.d, hipMemcpyDeviceToHost));
    return 0;
}

static int need_scratch(dz_engine* e, size_t rows)
{
    if (e->scratch_rows >= rows) return 0;
    if (e->d_scratch) (void)hipFree(e->d_scratch);
    e->d_scratch = nullptr; e->scratch_rows = 0;
    HIPCK(hipMalloc((void**)&e->d_scratch, sizeof(double) * (rows * e->p.ld + 4 * rows)));
    e->scratch_rows = rows;
    return 0;
}

int dz_eval_logp(dz_engine* e, const double* X, int64_t n, double* prior, double* like)
{
    HIPCK(hipSetDevice(e->c.device));
    if (n > (1 << 22)) return fail("too many points in one call");
    DZCK(need_scratch(e, (size_t)n));
    double* pts = e->d_scratch; double* dp = pts + (size_t)n * e->p.ld; double* dl = dp + n;
    DZCK(upload_padded(e, pts, X, (int)n, 0.0));
    DZCK(eval_logp(e, pts, (int)n, dp, dl));
    DZCK(sync_all(e));
    HIPCK(hipMemcpy(prior, dp, sizeof(double) * n, hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(like, dl, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

int dz_debug_propose(dz_engine* e, int32_t chain_local, int64_t gen, int32_t phase, const double* base,
                     int32_t run_snooker, int32_t cr_idx, int32_t delta, int32_t glev, double* pts, double* slogp)
{
    HIPCK(hipSetDevice(e->c.device));
    const int n = phase == 0 ? e->p.k : e->p.k - 1;
    if (n < 1) return fail("no reference phase for multitry=1");
    DZCK(need_scratch(e, (size_t)n + 1));
    double* dbase = e->d_scratch; double* dout = dbase + e->p.ld; double* dsl = dout + (size_t)n * e->p.ld;
    DZCK(upload_padded(e, dbase, base, 1, 0.0));
    NCH_DISPATCH(e, hipLaunchKernelGGL(dz::k_propose_debug<NCH>, dim3(1), dim3(64), 0, e->stream, e->p, phase, (uint32_t)gen, (uint32_t)e->M,
                                       chain_local, n, dbase, dout, dsl, run_snooker, cr_idx, delta, glev));
    DZCK(launch_check("propose(debug)"));
    DZCK(download_rows(e, pts, dout, (size_t)n));
    HIPCK(hipMemcpy(slogp, dsl, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

int dz_profile_enable(dz_engine* e, int32_t on)
{
    e->prof = on != 0;
    if (on > 1) {   // on = number of event pairs to pre-create, so that the timed region only records
        HIPCK(hipSetDevice(e->c.device));
        while ((int64_t)e->ev_pool.size() < 2 * (int64_t)on) { hipEvent_t x; HIPCK(hipEventCreate(&x)); e->ev_pool.push_back(x); }
    }
    return 0;
}
int dz_profile_reset(dz_engine* e)
{
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    for (auto& v : e->ev) { for (auto& pr : v) { e->ev_pool.push_back(pr.first); e->ev_pool.push_back(pr.second); } v.clear(); }
    if (e->prof) {   // class 6: event pairs with nothing in between = what one bracket adds to every measured launch
        for (int i = 0; i < 64; ++i) { ProfScope ps(e, PR_EMPTY, e->stream); }
        DZCK(sync_all(e));
    }
    return 0;
}
int dz_profile_get(dz_engine* e, int32_t which, double* total_ms, int64_t* launches)
{
    if (which < 0 || which >= PR_COUNT) return fail("bad profile class");
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    double tot = 0.0;
    for (auto& pr : e->ev[which]) { float ms = 0.f; HIPCK(hipEventElapsedTime(&ms, pr.first, pr.second)); tot += ms; }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int64_t)e->ev[which].size();
    return 0;
}

int dz_profile_get_list(dz_engine* e, int32_t which, double* ms, int64_t cap, int64_t* launches)
{   // the individual launches of a class, in launch order (dz_profile_get gives their sum)
    if (which < 0 || which >= PR_COUNT) return fail("bad profile class");
    HIPCK(hipSetDevice(e->c.device));
    DZCK(sync_all(e));
    int64_t i = 0;
    for (auto& pr : e->ev[which]) { if (i >= cap) break; float t = 0.f; HIPCK(hipEventElapsedTime(&t, pr.first, pr.second)); ms[i++] = t; }
    if (launches) *launches = (int64_t)e->ev[which].size();
    return 0;
}

}  // extern "C"
