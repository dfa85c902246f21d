// dz_kernels.h -- HIP kernels of the MT-DREAM(ZS) generation step for gfx950.
//
// One generation (schedule S2) =
//   k_propose(phase 0) -> k_logp_* -> k_propose(phase 1) -> k_logp_* -> k_accept
//   [-> adaptation kernels while g <= crossover_burnin]
// Mapping: one 64-lane wave per proposal point; lane l owns dimensions 128*it + 2*l + {0,1}
// (16-byte loads of two consecutive doubles, rows padded to a multiple of 16 doubles).
// Citations are to /root/reference/pydream/Dream.py unless stated.
#pragma once
#include "dz_device.h"

namespace dz {

constexpr int MAXK = 32;      // multitry limit (tries 17..32: the multi-kernel path; the persistent kernels hold a generation's draw slots in one wave's lanes: multitry <= 15)
constexpr int MAXPAIR = 8;    // DEpairs limit

// per-chain control decisions of one generation, computed once (lane-parallel) instead of by every wave
struct ChainCtl { int snk, cr_idx, delta, glev; double u_sel, u_acc; };

struct Params {
    int N, nl, off, d, ld, k, depairs, ncr, ngamma, thin, burnin, adapt_cr, adapt_g, hard;
    uint32_t k0, k1;
    double lamb, zeta, snooker, pgu, T;
    double* Z;
    double *X, *lprior, *llike;
    double *P, *R;
    double *p_prior, *p_like, *p_slogp, *r_prior, *r_like, *r_slogp, *cur_snk;
    const double *mins, *maxs, *gtab;
    double *cr_probs, *cr_delta, *cr_n, *g_probs, *g_delta, *g_n;
    double *cp_prev, *cp_new;
    // trace
    double *tX, *tlogp; uint8_t *tmoved, *tsnk; int32_t *ttry, *tcr;
    long long tcap;      // trace capacity in generations; tX is CHAIN-major [nl][tcap][ld] (a chain's samples are one block, as run_dream returns them)
    // likelihood / prior
    const int32_t* pkind; const double *pa, *pb, *plogb; int have_prior;      // plogb[j] = dlog(pb[j]), made once by k_prior_consts
    // pc2[j]: the second constant a prior evaluation needs -- normal (kind 1): 1 / scale (z = (x - loc) pc2: no division per try and
    // dimension); uniform (kind 2): loc + scale, the upper end of the support.  prior_nonormal: no dimension has a normal prior: the log
    // prior of a point is either -inf or ONE constant (the sum of -log scale over the uniform dimensions)
    const double* pc2; int prior_nonormal;
    int prior_const;     // ... and every uniform support contains the hard boundaries' box: after the boundary handling a proposal is inside every support,
                         // its log prior is that constant (set by the host per launch: mega_set_pb_lds)
    const double *mu, *Mt; double logF; int tri; int J; const double* mixF;
    // triangular factor, packed for k_logp_mvn_lds: k-row r keeps its first 16*(r/16+1) columns (the row tiles that
    // use it), rows back to back; mtp_len doubles (even)
    const double* Mtp; int mtp_len;
    const double* Tc;        // [N] per-chain temperatures (parallel tempering, core.py:133-136) or null: every chain at T
    int32_t* tswap;          // [trace capacity][3] swap log: chain a, chain b, accepted
    int mu_zero;     // every entry of mu is +0.0: x - mu == x exactly, kernels may skip the subtraction
    // wave-uniform Philox outputs of one generation, precomputed lane-parallel (k_draws / k_accept):
    // [nl][nslots] uint4; slot 0..2 = control stream idx 0..2, then npt slots per (phase, try)
    const uint4* draws; uint4* draws_next; int nslots, npt;
    const ChainCtl* ctl; ChainCtl* ctl_next;
    int* sel;      // [nl] selected try | anyfinite<<8, written by the reference-phase proposal wave
    double ec1, ec0;          // e ~ U(-lamb, lamb) from a 16-bit draw h as fma(h, ec1, ec0) (dz_device.h uniform16)
    uint32_t crthr[32];       // crossover_threshold(CR_values[m]), m < ncr: `U_j < CR` as an integer test on the 16-bit draw
    unsigned long long pgu_thr;   // ceil(p_gamma_unity 2^53): u53(hi, lo) < p_gamma_unity as an integer test on the 53-bit draw (u53_below)
    unsigned long long snk_thr;   // ceil(snooker 2^53): the same for set_snooker's draw (0 when snooker == 0)
    const uint8_t* redo;      // redraw round only: [nl] 1 = every try of the chain's current set is impossible (k_redo_flags); a listed chain whose flag has cleared is left alone
    const int32_t* redo_list; // redraw round only (Dream.py:281-289; redraw_impossible_sets): the local chains whose proposal set is drawn again -- wave w works on chain redo_list[w / split]; null otherwise
    // Dream.astep driven chain by chain (dz_step_range, schedule S1): every Dream instance keeps its OWN copy of the crossover / gamma-level
    // probabilities (Dream.py:375, :383, :497, :538, :409-415), refreshed only by its own adaptation updates; [nl][ncr] / [nl][ngamma], or
    // null in lockstep mode (every chain reads the shared vectors)
    const double *own_cr, *own_g;
    unsigned long long* redraw_count;    // redraw rounds made inside the persistent kernel (one count per block and round), or null
    // large-d MVN likelihood (k_logp_mvn_gemm / k_logp_mvn_mfma_tiled) without a k_q_finish launch: the kernels that consume the proposal
    // set's (qfin_p) / reference set's (qfin_r) log likelihoods add the row-tile sums [row tile][point] themselves (like_from_q); the
    // points are those of local chains [qfin_c0, qfin_c0 + qfin_nc).  Null: p_like / r_like hold the values already.
    const double *qfin_p, *qfin_r; int qfin_nrt, qfin_c0, qfin_nc;
    int pb_lds;      // persistent kernel, full-code instantiations: the prior / boundary constants are staged in LDS (PBConsts; set by the host when they fit)
    unsigned long long* dbg;  // cycle-stamp buffer of instrumented builds (-DDZ_EXPERIMENTS, dz_experiments.h); null otherwise
    const void* udata;        // a user's device likelihood inside the persistent kernel (dz_user_generations.hip.in): its data block
};
// The layout generation of Params / Publish as a code object built at run time sees them (the names of the persistent kernels such an object exports
// carry it: dz_user_generations_v<N> / dz_user_generations_full_v<N>; an object built against other headers simply does not have them)
#define DZ_USER_ABI 6
#define DZ_USER_STR2(x) #x
#define DZ_USER_STR(x) DZ_USER_STR2(x)
// What a persistent launch inside the crossover burn-in (one generation per launch) leaves behind besides the new states:
// to: the published positions (set_current_position_arr, Dream.py:364-366, :447-449: [N][ld], row = global chain), or null outside the
// burn-in; PR / PC / shift: when set, the block also makes the adaptation sums of its unit of 16 chains (adapt_unit_sums; contract v3),
// shift = the previous published positions (row 0 is the shift of the column sums).
// sh: the shared adaptation state the launch's chains decide with (cr_probs|cr_delta|cr_n|g_probs|g_delta|g_n, core.py:287-293).
// TOT / CNT: when set, the totals of the PREVIOUS generation's sums, not applied yet: every block makes the update for itself in its
// prologue (adapt_apply_wave: sd, weights, the bins' dot products, the new probabilities -- the same arithmetic everywhere, so every
// block decides with the same bits) and block 0 leaves the new state in sh_out (another buffer than sh: blocks start at different times).
// c0: the first local chain of this launch (a generation may be two launches: whole rounds of 16-chain blocks, then the remainder in smaller
// blocks -- run_mega_segment); the launch's chains end at the engine's nl or, for the first part, at c1.
// adapt_lag >= 1 (round 6; dz_config.adapt_lag): generation g of the burn-in decides with the probabilities as they were after the updates
// of generations <= g - 1 - lag, so a launch holds up to lag + 1 burn-in generations (`multi`).  Then: the totals of every generation are kept
// in rings of lag + 1 slots (slot = generation mod (lag + 1)): the units' sums go to PR / PC + slot * pr_stride / pc_stride; DOT / CNTR
// [slot][nbp] are the bins' dot products and counts of the generations whose update is still pending -- pend0 .. pend1 - 1, all of them made
// by earlier launches -- which the launch's prologue applies in order (adapt_pending_apply), leaving the probabilities of its i-th generation
// in an LDS table; x0ring [2 (lag + 1)][ld]: global chain 0's position after generation h at slot h mod 2 (lag + 1), x0start: before
// generation 0 -- the shift of generation g's column sums is that of generation g - 1 - lag.  `to` then receives the positions of the
// launch's LAST generation only (what a later launch that is not fused measures its jumps from).
struct Publish {
    double* to; const double* shift; double* PR; double* PC; const double* sh; double* sh_out; const double* TOT; const double* CNT; int c0, c1;
    int multi = 0, lag = 0, burnin = 0, nbp = 0; long long pend0 = 0, pend1 = 0;
    const double* DOT = nullptr; const double* CNTR = nullptr; double* x0ring = nullptr; const double* x0start = nullptr;
    long long pr_stride = 0, pc_stride = 0;
    // multi == 2 (any instantiation): several burn-in generations per launch whose unit sums are made BEHIND the launch -- `to` is a ring of
    // lag + 2 position arrays, pos_stride doubles apart (generation g's at slot g mod (lag + 2)); PG [generations of the launch][nbp]: the
    // probabilities each of them decided with (block 0 leaves its table there), read by k_adapt_partials_ring
    double* PG = nullptr; long long pos_stride = 0;
};

// (a code object built at run time around a user's device function takes both structures: whoever changes their layout raises DZ_USER_ABI with it)
static_assert(sizeof(Params) == 784 && sizeof(Publish) == 168, "Params / Publish changed: raise DZ_USER_ABI (a stale user code object must not be looked up by the old name) and these sizes");

// per-phase cycle stamps: defined by dz_experiments.h in instrumented builds only (tools/variants.sh), no-ops in the product
#ifdef DZ_EXPERIMENTS
#include "dz_experiments.h"
#else
#define DZ_STAMP(p_, phase_, c_, i_) do { } while (0)
#define DZ_LSTAMP(p_, w_, i_) do { } while (0)
#define DZ_MSTAMP(i_) do { } while (0)
#define DZ_WSTAMP(i_) do { } while (0)
#define DZ_W0STAMP(i_) do { } while (0)
#endif

// offset of k-row r in the packed triangular layout: row block b = r/16 has 16*(b+1) columns
__host__ __device__ inline int tri_row_offset(int r) { const int b = r >> 4; return 128 * b * (b + 1) + (r - 16 * b) * 16 * (b + 1); }

// temperature of local chain c (Dream.astep's T argument, Dream.py:193)
DZ_DEV double chain_T(const Params& p, int c) { return p.Tc ? p.Tc[p.off + c] : p.T; }

struct StepFlags { bool snk; int cr_idx, delta, glev; };

DZ_DEV StepFlags step_flags_from(const Params& p, const Ctrl& u, const double* cr_probs, const double* g_probs)
{
    StepFlags f;
    f.snk = (p.snooker != 0.0) && (u.u_snk < p.snooker);                      // set_snooker :542-554
    f.cr_idx = invcdf(cr_probs, p.ncr, u.u_cr);                                // set_CR :556-569
    f.delta = p.depairs > 1 ? 1 + (int)floor(u.u_de * (double)p.depairs) : 1;  // set_DEpair :571-583
    f.glev = 1 + invcdf(g_probs, p.ngamma, u.u_glev);                          // set_gamma_level :585-599
    return f;
}
DZ_DEV StepFlags step_flags(const Params& p, const Ctrl& u) { return step_flags_from(p, u, p.cr_probs, p.g_probs); }
// ... of LOCAL chain c: its own copy of the probabilities under single-chain stepping (Params::own_cr), else the shared ones
DZ_DEV StepFlags step_flags_chain(const Params& p, const Ctrl& u, int c)
{
    return step_flags_from(p, u, p.own_cr ? p.own_cr + (size_t)c * p.ncr : p.cr_probs, p.own_g ? p.own_g + (size_t)c * p.ngamma : p.g_probs);
}

// mt_choose_proposal_pt :883-917.  Lane i < k evaluates try i's weight (one dexp per wave instead of
// k); the sums run over the tries in order, so every lane ends with the same scalars.
// BIG: k may exceed 16 (tries in lanes 0..k-1 <= 31; the multi-kernel path's callers) -- the maximum is exact whatever the order, the
// sums are sequential in try order, so the bits do not depend on it.
template <bool BIG = false>
DZ_DEV int mt_select_vals(int k, double lp, double u_sel, int lane, bool* anyfinite)
{   // lp: lane i < k holds prior_i + T like_i (:900), other lanes -inf
    const double rmx = rowmax16(lp);
    const double mx = BIG ? fmax(readlane_f64(rmx, 0), readlane_f64(rmx, 16)) : readlane_f64(rmx, 0);
    *anyfinite = __any(lane < k && is_finite(lp)) != 0;
    const double w = dexp(lp - mx);
    double S = 0.0;
    for (int i = 0; i < k; ++i) S = S + readlane_f64(w, i);
    const double pr = w / S;                          // lane i: probability of try i (:907)
    // first i with u_sel < cum_i, else k-1 (:908-913); without a data-dependent branch (the values are wave-uniform, but live
    // in vector registers: a branch on them costs a compare -> mask -> branch round trip per step)
    double cum = 0.0; int sel = k - 1; bool found = false;
    for (int i = 0; i < k; ++i) {
        cum = cum + readlane_f64(pr, i);
        const bool hit = !found && (u_sel < cum);
        sel = hit ? i : sel; found = found || hit;
    }
    return __builtin_amdgcn_readfirstlane(sel);
}
DZ_DEV int mt_select(const Params& p, int c, double u_sel, int lane, bool* anyfinite)
{
    const int k = p.k;
    double lp = -__builtin_huge_val();
    if (lane < k) lp = p.p_prior[c * k + lane] + chain_T(p, c) * p.p_like[c * k + lane];
    return mt_select_vals<true>(k, lp, u_sel, lane, anyfinite);
}

// log of the multi-try ratio (:305-323).  val: lane i < k holds proposal term A_i, lane mt_boff(k) + i reference term B_i,
// every other lane -inf.
// u_acc: the Metropolis uniform (:993); its logarithm is evaluated in lane 1 of the same dlog pass (one pass instead of two).
DZ_DEV int mt_boff(int k) { return k > 16 ? 32 : 16; }
template <bool BIG = false>
DZ_DEV double mt_log_ratio(int k, double val, double u_acc, int lane, double* log_u)
{
    const double rm = rowmax16(val);
    double m2 = fmax(readlane_f64(rm, 0), readlane_f64(rm, 16));                                       // :320
    if (BIG) m2 = fmax(m2, fmax(readlane_f64(rm, 32), readlane_f64(rm, 48)));
    const int boff = BIG ? mt_boff(k) : 16;
    const double ev = dexp(val - m2);                                                                // :321-322
    double SA = 0.0, SB = 0.0;
    for (int i = 0; i < k; ++i) SA = SA + readlane_f64(ev, i);
    for (int i = 0; i < k; ++i) SB = SB + readlane_f64(ev, boff + i);
    const double lg = dlog(lane == 1 ? u_acc : SA / SB);
    *log_u = readlane_f64(lg, 1);
    return nan_to_num(readlane_f64(lg, 0));                                                          // :323
}
template <bool BIG = false>
DZ_DEV double mt_log_ratio(int k, double val) { double lu; return mt_log_ratio<BIG>(k, val, 0.5, 0, &lu); }

// u53(hi, lo) < q for q in [0, 1], as an integer test: u53 = k 2^-53 with k = (hi >> 5) 2^26 + (lo >> 6), so the test is
// k < ceil(q 2^53) =: thr (made once on the host).  Wave-uniform operands stay on the scalar unit.
DZ_DEV bool u53_below(uint32_t hi, uint32_t lo, unsigned long long thr)
{
    const unsigned long long k = ((unsigned long long)(hi >> 5) << 26) | (unsigned long long)(lo >> 6);
    return k < thr;
}

DZ_DEV uint32_t mulhi_idx(uint32_t w, uint32_t M) { return (uint32_t)(((uint64_t)w * (uint64_t)M) >> 32); }

// slot layout of the precomputed uniform draws
DZ_DEV int pt_slot(const Params& p, int phase, int tr, int idx) { return 3 + ((phase ? p.k + tr : tr) * p.npt + idx); }
DZ_DEV u32x4 slot_counter_draw_key(const Params& p, int slot, uint32_t gc, uint32_t g, uint32_t k0, uint32_t k1)
{   // the Philox call a slot stands for
    if (slot < 3) return philox(k0, k1, (uint32_t)slot, stream_id(K_CTRL, 0, 0), gc, g);
    const int q = (slot - 3) / p.npt, idx = (slot - 3) % p.npt;
    const int phase = q >= p.k ? 1 : 0, tr = phase ? q - p.k : q;
    return philox(k0, k1, (uint32_t)idx, stream_id(K_PT, (uint32_t)tr, (uint32_t)phase), gc, g);
}
DZ_DEV u32x4 slot_counter_draw(const Params& p, int slot, uint32_t gc, uint32_t g) { return slot_counter_draw_key(p, slot, gc, g, p.k0, p.k1); }
// Where a wave gets its uniform draws from: lane s of the wave holds slot s of the chain's precomputed table
// (ONE coalesced 16-byte load per lane at wave start, then v_readlane), or nothing (evaluate Philox in place).
// xf: the point-stream slots have been turned, lane-parallel and once per generation, into what the tries read from them (persistent
// kernel, lean instantiations; dz_megakernel.h finish_draws): idx 0 -- .x = 1 if the gamma-unity draw says gamma = 1 (:615), .z/.w the raw
// words of the snooker gamma; idx 1 -- the archive ROW numbers themselves: DE .x/.y = random.sample(range(M), 2) (:662), snooker
// .x/.y/.z = z and the projected pair (:808-810).
// rekey: a redraw round of the persistent kernel (Dream.py:281-289) -- the point, dimension and boundary streams of the proposal set come
// from the Philox key (k0, k1) = seed + round * DZ_REDRAW_KEY_STEP instead of the run's (DESIGN.md section 4 "Redraw rounds").
// base (round 6, k_generations_d2 with more than 15 tries): lane s holds slot base + s -- a generation whose draw slots exceed a wave's 64 lanes
// keeps one phase's tries at a time (base = the phase's first slot); 0 everywhere else.
struct DrawSrc { uint4 mine; bool have; bool xf = false; bool rekey = false; uint32_t k0 = 0, k1 = 0; int base = 0; };
DZ_DEV DrawSrc load_draws(const Params& p, const uint4* dr, int lane)
{
    DrawSrc d; d.have = (dr != nullptr) && p.nslots <= 64; d.mine = make_uint4(0, 0, 0, 0);
    if (d.have && lane < p.nslots) d.mine = dr[lane];
    const unsigned zero = threadIdx.x >> 12;        // runtime 0: re-defines the registers by a VALU op, so later
    d.mine.x += zero; d.mine.y += zero; d.mine.z += zero; d.mine.w += zero;   // v_readlane never waits on the memory counter
    return d;
}
DZ_DEV u32x4 uniform_draw(const Params& p, const DrawSrc& d, int slot, uint32_t gc, uint32_t g)
{
    if (d.have) {
        const int sl = __builtin_amdgcn_readfirstlane(slot) - d.base;
        return u32x4{(uint32_t)__builtin_amdgcn_readlane((int)d.mine.x, sl), (uint32_t)__builtin_amdgcn_readlane((int)d.mine.y, sl),
                     (uint32_t)__builtin_amdgcn_readlane((int)d.mine.z, sl), (uint32_t)__builtin_amdgcn_readlane((int)d.mine.w, sl)};
    }
    return slot_counter_draw(p, slot, gc, g);
}
// uniform draw: from the precomputed table when available (dr != nullptr), else evaluated in place
DZ_DEV u32x4 uniform_draw(const Params& p, const uint4* dr, int slot, uint32_t gc, uint32_t g)
{
    if (dr) { const uint4 t = dr[slot]; return u32x4{t.x, t.y, t.z, t.w}; }
    return slot_counter_draw(p, slot, gc, g);
}
DZ_DEV Ctrl ctrl_from(const Params& p, const uint4* dr, uint32_t gc, uint32_t g)
{
    Ctrl c;
    u32x4 w = uniform_draw(p, dr, 0, gc, g); c.u_snk = u53(w.x, w.y); c.u_cr = u53(w.z, w.w);
    w = uniform_draw(p, dr, 1, gc, g); c.u_de = u53(w.x, w.y); c.u_glev = u53(w.z, w.w);
    w = uniform_draw(p, dr, 2, gc, g); c.u_sel = u53(w.x, w.y); c.u_acc = u53(w.z, w.w);
    return c;
}

// ------------------------------------------------------------------------------------------
// generate_proposal_points :670-796 (+ snooker_update :798-837, sample_from_history :646-668,
// set_gamma :601-626).  grid: ceil(nc*n/4) blocks of 256; one wave per (chain, try).
// ------------------------------------------------------------------------------------------
// number of 16-bit values h with (h + 1/2) 2^-16 < CR, so that `U_j < CR` (:704, :723) is the integer test
// h < crossover_threshold(CR) -- exactly the same predicate as the double comparison
__host__ __device__ inline uint32_t crossover_threshold(double CR)
{   // evaluated once per crossover value on the host (Params::crthr)
    const double t = CR * 65536.0 - 0.5;               // exact
    const double c = ceil(t);
    return c <= 0.0 ? 0u : (uint32_t)c;
}

template <int NCH>
DZ_DEV void load_row(const double* __restrict__ base, int ld, int lane, double (&xb)[NCH][2])
{
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        xb[it][0] = 0.0; xb[it][1] = 0.0;
        if (jj < ld) { const double2 t = *reinterpret_cast<const double2*>(base + jj); xb[it][0] = t.x; xb[it][1] = t.y; }
    }
}

// Z rows of one try.  DE: a = sum of the first delta rows, b = sum of
// the last delta rows (sample_from_history :646-668, chain_differences :692); snooker: a = z, b, c = the
// projected pair (:808-810).
template <int NCH> struct ZRows { double a[NCH][2], b[NCH][2], c[NCH][2]; };

template <int NCH>
DZ_DEV void fetch_rows(const Params& p, int phase, uint32_t g, uint32_t M, uint32_t gc, int i, int lane, bool snk, int delta,
                       const DrawSrc& dr, ZRows<NCH>& zr)
{
    const int ld = p.ld;
    if (!snk) {
        uint32_t rows[2 * MAXPAIR];
        {   // random.sample(range(M), 2*delta) :662
            uint32_t sorted[2 * MAXPAIR];
            const int nidx = 2 * delta;
            u32x4 w = uniform_draw(p, dr, pt_slot(p, phase, i, 1), gc, g);
            for (int t = 0; t < nidx; ++t) {
                if (t && (t & 3) == 0) w = uniform_draw(p, dr, pt_slot(p, phase, i, 1 + (t >> 2)), gc, g);
                const uint32_t word = (t & 3) == 0 ? w.x : (t & 3) == 1 ? w.y : (t & 3) == 2 ? w.z : w.w;
                uint32_t r = mulhi_idx(word, M - (uint32_t)t);
                int pos = 0;
                for (int s = 0; s < t; ++s) { if (r >= sorted[s]) { r++; pos = s + 1; } else break; }
                for (int s = t; s > pos; --s) sorted[s] = sorted[s - 1];
                sorted[pos] = r; rows[t] = r;
            }
        }
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int jj = 128 * it + 2 * lane;
            double2 a = {0.0, 0.0}, b = {0.0, 0.0};
            if (jj < ld) {
                if (delta == 1) {       // common case kept free of arithmetic on the loaded values: the loads stay in
                    a = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[0] * ld + jj);      // flight until the
                    b = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[1] * ld + jj);      // consumer needs them
                } else {
                    a = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[0] * ld + jj);
                    b = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[delta] * ld + jj);
                    for (int t = 1; t < delta; ++t) {
                        const double2 a2 = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[t] * ld + jj);
                        const double2 b2 = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[delta + t] * ld + jj);
                        a.x = a.x + a2.x; a.y = a.y + a2.y; b.x = b.x + b2.x; b.y = b.y + b2.y;
                    }
                }
            }
            zr.a[it][0] = a.x; zr.a[it][1] = a.y; zr.b[it][0] = b.x; zr.b[it][1] = b.y; zr.c[it][0] = 0.0; zr.c[it][1] = 0.0;
        }
    } else {
        const u32x4 wi = uniform_draw(p, dr, pt_slot(p, phase, i, 1), gc, g);  // :808-810
        const uint32_t iz = mulhi_idx(wi.x, M), i1 = mulhi_idx(wi.y, M), i2 = mulhi_idx(wi.z, M);
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int jj = 128 * it + 2 * lane;
            double2 z = {0.0, 0.0}, r1 = {0.0, 0.0}, r2 = {0.0, 0.0};
            if (jj < ld) {
                z = *reinterpret_cast<const double2*>(p.Z + (size_t)iz * ld + jj);
                r1 = *reinterpret_cast<const double2*>(p.Z + (size_t)i1 * ld + jj);
                r2 = *reinterpret_cast<const double2*>(p.Z + (size_t)i2 * ld + jj);
            }
            zr.a[it][0] = z.x; zr.a[it][1] = z.y; zr.b[it][0] = r1.x; zr.b[it][1] = r1.y; zr.c[it][0] = r2.x; zr.c[it][1] = r2.y;
        }
    }
}

// gamma_arr[level-1][delta-1][:] (Dream.py:172-179): the row the look-up gamma_arr[..][d'-1] (:624) reads from
DZ_DEV const double* gamma_row(const Params& p, int glev, int delta)
{
    return p.gtab + ((size_t)(__builtin_amdgcn_readfirstlane(glev) - 1) * p.depairs + (__builtin_amdgcn_readfirstlane(delta) - 1)) * p.d;
}

// What propose_point needs from the fetched rows, reduced to VALU-defined registers as soon as the loads land
// (so that later instructions never wait on the memory counter for them): DE: a = Z_a - Z_b (:692);
// snooker: a = z, b = zR1 - zR2 (:819).
template <int NCH> struct RowTerms { double a[NCH][2], b[NCH][2]; };
template <int NCH>
DZ_DEV void reduce_rows(const ZRows<NCH>& zr, bool snk, RowTerms<NCH>& rt)
{
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            rt.a[it][s] = snk ? zr.a[it][s] : zr.a[it][s] - zr.b[it][s];
            rt.b[it][s] = snk ? zr.b[it][s] - zr.c[it][s] : 0.0;
        }
}

// LEAN drops what the persistent kernel's fast path never needs (hard-boundary handling, the single-try snooker
// variant) so that the whole generation fits the register budget of a 16-wave block.
// What a set of tries reads from Params, fetched ONCE per set by the persistent kernel (Params lives in memory there: every p.x inside
// the try loop is a scalar load plus a wait, and scalar instructions cost nearly as much issue time as vector ones -- measured: 200 extra
// s_add per generation = -3 %).  slot(i, idx) = slot0 + i npt + idx is pt_slot() of the set's phase.
struct SetConsts { uint32_t thr; unsigned long long pgu_thr; double zeta, ec1, ec0; int npt, slot0; };
// Per-dimension constants of the priors (SampledParam: kind, loc, scale, log scale) and the hard boundaries, staged in LDS by the
// persistent kernel's full-code instantiations ([ld] each): a try then costs LDS reads instead of eight dependent global loads.
struct PBConsts { const double *a, *b, *logb, *lo, *hi; const int* kind; const double* inside; };      // (b: Params::pc2; inside: the log prior of a point inside every support, one LDS word)
DZ_DEV SetConsts set_consts(const Params& p, int phase, int cr_idx)
{
    SetConsts s;
    s.thr = p.crthr[__builtin_amdgcn_readfirstlane(cr_idx)]; s.pgu_thr = p.pgu_thr; s.zeta = p.zeta; s.ec1 = p.ec1; s.ec0 = p.ec0;
    s.npt = p.npt; s.slot0 = 3 + (phase ? p.k : 0) * p.npt;
    return s;
}

// Returns, for a snooker try, the squared distance |proposal - z|^2 (wave-uniform): the caller turns the tries' distances into
// snooker_logp = (d - 1) log sqrt(.) (:823-824 / :834-835) in ONE pass of the logarithm (snooker_logps below) instead of one per
// try; 0 for a DE try (snooker_logp = 0).  grow: gamma_arr[level-1][delta-1][:] (any address space; the look-up address is
// wave-uniform).
template <int NCH, bool AL16 = true, int LEAN = 0>      // LEAN: 0 full, 1 lean (multi-try sets only), 2 lean with the single-try snooker formula
DZ_DEV double propose_point(const Params& p, int phase, uint32_t g, uint32_t M, int c, int i, int n, int lane,
                          const double (&xb)[NCH][2], const double* __restrict__ grow, const RowTerms<NCH>& zr, double* __restrict__ out,
                          double* cur_snk_out, bool snk, int cr_idx, int delta, int glev, const DrawSrc& dr, const u32x4* wpre = nullptr,
                          const SetConsts* sc = nullptr, const PBConsts* pc = nullptr, double (*prv)[2] = nullptr)
{   // wpre: the DIM draw of chunk 0, computed by the caller one try ahead (software pipelining, NCH == 1)
    // pc: bounds from LDS; prv: receives the lane's proposal values as stored (for the prior evaluation: no read-back of the row)
    const int d = p.d, ld = p.ld;
    const uint32_t gc = (uint32_t)(p.off + c);
    const uint32_t thr = sc ? sc->thr : p.crthr[__builtin_amdgcn_readfirstlane(cr_idx)];   // CR = CR_values[m], :146 (the decision is wave-uniform)
    const uint32_t s_dim = stream_id(K_DIM, (uint32_t)i, (uint32_t)phase),
                   s_bnd = stream_id(K_BND, (uint32_t)i, (uint32_t)phase);
    const uint32_t K0 = dr.rekey ? dr.k0 : p.k0, K1 = dr.rekey ? dr.k1 : p.k1;      // (a redraw round's key)
    double pr[NCH][2];
    double sqdist = 0.0;
    if (!snk) {
        bool keep[NCH][2]; double e1[NCH][2], zt[NCH][2];
        const double ec1 = sc ? sc->ec1 : p.ec1, ec0 = sc ? sc->ec0 : p.ec0, zeta = sc ? sc->zeta : p.zeta;
        int dprime = 0;
#pragma unroll
        for (int it = 0; it < NCH; ++it) {        // zeta, e, U :694-700 -- one Philox call and one Box-Muller pair per lane
            const int j0 = 128 * it + 2 * lane;    // the lane's two dimensions j0, j0+1 = pair j0/2
            {   // no lane predicate around the arithmetic (the lanes past d compute on their own, unused draws; `keep` masks them):
                // a predicated region costs the zero defaults of every value it defines plus the exec-mask round trip
                const u32x4 w = (NCH == 1 && wpre) ? *wpre : philox(K0, K1, (uint32_t)(j0 >> 1), s_dim, gc, g);
                float z0, z1;
                normal32_pair(w.z, w.w, z0, z1);
                keep[it][0] = j0 < d && (w.x & 0xffffu) < thr;               // U_j < CR
                e1[it][0] = uniform16(w.y, ec1, ec0) + 1.0;                  // :696-697
                zt[it][0] = zeta * (double)z0;
                keep[it][1] = j0 + 1 < d && (w.x >> 16) < thr;
                e1[it][1] = uniform16(w.y >> 16, ec1, ec0) + 1.0;
                zt[it][1] = zeta * (double)z1;
                // d' :704 / :709 -- ballots of the two plain comparisons, masked by the (loop-invariant) ballots of the lanes that own a
                // dimension: a ballot of the conjunction is materialised as 0 / 1 and compared again (two extra vector instructions each)
                dprime += __popcll(__builtin_amdgcn_ballot_w64((w.x & 0xffffu) < thr) & __builtin_amdgcn_ballot_w64(j0 < d))
                        + __popcll(__builtin_amdgcn_ballot_w64((w.x >> 16) < thr) & __builtin_amdgcn_ballot_w64(j0 + 1 < d));
            }
        }
        const u32x4 wg = uniform_draw(p, dr, sc ? sc->slot0 + i * sc->npt : pt_slot(p, phase, i, 0), gc, g);  // set_gamma :615
        double gamma = 1.0;
        if (dr.xf ? (wg.x == 0u) : !u53_below(wg.x, wg.y, sc ? sc->pgu_thr : p.pgu_thr))      // u53(wg.x, wg.y) < p_gamma_unity
            gamma = grow[(dprime == 0 ? d : dprime) - 1];                      // gamma_arr[level-1][delta-1][d'-1], :624 (one address for the whole wave)
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) {     // :714 / :717, crossover :720-726
                double t = e1[it][s] * gamma; t = t * zr.a[it][s];                          // chain_differences :692
                double q = xb[it][s] + t; q = q + zt[it][s];
                pr[it][s] = keep[it][s] ? q : xb[it][s];
            }
    } else {
        const u32x4 wg = uniform_draw(p, dr, pt_slot(p, phase, 0, 0), gc, g);
        const double gamma_s = 1.2 + (2.2 - 1.2) * u53(wg.z, wg.w);            // :618
        double v[NCH][2], dzz[NCH][2], zz[NCH][2];
        double accD = 0.0, accS = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int jj = 128 * it + 2 * lane;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                zz[it][s] = zr.a[it][s];
                v[it][s] = xb[it][s] - zr.a[it][s];                            // :813
                dzz[it][s] = zr.b[it][s];                                      // :819
                if (jj + s < d) { accD = fma(v[it][s], v[it][s], accD); accS = fma(dzz[it][s], v[it][s], accS); }
            }
        }
        const double DS = wave_bfly2(accD, accS);                              // (both sums for the price of one: D in the lower lanes, S in the upper)
        const double D = readlane_f64(DS, 0);                                  // :816 / :827
        if (LEAN == 1 || n > 1) {
            const double cc = readlane_f64(DS, 32) / D;                        // :820
#pragma unroll
            for (int it = 0; it < NCH; ++it)
#pragma unroll
                for (int s = 0; s < 2; ++s) pr[it][s] = xb[it][s] + gamma_s * nan_to_num(cc * v[it][s]);   // :820-822
        } else {
            double sp = 0.0;
            if (D != 0.0) {
                double acc = 0.0;
#pragma unroll
                for (int it = 0; it < NCH; ++it)
#pragma unroll
                    for (int s = 0; s < 2; ++s) if (128 * it + 2 * lane + s < d) acc = acc + (dzz[it][s] * v[it][s]) / D;   // :831
                sp = wave_bfly(acc);
            }
            sp = nan_to_num(sp);
#pragma unroll
            for (int it = 0; it < NCH; ++it)
#pragma unroll
                for (int s = 0; s < 2; ++s) { const double zp = sp * v[it][s]; pr[it][s] = xb[it][s] + gamma_s * zp; }     // :832-833
        }
        double accN = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) if (128 * it + 2 * lane + s < d) { const double t = pr[it][s] - zz[it][s]; accN = fma(t, t, accN); }
        sqdist = wave_bfly(accN);                                              // :823 / :834 (the norm's square; snooker_logps finishes it)
        if (cur_snk_out && i == 0) { const double nc = sqrt(D); *cur_snk_out = nc != 0.0 ? dlog(nc) * (double)(d - 1) : 0.0; }   // :328-329
    }
    // hard boundaries :733-791
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        if (jj < ld) {
            if (!LEAN && p.hard) {
                const double2 mn = *reinterpret_cast<const double2*>((pc ? pc->lo : p.mins) + jj);
                const double2 mxx = *reinterpret_cast<const double2*>((pc ? pc->hi : p.maxs) + jj);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const double lo = s ? mn.y : mn.x, hi = s ? mxx.y : mxx.x;
                    double x = pr[it][s];
                    const bool bl = x < lo, bh = x > hi;
                    if (bl) x = 2 * lo - x;
                    if (bh) x = 2 * hi - x;
                    if (x < lo || x > hi) {
                        const u32x4 w = philox(K0, K1, (uint32_t)(jj + s), s_bnd, gc, g);
                        x = lo + u32d(w.x) * (hi - lo);
                    }
                    pr[it][s] = x;
                }
            }
            double2 o; o.x = (jj < d) ? pr[it][0] : 0.0; o.y = (jj + 1 < d) ? pr[it][1] : 0.0;
            if (prv) { prv[it][0] = o.x; prv[it][1] = o.y; }
            if (AL16) *reinterpret_cast<double2*>(out + jj) = o;
            // 8-byte aligned row of an LDS tile, zero padded to the k-steps (4 ceil(d / 4) + 1 columns: with d odd the lane that owns the
            // last dimension also rewrites the first pad column with its zero): ONE predicated region, two adjacent stores
            else if (jj < d) { out[jj] = o.x; out[jj + 1] = o.y; }
        }
    }
    return sqdist;
}

// snooker_logp of the tries of one set (:823-824 / :834-835): lane j holds the squared distance of try j (propose_point's return
// value), j < cnt; sl[j] = (d - 1) log sqrt(.), or 0 where the distance is 0 (np.log(..., where = norm != 0)).  One
// pass of sqrt / dlog for all tries.
DZ_DEV void snooker_logps(const Params& p, double sq, int cnt, int lane, double* __restrict__ sl)
{
    const double norm = sqrt(lane < cnt ? sq : 1.0);
    const double v = norm != 0.0 ? dlog(norm) * (double)(p.d - 1) : 0.0;
    if (lane < cnt) sl[lane] = v;
}

// sum over dims of the per-dimension prior log densities (parameters.py:37-47), lane/butterfly order
template <int NCH>
DZ_DEV double prior_of_point(const Params& p, const double (&x)[NCH][2], int lane)
{
    if (!p.have_prior) return 0.0;
    double acc = 0.0;
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = 128 * it + 2 * lane + s;
            if (j < p.d) {
                const int kd = p.pkind[j];
                double t = 0.0;
                if (kd == 1) { const double z = (x[it][s] - p.pa[j]) * p.pc2[j]; t = (-(z * z) / 2.0 - 0.91893853320467274178) - p.plogb[j]; }
                else if (kd == 2) t = (x[it][s] >= p.pa[j] && x[it][s] <= p.pc2[j]) ? -p.plogb[j] : -__builtin_huge_val();
                acc = acc + t;
            }
        }
    return wave_bfly(acc);
}
DZ_DEV double nan_to_ninf(double x) { return x != x ? -__builtin_huge_val() : x; }
// ... the same with the constants from LDS (PBConsts; b = Params::pc2) and the point in registers (NCH == 1: the persistent kernels).
// prior_lane_lds: the lane's partial sum (its two dimensions in order); the butterfly over the lanes finishes it.
DZ_DEV double prior_lane_lds(const Params& p, const PBConsts& pc, const double (&x)[1][2], int lane)
{
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = 2 * lane + s;
        if (j < p.d) {
            const int kd = pc.kind[j];
            const double a = pc.a[j], c2 = pc.b[j], lb = pc.logb[j];
            double t = 0.0;
            if (kd == 1) { const double z = (x[0][s] - a) * c2; t = (-(z * z) / 2.0 - 0.91893853320467274178) - lb; }
            else if (kd == 2) t = (x[0][s] >= a && x[0][s] <= c2) ? -lb : -__builtin_huge_val();
            acc = acc + t;
        }
    }
    return acc;
}
DZ_DEV double prior_of_point_lds(const Params& p, const PBConsts& pc, const double (&x)[1][2], int lane) { return wave_bfly(prior_lane_lds(p, pc, x, lane)); }
// No normal prior anywhere (Params::prior_nonormal): a point's log prior is -inf if any uniform dimension is outside its support,
// else the constant `inside_value` = the same sum with every point inside (made once per launch by the same code) -- a ballot
// instead of a butterfly per try.
DZ_DEV double prior_of_point_nonormal(const Params& p, const PBConsts& pc, const double (&x)[1][2], int lane, double inside_value)
{
    bool out = false;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int j = 2 * lane + s;
        if (j < p.d) out = out || (pc.kind[j] == 2 && !(x[0][s] >= pc.a[j] && x[0][s] <= pc.b[j]));
    }
    return __any(out) ? -__builtin_huge_val() : inside_value;
}
// the log prior of a try of the persistent kernels' full-code instantiations (wave-uniform); pc.inside: see prior_of_point_nonormal
DZ_DEV double prior_try_lds(const Params& p, const PBConsts& pc, const double (&x)[1][2], int lane)
{
    if (p.prior_const) return *pc.inside;
    if (p.prior_nonormal) return prior_of_point_nonormal(p, pc, x, lane, *pc.inside);
    return nan_to_ninf(prior_of_point_lds(p, pc, x, lane));
}
// Q = q_0 + q_1 + ... in ascending row tile t (the MVN contract, v2) for point `pt` of the scratch array [row tile][npts] the tiled
// likelihood kernels leave -- what k_q_finish does, for the kernels that take the sums over themselves (Params::qfin_*).  32 loads in
// flight per round trip (index clamped, the add skipped beyond nrt).
DZ_DEV double q_tile_sum(const double* __restrict__ q, size_t npts, int nrt, size_t pt)
{
    double Q = 0.0;
    for (int t0 = 0; t0 < nrt; t0 += 32) {
        double v[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) v[j] = q[(size_t)min(t0 + j, nrt - 1) * npts + pt];
#pragma unroll
        for (int j = 0; j < 32; ++j) if (t0 + j < nrt) Q = Q + v[j];
    }
    return Q;
}
DZ_DEV double like_from_q(const Params& p, const double* __restrict__ q, size_t npts, size_t pt)
{
    return nan_to_ninf(p.logF - 0.5 * q_tile_sum(q, npts, p.qfin_nrt, pt));
}

// The transition of chain c at generation g (one wave).  Leaves the new state in xn, and -- when prep_next --
// the wave-uniform draws and control decisions of generation g+1 in registers (dnext: lane s holds slot s;
// cnext) as well as in memory (draws_out / ctl_out) for the kernels that follow.
template <int NCH>
DZ_DEV void accept_chain(const Params& p, uint32_t g, int64_t zbase, int c, int lane, int64_t trace_slot, int append, int publish, int prep_next,
                         const ChainCtl* ctl_cur, ChainCtl* ctl_out, uint4* draws_out,
                         double (&xn)[NCH][2], DrawSrc& dnext, ChainCtl& cnext)
{
    const int k = p.k, ld = p.ld;
    const uint32_t gc = (uint32_t)(p.off + c);
    const ChainCtl ct = ctl_cur[c];
    StepFlags f; f.snk = ct.snk != 0; f.cr_idx = ct.cr_idx; f.delta = ct.delta; f.glev = ct.glev;
    Ctrl u; u.u_sel = ct.u_sel; u.u_acc = ct.u_acc;
    dnext.have = false; dnext.mine = make_uint4(0, 0, 0, 0);
    cnext = ct;
    if (prep_next) {   // lane-parallel: the wave-uniform Philox outputs and control decisions of generation g+1
        u32x4 w0 = u32x4{0, 0, 0, 0};
        for (int slot = lane; slot < p.nslots; slot += 64) {
            const u32x4 w = slot_counter_draw(p, slot, gc, g + 1);
            if (slot == lane) w0 = w;
            draws_out[(size_t)c * p.nslots + slot] = make_uint4(w.x, w.y, w.z, w.w);
        }
        Ctrl un;
        un.u_snk = u53(__shfl(w0.x, 0, 64), __shfl(w0.y, 0, 64)); un.u_cr = u53(__shfl(w0.z, 0, 64), __shfl(w0.w, 0, 64));
        un.u_de = u53(__shfl(w0.x, 1, 64), __shfl(w0.y, 1, 64)); un.u_glev = u53(__shfl(w0.z, 1, 64), __shfl(w0.w, 1, 64));
        un.u_sel = u53(__shfl(w0.x, 2, 64), __shfl(w0.y, 2, 64)); un.u_acc = u53(__shfl(w0.z, 2, 64), __shfl(w0.w, 2, 64));
        const StepFlags fn = step_flags(p, un);
        cnext.snk = fn.snk ? 1 : 0; cnext.cr_idx = fn.cr_idx; cnext.delta = fn.delta; cnext.glev = fn.glev; cnext.u_sel = un.u_sel; cnext.u_acc = un.u_acc;
        if (lane == 0) ctl_out[c] = cnext;
        dnext.have = p.nslots <= 64; dnext.mine = make_uint4(w0.x, w0.y, w0.z, w0.w);
    }
    const double last_prior = p.lprior[c], last_like = p.llike[c];
    const double T = chain_T(p, c);
    const double last_logp = T * last_like + last_prior;                       // :243, :268
    double ratio; int sel = 0;
    if (k == 1) {
        const double q_logp = T * p.p_like[c] + p.p_prior[c];                  // :274
        if (f.snk) ratio = nan_to_num((q_logp + p.p_slogp[c]) - (last_logp + p.cur_snk[c]));   // :326-332
        else ratio = nan_to_num(q_logp) - nan_to_num(last_logp);               // :334
    } else {
        const int sf = p.sel[c]; sel = sf & 255; const bool fin = (sf & 256) != 0;      // mt_choose_proposal_pt result of this chain (:291)
        // lane i < k holds proposal term A_i, lane 16+i holds reference term B_i (:306-317)
        double val = -__builtin_huge_val();
        if (lane < k) {
            val = p.p_prior[c * k + lane] + T * p.p_like[c * k + lane];                                    // :279
            if (f.snk) val = val + p.p_slogp[c * k + lane];                                              // :307
        } else if (lane >= mt_boff(k) && lane < mt_boff(k) + k) {
            const int i = lane - mt_boff(k);
            if (i < k - 1) {
                double rl, rp;
                if (p.qfin_r) {      // the reference set's row-tile sums are still in the scratch array: finish them here
                    rl = like_from_q(p, p.qfin_r, (size_t)p.qfin_nc * (k - 1), (size_t)(c - p.qfin_c0) * (k - 1) + i);
                    rp = p.have_prior ? p.r_prior[c * (k - 1) + i] : 0.0;
                    p.r_like[c * (k - 1) + i] = rl;
                    if (!p.have_prior) p.r_prior[c * (k - 1) + i] = 0.0;
                } else { rl = p.r_like[c * (k - 1) + i]; rp = p.r_prior[c * (k - 1) + i]; }
                val = T * rl + rp;                                                                       // :303
            } else val = T * last_like + last_prior;                                                     // :877-879
            if (f.snk) { const double sr = i < k - 1 ? p.r_slogp[c * (k - 1) + i] : 0.0; val = (val + sr) + p.p_slogp[c * k + i]; }   // :312-313
        }
        ratio = mt_log_ratio<true>(k, val);
        if (!fin) ratio = -__builtin_huge_val();                               // DESIGN.md deviation D1 (:282-289)
    }
    const bool accept = is_finite(ratio) && (dlog(u.u_acc) < ratio);           // :993
    const double* src = p.P + ((size_t)c * k + sel) * ld;
    double* xrow = p.X + (size_t)c * ld;
    bool diff = false;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        xn[it][0] = 0.0; xn[it][1] = 0.0;
        if (jj < ld) {
            const double2 xo = *reinterpret_cast<const double2*>(xrow + jj);
            double2 t = xo;
            if (accept) { t = *reinterpret_cast<const double2*>(src + jj); diff = diff || (t.x != xo.x) || (t.y != xo.y); }
            xn[it][0] = t.x; xn[it][1] = t.y;
        }
    }
    const bool moved = __any(diff);                                            // core.py:120
    const double npri = accept ? p.p_prior[c * k + sel] : last_prior;          // :345-347
    const double nlik = accept ? p.p_like[c * k + sel] : last_like;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        if (jj < ld) {
            const double2 t = {xn[it][0], xn[it][1]};
            if (accept) *reinterpret_cast<double2*>(xrow + jj) = t;
            if (trace_slot >= 0) *reinterpret_cast<double2*>(p.tX + ((size_t)c * p.tcap + trace_slot) * ld + jj) = t;
            if (append) *reinterpret_cast<double2*>(p.Z + (size_t)(zbase + (int64_t)gc) * ld + jj) = t;   // :933-936
            if (publish) *reinterpret_cast<double2*>(p.cp_new + (size_t)gc * ld + jj) = t;       // :447-449
        }
    }
    if (lane == 0) {
        p.lprior[c] = npri; p.llike[c] = nlik;
        if (trace_slot >= 0) {
            const size_t o = (size_t)trace_slot * p.nl + c;
            p.tlogp[o] = T * nlik + npri;                                      // core.py:115; with a temperature ladder core.py:178 (1.0 * x == x)
            p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = f.cr_idx; p.tsnk[o] = f.snk ? 1 : 0;
        }
    }
}


// prior of a point that was just written to `row` (lane l reads back the dimensions it wrote)
template <int NCH>
DZ_DEV void point_prior(const Params& p, const double* row, int lane, double* prior_out)
{
    double pr = 0.0;
    if (p.have_prior) {
        double pt[NCH][2];
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { const int j = 128 * it + 2 * lane + s2; pt[it][s2] = j < p.d ? row[j] : 0.0; }
        pr = nan_to_ninf(prior_of_point<NCH>(p, pt, lane));
    }
    if (lane == 0) *prior_out = pr;
}

// Tries i0..i1-1 of one chain's proposal set (phase 0: around the current state; phase 1: the reference set
// around the selected proposal).  out / sl / prior_out address try 0's row and scalars.
template <int NCH, bool AL16, bool GENERIC = true, int LEAN = 0>
DZ_DEV void propose_set(const Params& p, int phase, uint32_t g, uint32_t M, int c, uint32_t gc, int i0, int i1, int n, int lane,
                        const double (&xb)[NCH][2], const double* __restrict__ grow, bool snk, int cr_idx, int delta, int glev, const DrawSrc& dsrc,
                        double* out, int out_stride, double* sl, double* csn, double* prior_out, const PBConsts* pc = nullptr)
{
    // pc (persistent kernel, NCH == 1): boundary / prior constants in LDS, the prior evaluated on the values the lane has just stored
    auto point_and_prior = [&](int i, const RowTerms<NCH>& rt, bool snk_, int delta_) -> double {
        if (NCH == 1 && !LEAN && pc) {
            double pv[NCH][2];
            const double sq = propose_point<NCH, AL16, LEAN>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * out_stride, csn, snk_, cr_idx, delta_, glev, dsrc, nullptr, nullptr, pc, pv);
            if (prior_out) {
                double pr = 0.0;
                if (p.have_prior) { const double (&pv1)[1][2] = reinterpret_cast<const double (&)[1][2]>(pv); pr = prior_try_lds(p, *pc, pv1, lane); }
                if (lane == 0) prior_out[i] = pr;
            }
            return sq;
        }
        const double sq = propose_point<NCH, AL16, LEAN>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * out_stride, csn, snk_, cr_idx, delta_, glev, dsrc);
        if (prior_out) { if (LEAN) { if (lane == 0) prior_out[i] = 0.0; } else point_prior<NCH>(p, out + (size_t)i * out_stride, lane, prior_out + i); }
        return sq;
    };
    // software pipeline over the tries: the Z rows of try i+1 are requested before try i's arithmetic starts,
    // and no scalar memory wait sits in between because the draws are already in registers
    if (!snk && delta == 1) {
        // Common case (DE move, one pair): straight-line software pipeline.  The two Z rows of try i+1 are
        // requested right after try i's difference has been formed, and stay in flight during try i's
        // random numbers and arithmetic.  (Kept free of other control flow on purpose: any merge of paths that
        // define the row registers makes the compiler wait for the loads at the merge.)
        double2 ra[NCH], rb[NCH];
        auto request = [&](int i) {
            const u32x4 w = uniform_draw(p, dsrc, pt_slot(p, phase, i, 1), gc, g);
            const uint32_t r0 = __builtin_amdgcn_readfirstlane(mulhi_idx(w.x, M));
            uint32_t r1 = __builtin_amdgcn_readfirstlane(mulhi_idx(w.y, M - 1u));
            if (r1 >= r0) r1++;                                   // random.sample(range(M), 2) :662  (wave-uniform: kept on the scalar unit)
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                const int jj = 128 * it + 2 * lane;
                // no lane predicate (it would keep the previous rows alive in the lanes past ld and cost a register copy per row
                // and try): those lanes re-read the row's last pair, their results are masked where they are used (j < d)
                const int jc = min(jj, p.ld - 2);
                ra[it] = gload2(p.Z + (size_t)r0 * p.ld + jc);
                rb[it] = gload2(p.Z + (size_t)r1 * p.ld + jc);
            }
        };
#pragma unroll
        for (int it = 0; it < NCH; ++it) { ra[it] = double2{0.0, 0.0}; rb[it] = double2{0.0, 0.0}; }
        request(i0);
        // the Philox call of try i+1 is issued before try i's Box-Muller and proposal arithmetic: two independent
        // dependency chains for the scheduler to interleave (a wave is latency-bound at 4 waves per SIMD)
        auto dimdraw = [&](int i) { return philox(p.k0, p.k1, (uint32_t)lane, stream_id(K_DIM, (uint32_t)i, (uint32_t)phase), gc, g); };
        // (not in the persistent kernel: LEAN -- its 128-register budget has no room for the extra draw; measured -4% there, +1% here)
        constexpr bool AHEAD = NCH == 1 && !LEAN && AL16;     // (AL16: k_propose's global rows; the persistent kernel writes LDS rows)
        u32x4 wn = AHEAD ? dimdraw(i0) : u32x4{0, 0, 0, 0};
        for (int i = i0; i < i1; ++i) {
            const u32x4 wcur = wn;
            if (AHEAD) wn = dimdraw(min(i + 1, i1 - 1));
            RowTerms<NCH> rt;
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                rt.a[it][0] = ra[it].x - rb[it].x; rt.a[it][1] = ra[it].y - rb[it].y;   // chain_differences :692
                rt.b[it][0] = 0.0; rt.b[it][1] = 0.0;
            }
            // (a deeper pipeline -- three buffers, straight-line code, rows of tries i+1..i+3 in flight -- measured
            //  10% SLOWER: during the tries the kernel already moves ~4 TB/s, so latency is not what limits it)
            DZ_STAMP(p, phase, c, 2 + 2 * i);          // rows of try i have arrived
            // (unconditional: after the last try the same rows are requested again and never used -- a conditional request makes
            //  the compiler keep the old and the new rows apart and copy one set into the other every try)
            request(min(i + 1, i1 - 1));
            if (NCH == 1 && !LEAN && pc) {      // constants through PBConsts: the prior on the values in registers (no read-back of the row)
                double pv[NCH][2];
                propose_point<NCH, AL16, LEAN>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * out_stride, csn, false, cr_idx, 1, glev, dsrc,
                                               AHEAD ? &wcur : nullptr, nullptr, pc, pv);
                if (lane == 0) sl[i] = 0.0;
                if (prior_out) {
                    double pr = 0.0;
                    if (p.have_prior) { const double (&pv1)[1][2] = reinterpret_cast<const double (&)[1][2]>(pv); pr = prior_try_lds(p, *pc, pv1, lane); }
                    if (lane == 0) prior_out[i] = pr;
                }
                continue;
            }
            propose_point<NCH, AL16, LEAN>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * out_stride, csn, false, cr_idx, 1, glev, dsrc,
                                           AHEAD ? &wcur : nullptr);
            if (lane == 0) sl[i] = 0.0;
            DZ_STAMP(p, phase, c, 3 + 2 * i);          // try i's arithmetic issued
            if (prior_out) { if (LEAN) { if (lane == 0) prior_out[i] = 0.0; } else point_prior<NCH>(p, out + (size_t)i * out_stride, lane, prior_out + i); }   // LEAN: flat priors only
        }
        return;
    }
    if (snk) {
        // snooker move: same pipeline with three rows per try (z and the projected pair, :808-810)
        double2 rz[NCH], r1[NCH], r2[NCH];
        auto request = [&](int i) {
            const u32x4 w = uniform_draw(p, dsrc, pt_slot(p, phase, i, 1), gc, g);
            const uint32_t iz = dsrc.xf ? w.x : mulhi_idx(w.x, M), i1x = dsrc.xf ? w.y : mulhi_idx(w.y, M), i2x = dsrc.xf ? w.z : mulhi_idx(w.z, M);
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                const int jc = min(128 * it + 2 * lane, p.ld - 2);            // (no lane predicate: see the DE path)
                rz[it] = gload2(p.Z + (size_t)iz * p.ld + jc);
                r1[it] = gload2(p.Z + (size_t)i1x * p.ld + jc);
                r2[it] = gload2(p.Z + (size_t)i2x * p.ld + jc);
            }
        };
#pragma unroll
        for (int it = 0; it < NCH; ++it) { rz[it] = double2{0.0, 0.0}; r1[it] = rz[it]; r2[it] = rz[it]; }
        double sqv = 1.0;                             // lane j: squared distance |proposal - z|^2 of try i0 + j
        request(i0);
        for (int i = i0; i < i1; ++i) {
            RowTerms<NCH> rt;
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                rt.a[it][0] = rz[it].x; rt.a[it][1] = rz[it].y;
                rt.b[it][0] = r1[it].x - r2[it].x; rt.b[it][1] = r1[it].y - r2[it].y;          // :819
            }
            if (i + 1 < i1) request(i + 1);
            const double sq = point_and_prior(i, rt, true, delta);
            sqv = (lane == i - i0) ? sq : sqv;
        }
        snooker_logps(p, sqv, i1 - i0, lane, sl + i0);
        return;
    }
    if (GENERIC) for (int i = i0; i < i1; ++i) {     // DEpairs > 1
        ZRows<NCH> raw;
        RowTerms<NCH> rt;
        fetch_rows<NCH>(p, phase, g, M, gc, i, lane, false, delta, dsrc, raw);
        reduce_rows<NCH>(raw, false, rt);
        point_and_prior(i, rt, false, delta);
        if (lane == 0) sl[i] = 0.0;
    }
}

// split = 1: one wave per CHAIN (control decisions, crossover threshold and base row fetched once, the wave
// loops over the tries); split = n: one wave per (chain, try).  The host picks by problem size: per-chain
// waves do ~35% fewer instructions, per-try waves expose 5x more parallelism.
template <int NCH>
// (blocks of 16 waves cap a wave at 128 registers; from four 128-dimension chunks per lane on that spilled hundreds of them --
//  1093 at NCH = 8 --, so those variants run in 4-wave blocks with the full register file per wave)
__global__ __launch_bounds__(NCH >= 4 ? 256 : 1024) void k_propose(Params p, int phase, uint32_t g, uint32_t M, int c0, int nc, int split, int fuse_accept, int64_t fuse_slot)
{
    const int n = phase == 0 ? p.k : p.k - 1;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (wave >= nc * split) return;
    const int lane = threadIdx.x & 63;
    const int c = p.redo_list ? p.redo_list[wave / split] : c0 + wave / split;      // (redraw round: only the chains whose tries were all impossible)
    if (p.redo_list && !p.redo[c]) return;                                          // (... and still are: several rounds are queued per read-back)
    DZ_STAMP(p, phase, c, 0);
    const int per = (n + split - 1) / split;
    const int i0 = (wave % split) * per, i1 = min(n, i0 + per);
    DrawSrc dsrc; ChainCtl ct;
    double xb[NCH][2];
    double* out; double* sl;
    if (fuse_accept) {
        // phase 0 of generation g with the Metropolis step of generation g-1 in front (same wave, same chain):
        // the new state, the draws and the control decisions of generation g never leave the registers.
        // Params are those of generation g: generation g-1's decisions sit in ctl_next, generation g's go to ctl/draws.
        accept_chain<NCH>(p, g - 1, 0, c, lane, fuse_slot, 0, 0, 1, p.ctl_next, const_cast<ChainCtl*>(p.ctl), const_cast<uint4*>(p.draws), xb, dsrc, ct);
        if (!dsrc.have) dsrc = load_draws(p, nullptr, lane);
        out = p.P + (size_t)c * p.k * p.ld; sl = p.p_slogp + (size_t)c * p.k;
    } else {
        dsrc = load_draws(p, p.draws ? p.draws + (size_t)c * p.nslots : nullptr, lane);      // (no table in a redraw round: its key differs)
        ct = p.ctl[c];
        const double* base;
        if (phase == 0) { base = p.X + (size_t)c * p.ld; out = p.P + (size_t)c * p.k * p.ld; sl = p.p_slogp + (size_t)c * p.k; }
        else {
            bool fin; int sel;
            if (p.qfin_p) {      // the proposal set's row-tile sums are still in the scratch array (no k_q_finish launch): every wave of the chain adds them
                double lp = -__builtin_huge_val();
                if (lane < p.k) {
                    const double like = like_from_q(p, p.qfin_p, (size_t)p.qfin_nc * p.k, (size_t)(c - p.qfin_c0) * p.k + lane);
                    const double prior = p.have_prior ? p.p_prior[c * p.k + lane] : 0.0;
                    lp = prior + chain_T(p, c) * like;
                    if (i0 == 0) { p.p_like[c * p.k + lane] = like; if (!p.have_prior) p.p_prior[c * p.k + lane] = 0.0; }
                }
                sel = mt_select_vals<true>(p.k, lp, ct.u_sel, lane, &fin);
            } else sel = mt_select(p, c, ct.u_sel, lane, &fin);
            if (lane == 0 && i0 == 0) p.sel[c] = sel | (fin ? 256 : 0);
            base = p.P + ((size_t)c * p.k + sel) * p.ld; out = p.R + (size_t)c * (p.k - 1) * p.ld; sl = p.r_slogp + (size_t)c * (p.k - 1);
        }
        load_row<NCH>(base, p.ld, lane, xb);
    }
    {
        const double zero = (double)(threadIdx.x >> 12);
#pragma unroll
        for (int it = 0; it < NCH; ++it) { xb[it][0] = xb[it][0] + zero; xb[it][1] = xb[it][1] + zero; }   // VALU-defined (see load_draws)
    }
    const double* grow = gamma_row(p, ct.glev, ct.delta);
    const uint32_t gc = (uint32_t)(p.off + c);
    const bool snk = ct.snk != 0;
    double* csn = (phase == 0 && p.k == 1) ? p.cur_snk + c : nullptr;
    DZ_STAMP(p, phase, c, 1);
    propose_set<NCH, true>(p, phase, g, M, c, gc, i0, i1, n, lane, xb, grow, snk, ct.cr_idx, ct.delta, ct.glev, dsrc, out, p.ld, sl, csn, nullptr);
    DZ_STAMP(p, phase, c, 15);
}

// Large d (ld > 256): one wave per (chain, try) STREAMS over the 128-dimension chunks instead of holding all of them in registers
// (k_propose<8> needs 344 registers: one wave per SIMD, every latency exposed).  Two passes, each a chunk at a time with the next
// chunk's base and archive rows in flight: DE -- pass 1 counts the crossed-over dimensions d' (the crossover uniforms are the
// first word of the pair's Philox call, which pass 2 makes again), then gamma, pass 2 makes the proposal; snooker -- pass 1 the two
// dot products, pass 2 the proposal and its distance to z.  Same arithmetic and summation order as propose_point (a lane adds its
// dimensions chunk by chunk in increasing order, then the butterfly): bit-identical to it.  DEpairs = 1, multitry >= 3.
// (BLDS: the base row lies in LDS -- k_accept_propose, where the Metropolis step of the generation before has just produced it.)
template <bool BLDS>
DZ_DEV void stream_try(const Params& p, int phase, uint32_t g, uint32_t M, uint32_t gc, int i, int lane, const double* base, const DrawSrc& dsrc,
                       const ChainCtl& ct, double* out, double* sl)
{
    const int d = p.d, ld = p.ld, nchunk = (ld + 127) >> 7;
    const uint32_t s_dim = stream_id(K_DIM, (uint32_t)i, (uint32_t)phase), s_bnd = stream_id(K_BND, (uint32_t)i, (uint32_t)phase);
    auto off = [&](int it) { return min(128 * it + 2 * lane, ld - 2); };            // (lanes past ld re-read the row's last pair; masked where used)
    auto bload = [&](int it) -> double2 {
        if (BLDS) { const dz_d2v v = *(const __attribute__((address_space(3))) dz_d2v*)(base + off(it)); double2 r; r.x = v.x; r.y = v.y; return r; }
        return gload2(base + off(it));
    };
    const bool snk = __builtin_amdgcn_readfirstlane(ct.snk) != 0;
    auto bounded = [&](double x, int j) {                                           // hard boundaries :733-791
        const double lo = p.mins[j], hi = p.maxs[j];
        const bool bl = x < lo, bh = x > hi;
        if (bl) x = 2 * lo - x;
        if (bh) x = 2 * hi - x;
        if (x < lo || x > hi) { const u32x4 w = philox(p.k0, p.k1, (uint32_t)j, s_bnd, gc, g); x = lo + u32d(w.x) * (hi - lo); }
        return x;
    };
    auto store = [&](int it, double p0, double p1) {
        const int jj = 128 * it + 2 * lane;
        if (jj < ld) {
            if (p.hard) { if (jj < d) p0 = bounded(p0, jj); if (jj + 1 < d) p1 = bounded(p1, jj + 1); }
            double2 o; o.x = jj < d ? p0 : 0.0; o.y = jj + 1 < d ? p1 : 0.0;
            *reinterpret_cast<double2*>(out + jj) = o;
        }
    };
    if (!snk) {
        const u32x4 wr = uniform_draw(p, dsrc, pt_slot(p, phase, i, 1), gc, g);
        const uint32_t r0 = __builtin_amdgcn_readfirstlane(mulhi_idx(wr.x, M));
        uint32_t r1 = __builtin_amdgcn_readfirstlane(mulhi_idx(wr.y, M - 1u));
        if (r1 >= r0) r1++;                                                         // random.sample(range(M), 2) :662
        const double* za = p.Z + (size_t)r0 * ld; const double* zb = p.Z + (size_t)r1 * ld;
        double2 xn = bload(0), an = gload2(za + off(0)), bn = gload2(zb + off(0));      // chunk 0 travels during pass 1
        const uint32_t thr = p.crthr[__builtin_amdgcn_readfirstlane(ct.cr_idx)];
        int dprime = 0;
        for (int it = 0; it < nchunk; ++it) {                                       // pass 1: d' :704 / :709
            const int j0 = 128 * it + 2 * lane;
            const u32x4 w = philox(p.k0, p.k1, (uint32_t)(j0 >> 1), s_dim, gc, g);
            const bool k0b = j0 < d && (w.x & 0xffffu) < thr, k1b = j0 + 1 < d && (w.x >> 16) < thr;
            dprime += __popcll(__builtin_amdgcn_ballot_w64(k0b)) + __popcll(__builtin_amdgcn_ballot_w64(k1b));
        }
        const u32x4 wg = uniform_draw(p, dsrc, pt_slot(p, phase, i, 0), gc, g);     // set_gamma :615
        double gamma = 1.0;
        if (!u53_below(wg.x, wg.y, p.pgu_thr)) gamma = gamma_row(p, ct.glev, 1)[(dprime == 0 ? d : dprime) - 1];   // :624
        const double ec1 = p.ec1, ec0 = p.ec0;
        for (int it = 0; it < nchunk; ++it) {                                       // pass 2: the proposal :714-726 (the Philox calls of pass 1
            const double2 x = xn, a = an, b = bn;                                   //  again; keeping their outputs in registers changed nothing)
            if (it + 1 < nchunk) { xn = bload(it + 1); an = gload2(za + off(it + 1)); bn = gload2(zb + off(it + 1)); }
            const int j0 = 128 * it + 2 * lane;
            const u32x4 w = philox(p.k0, p.k1, (uint32_t)(j0 >> 1), s_dim, gc, g);
            float z0, z1;
            normal32_pair(w.z, w.w, z0, z1);
            const bool k0b = j0 < d && (w.x & 0xffffu) < thr, k1b = j0 + 1 < d && (w.x >> 16) < thr;
            double t0 = (uniform16(w.y, ec1, ec0) + 1.0) * gamma; t0 = t0 * (a.x - b.x);              // :696-697, chain_differences :692
            double t1 = (uniform16(w.y >> 16, ec1, ec0) + 1.0) * gamma; t1 = t1 * (a.y - b.y);
            double q0 = x.x + t0; q0 = q0 + p.zeta * (double)z0;
            double q1 = x.y + t1; q1 = q1 + p.zeta * (double)z1;
            store(it, k0b ? q0 : x.x, k1b ? q1 : x.y);
        }
        if (lane == 0) sl[i] = 0.0;
    } else {
        const u32x4 wi = uniform_draw(p, dsrc, pt_slot(p, phase, i, 1), gc, g);     // :808-810
        const double* zz = p.Z + (size_t)mulhi_idx(wi.x, M) * ld;
        const double* q1r = p.Z + (size_t)mulhi_idx(wi.y, M) * ld;
        const double* q2r = p.Z + (size_t)mulhi_idx(wi.z, M) * ld;
        const u32x4 wg = uniform_draw(p, dsrc, pt_slot(p, phase, 0, 0), gc, g);
        const double gamma_s = 1.2 + (2.2 - 1.2) * u53(wg.z, wg.w);                 // :618
        double accD = 0.0, accS = 0.0;
        double2 xn = bload(0), zn = gload2(zz + off(0)), an = gload2(q1r + off(0)), bn = gload2(q2r + off(0));
        for (int it = 0; it < nchunk; ++it) {                                       // pass 1: |x - z|^2 and (zR1 - zR2).(x - z) :813-819
            const double2 x = xn, z = zn, a = an, b = bn;
            if (it + 1 < nchunk) { xn = bload(it + 1); zn = gload2(zz + off(it + 1)); an = gload2(q1r + off(it + 1)); bn = gload2(q2r + off(it + 1)); }
            const int jj = 128 * it + 2 * lane;
            const double v0 = x.x - z.x, v1 = x.y - z.y;
            if (jj < d) { accD = fma(v0, v0, accD); accS = fma(a.x - b.x, v0, accS); }
            if (jj + 1 < d) { accD = fma(v1, v1, accD); accS = fma(a.y - b.y, v1, accS); }
        }
        const double DS = wave_bfly2(accD, accS);
        const double D = readlane_f64(DS, 0);                                       // :816
        const double cc = readlane_f64(DS, 32) / D;                                 // :820
        double accN = 0.0;
        xn = bload(0); zn = gload2(zz + off(0));
        for (int it = 0; it < nchunk; ++it) {                                       // pass 2: the proposal :820-822 and its distance to z :823
            const double2 x = xn, z = zn;
            if (it + 1 < nchunk) { xn = bload(it + 1); zn = gload2(zz + off(it + 1)); }
            const int jj = 128 * it + 2 * lane;
            const double p0 = x.x + gamma_s * nan_to_num(cc * (x.x - z.x)), p1 = x.y + gamma_s * nan_to_num(cc * (x.y - z.y));
            if (jj < d) { const double t = p0 - z.x; accN = fma(t, t, accN); }
            if (jj + 1 < d) { const double t = p1 - z.y; accN = fma(t, t, accN); }
            store(it, p0, p1);
        }
        snooker_logps(p, wave_bfly(accN), 1, lane, sl + i);                         // :823-824
    }
}
#ifndef DZ_TEMPLATES_ONLY   // plain (non-template) kernels: defined once, in dz_engine.hip's translation unit
__global__ __launch_bounds__(256) void k_propose_stream(Params p, int phase, uint32_t g, uint32_t M, int c0, int nc)
{
    const int k = p.k, n = phase == 0 ? k : k - 1;
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (wave >= nc * n) return;
    const int lane = threadIdx.x & 63;
    const int c = c0 + wave / n, i = wave % n;
    const int ld = p.ld;
    const uint32_t gc = (uint32_t)(p.off + c);
    const DrawSrc dsrc = load_draws(p, p.draws + (size_t)c * p.nslots, lane);
    const ChainCtl ct = p.ctl[c];
    const double* base; double* out; double* sl;
    if (phase == 0) { base = p.X + (size_t)c * ld; out = p.P + ((size_t)c * k + i) * ld; sl = p.p_slogp + (size_t)c * k; }
    else {
        bool fin; int sel;
        if (p.qfin_p) {      // the proposal set's row-tile sums are still in the scratch array: every wave of the chain finishes them
            double lp = -__builtin_huge_val();
            if (lane < k) {
                const double like = like_from_q(p, p.qfin_p, (size_t)p.qfin_nc * k, (size_t)(c - p.qfin_c0) * k + lane);
                const double prior = p.have_prior ? p.p_prior[c * k + lane] : 0.0;
                lp = prior + chain_T(p, c) * like;
                if (i == 0) { p.p_like[c * k + lane] = like; if (!p.have_prior) p.p_prior[c * k + lane] = 0.0; }
            }
            sel = mt_select_vals<true>(k, lp, ct.u_sel, lane, &fin);
        } else sel = mt_select(p, c, ct.u_sel, lane, &fin);
        if (lane == 0 && i == 0) p.sel[c] = sel | (fin ? 256 : 0);
        base = p.P + ((size_t)c * k + sel) * ld; out = p.R + ((size_t)c * (k - 1) + i) * ld; sl = p.r_slogp + (size_t)c * (k - 1);
    }
    stream_try<false>(p, phase, g, M, gc, i, lane, base, dsrc, ct, out, sl);
}
#endif  // DZ_TEMPLATES_ONLY

// debug entry: flags supplied by the host (function-level parity tests); one wave, all tries
template <int NCH>
__global__ __launch_bounds__(64) void k_propose_debug(Params p, int phase, uint32_t g, uint32_t M, int c, int n,
                                                     const double* base, double* out, double* sl, int snk, int cr_idx, int delta, int glev)
{
    const int lane = threadIdx.x & 63;
    double xb[NCH][2];
    load_row<NCH>(base, p.ld, lane, xb);
    ZRows<NCH> zr;
    const double* grow = gamma_row(p, glev, delta);
    const DrawSrc none = load_draws(p, nullptr, lane);
    double sqv = 1.0;
    for (int i = 0; i < n; ++i) {
        fetch_rows<NCH>(p, phase, g, M, (uint32_t)(p.off + c), i, lane, snk != 0, delta, none, zr);
        RowTerms<NCH> rt;
        reduce_rows<NCH>(zr, snk != 0, rt);
        const double sq = propose_point<NCH>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * p.ld, nullptr, snk != 0, cr_idx, delta, glev, none);
        sqv = (lane == i) ? sq : sqv;
        if (!snk && lane == 0) sl[i] = 0.0;
    }
    if (snk) snooker_logps(p, sqv, n, lane, sl);
}

// ------------------------------------------------------------------------------------------
// Model.total_logp (model.py:17-32) on the device.
// ------------------------------------------------------------------------------------------

// MVN likelihood (examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52) on the FP64 matrix pipe.
// v_mfma_f64_16x16x4_f64 accumulates exactly like an ascending-k fma chain (tools/mfma_f64_layout.hip,
// measured on gfx950: 0 mismatches in 51200 outputs, 77.6 TFLOP/s), so y[p][r] = sum_c M[r][c] v[p][c]
// (ascending c) comes out bit-identical to the scalar contract.  One wave = PT tiles of 16 points:
//   A operand (16 rows x 4 cols):   lane l supplies Mt[4 ks + l/16][16 t + l%16]   (shared by the PT tiles)
//   B operand (4 cols x 16 points): lane l supplies v[p0 + l%16][4 ks + l/16]
//   D: lane l, element e holds y[point p0 + l%16][row 16 t + l/16 + 4 e]      (the transposed tile: Y^T = M V^T)
// Q contract (v2): Q = q_0 + q_1 + ... in ascending row tile t; inside a tile the lane adds its four rows' products y_r s_r by an
// fma chain and the four lane groups are combined by two lane-swap steps (tile_q_tri / tile_q above) -- so the quadratic form never
// leaves the registers, and row tiles can be produced by different waves.  s = v (dense precision) or y (triangular factor; the
// k-steps left of a diagonal tile are structural zeros and are skipped).
typedef double dz_double4 __attribute__((ext_vector_type(4)));

// Row-tile sum q_t of the MVN contract (v2) from one MFMA accumulator in the TRANSPOSED tile layout -- matrix = A operand, points =
// B operand: lane l holds, for point (l & 15), the rows r = 16 t + (l >> 4) + 4 e, e = 0..3, of y.
//   s_kq = fma chain over e = 0..3 of y_r s_r (rows r >= d contribute nothing), q_t = (s_0 + s_2) + (s_1 + s_3)
// i.e. two lane-swap steps (lanes ^32, then ^16); every lane of a point ends with q_t.
DZ_DEV double tile_q_tri(const dz_double4& acc)     // s = y (triangular factor); rows r >= d have y == +0 exactly: fma(+0, +0, s) == s
{
    double s = fma(acc[0], acc[0], 0.0);
    s = fma(acc[1], acc[1], s); s = fma(acc[2], acc[2], s); s = fma(acc[3], acc[3], s);
    return swap16_sum(swap32_sum(s));
}
DZ_DEV double tile_q(const dz_double4& acc, const double (&sv)[4], int r0, int d)     // general: row r0 + 4 e takes part iff it is < d
{
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) s = (r0 + 4 * e < d) ? fma(acc[e], sv[e], s) : s;
    return swap16_sum(swap32_sum(s));
}

// ld <= 128: NRT = ld/16 row tiles, everything in registers, k loop fully unrolled
template <int PT, int NRT, bool TRI>
__global__ __launch_bounds__(256) void k_logp_mvn_mfma(Params p, const double* __restrict__ pts, int npts, double* prior_out, double* like_out)
{
    constexpr int KSP = NRT * 4;
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int p0 = (blockIdx.x * 4 + wv) * 16 * PT;
    if (p0 >= npts) return;
    const int pi = l & 15, kq = l >> 4;
    const int ld = p.ld, d = p.d;
    const int KS = (d + 3) >> 2;
    double A[PT][KSP];
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const double* xr = pts + (size_t)min(p0 + 16 * u + pi, npts - 1) * ld;
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) A[u][ks] = xr[4 * ks + kq] - p.mu[4 * ks + kq];
    }
    dz_double4 acc[PT][NRT];
#pragma unroll
    for (int u = 0; u < PT; ++u)
#pragma unroll
        for (int t = 0; t < NRT; ++t) acc[u][t] = dz_double4{0.0, 0.0, 0.0, 0.0};
    const double* mbase = p.Mt + (size_t)kq * ld + pi;
#pragma unroll
    for (int ks = 0; ks < KSP; ++ks) {
        if (ks < KSP - 4 || ks < KS) {     // only the last tile column can lie beyond d
            const double* mrow = mbase + (size_t)(4 * ks) * ld;
#pragma unroll
            for (int t = 0; t < NRT; ++t) {
                if (!TRI || ks >= 4 * t) {
                    const double b = mrow[16 * t];
#pragma unroll
                    for (int u = 0; u < PT; ++u) acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b, A[u][ks], acc[u][t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int pt = p0 + 16 * u + pi;
        const double* xs = pts + (size_t)min(pt, npts - 1) * ld;
        double q = 0.0;
#pragma unroll
        for (int t = 0; t < NRT; ++t) {      // Q = q_0 + q_1 + ... (ascending t)
            if (TRI) q = q + tile_q_tri(acc[u][t]);      // (Mt is zero padded: rows r >= d are exact zeros)
            else {
                double sv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int r = 16 * t + kq + 4 * e; sv[e] = xs[r] - p.mu[r]; }
                q = q + tile_q(acc[u][t], sv, 16 * t + kq, d);
            }
        }
        if (kq == 0 && pt < npts) {
            like_out[pt] = nan_to_ninf(p.logF - 0.5 * q);
            if (!p.have_prior) prior_out[pt] = 0.0;
        }
    }
}

// Persistent variant (ld <= 128): operands through LDS.
//  * The 16 points of a tile are 16*ld contiguous doubles: the wave streams them with full-width coalesced
//    loads (16 B per lane) into a private LDS tile and only then picks the A operands out of it in the
//    MFMA layout.  (Fetching the A layout straight from global memory -- 16 rows x 32 B per instruction,
//    each line revisited four times -- measured 3-7x the footprint in L2 misses and 21-32% matrix-pipe
//    utilisation.)
//  * The matrix (the same for every point) is staged once per block; each block starts its copy at a
//    different offset so the CUs do not sweep the L2 channels in lockstep.  B reads are conflict-free
//    (16 consecutive doubles per k-row, k-rows 896 B apart = bank offset 32).
// One block (4 waves, one per SIMD) per CU loops over the point tiles; the next tile's rows are fetched
// into registers while the current tile's MFMAs run.
template <int NRT, bool TRI>
__global__ __launch_bounds__(512) void k_logp_mvn_lds(Params p, const double* __restrict__ pts, int npts, double* prior_out, double* like_out)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int KSP = NRT * 4;
    constexpr int LD = NRT * 16;          // == p.ld
    constexpr int LDT = LD + 1;           // padded tile row: the A-layout reads hit distinct banks
    constexpr int NV = LD / 8;            // double2 loads per lane for one tile (16*LD/2/64)
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, nwv = blockDim.x >> 6, bdim = blockDim.x;   // 4..8 waves: one per point tile of the CU's share
    const int d = p.d;
    const int KS = (d + 3) >> 2;
    double* Ms = smem;                                    // [4*KS][LD], or the packed triangle
    double* mus = Ms + (TRI ? (size_t)p.mtp_len : (size_t)4 * KS * LD);      // [LD]
    double* Vt = mus + LD + (size_t)wv * 16 * LDT;        // [16][LDT] per wave
    const int pi = l & 15, kq = l >> 4;
    const int ntiles = (npts + 15) >> 4;
    const double* mbase = Ms + (size_t)kq * LD + pi;
    int tile = blockIdx.x * nwv + wv;
    double2 nx[NV];
    auto fetch = [&](int tl) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int i2 = 2 * (l + 64 * q);                      // element index inside the 16 x LD tile
            const int row = i2 / LD, col = i2 - row * LD;
            const int pt = min(tl * 16 + row, npts - 1);
            nx[q] = *reinterpret_cast<const double2*>(pts + (size_t)pt * LD + col);
        }
    };
    DZ_LSTAMP(p, blockIdx.x * nwv + wv, 0);
    if (tile < ntiles) fetch(tile);                // the first tile's rows are requested before the matrix
    {   // stage the matrix: all loads of a batch are issued before the first LDS store, so the copy costs
        // one memory latency per batch, not one per element.  TRI: the packed layout (52% of the square at d=100).
        const int nvec = TRI ? (p.mtp_len >> 1) : ((4 * KS * LD) >> 1);
        const double2* src = reinterpret_cast<const double2*>(TRI ? p.Mtp : p.Mt);
        double2* dst = reinterpret_cast<double2*>(Ms);
        const int start = (int)(((size_t)blockIdx.x * 1024) % (size_t)nvec);
        constexpr int NIT = TRI ? (64 * NRT * (NRT + 1) + 255) / 256 : (4 * KSP * LD / 2 + 255) / 256;       // upper bound on elements per thread
        constexpr int BATCH = NIT;
#pragma unroll
        for (int b0 = 0; b0 < NIT; b0 += BATCH) {
            double2 tmp[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int i = threadIdx.x + bdim * (b0 + u);
                int j = i + start; if (j >= nvec) j -= nvec;
                tmp[u] = (b0 + u < NIT && i < nvec) ? src[j] : double2{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int i = threadIdx.x + bdim * (b0 + u);
                int j = i + start; if (j >= nvec) j -= nvec;
                if (b0 + u < NIT && i < nvec) dst[j] = tmp[u];
            }
        }
        if (threadIdx.x < LD) mus[threadIdx.x] = p.mu[threadIdx.x];
    }
    DZ_LSTAMP(p, blockIdx.x * nwv + wv, 1);
    __syncthreads();
    DZ_LSTAMP(p, blockIdx.x * nwv + wv, 2);
    for (; tile < ntiles; tile += gridDim.x * nwv) {
        const int p0 = tile * 16;
        {   // two passes: the compiler cannot move an LDS read of mus across an LDS write to Vt, so reading inside the
            // store loop costs one exposed LDS latency per element (measured ~3000 cycles per tile)
            double2 mv[NV];
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i2 = 2 * (l + 64 * q);
                const int col = i2 - (i2 / LD) * LD;
                mv[q] = *reinterpret_cast<const double2*>(mus + col);
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int i2 = 2 * (l + 64 * q);
                const int row = i2 / LD, col = i2 - row * LD;
                Vt[row * LDT + col] = nx[q].x - mv[q].x;
                Vt[row * LDT + col + 1] = nx[q].y - mv[q].y;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (tile + (int)gridDim.x * nwv < ntiles) fetch(tile + gridDim.x * nwv);
        DZ_LSTAMP(p, blockIdx.x * nwv + wv, 3);
        double A[KSP];
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) A[ks] = Vt[pi * LDT + 4 * ks + kq];
        dz_double4 acc[NRT];
#pragma unroll
        for (int t = 0; t < NRT; ++t) acc[t] = dz_double4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) {
            if (ks < KSP - 4 || ks < KS) {
                // k-rows 4ks..4ks+3 (lane group kq), columns 16t+pi
                const double* mrow = TRI ? Ms + tri_row_offset(4 * ks) + kq * (16 * (ks / 4 + 1)) + pi : mbase + (size_t)(4 * ks) * LD;
#pragma unroll
                for (int t = 0; t < NRT; ++t)
                    if (!TRI || ks >= 4 * t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(mrow[16 * t], A[ks], acc[t], 0, 0, 0);
            }
        }
        DZ_LSTAMP(p, blockIdx.x * nwv + wv, 4);
        {
            const int pt = p0 + pi;
            double q = 0.0;
#pragma unroll
            for (int t = 0; t < NRT; ++t) {      // Q = q_0 + q_1 + ... (ascending t)
                if (TRI) q = q + tile_q_tri(acc[t]);     // (the packed triangle is zero padded: rows r >= d are exact zeros)
                else {
                    double sv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) sv[e] = Vt[pi * LDT + 16 * t + kq + 4 * e];
                    q = q + tile_q(acc[t], sv, 16 * t + kq, d);
                }
            }
            if (kq == 0 && pt < npts) {
                like_out[pt] = nan_to_ninf(p.logF - 0.5 * q);
                if (!p.have_prior) prior_out[pt] = 0.0;
            }
        }
        DZ_LSTAMP(p, blockIdx.x * nwv + wv, 5);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}

// any ld <= 1024: row tiles RTC at a time, operands fetched one k-step ahead
template <int RTC>
__global__ __launch_bounds__(256) void k_logp_mvn_mfma_big(Params p, const double* __restrict__ pts, int npts, double* prior_out, double* like_out)
{
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int p0 = (blockIdx.x * 4 + wv) * 16;
    if (p0 >= npts) return;
    const int pi = l & 15, kq = l >> 4;
    const int ld = p.ld, d = p.d;
    const double* xrow = pts + (size_t)min(p0 + pi, npts - 1) * ld;
    const int KS = (d + 3) >> 2, NRT = (d + 15) >> 4;
    double Q = 0.0;
    for (int rt0 = 0; rt0 < NRT; rt0 += RTC) {
        dz_double4 acc[RTC];
#pragma unroll
        for (int t = 0; t < RTC; ++t) acc[t] = dz_double4{0.0, 0.0, 0.0, 0.0};
        const int ks0 = p.tri ? 4 * rt0 : 0;
        double a_n = xrow[4 * ks0 + kq] - p.mu[4 * ks0 + kq];
        double b_n[RTC];
#pragma unroll
        for (int t = 0; t < RTC; ++t) b_n[t] = p.Mt[(size_t)(4 * ks0 + kq) * ld + 16 * min(rt0 + t, NRT - 1) + pi];
        for (int ks = ks0; ks < KS; ++ks) {
            const double a = a_n;
            double b[RTC];
#pragma unroll
            for (int t = 0; t < RTC; ++t) b[t] = b_n[t];
            if (ks + 1 < KS) {
                const int c = 4 * (ks + 1) + kq;
                a_n = xrow[c] - p.mu[c];
#pragma unroll
                for (int t = 0; t < RTC; ++t) b_n[t] = p.Mt[(size_t)c * ld + 16 * min(rt0 + t, NRT - 1) + pi];
            }
#pragma unroll
            for (int t = 0; t < RTC; ++t) {
                const int rt = rt0 + t;
                if (rt < NRT && (!p.tri || ks >= 4 * rt)) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[t], a, acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < RTC; ++t) {
            if (rt0 + t < NRT) {
                double sv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int r = 16 * (rt0 + t) + kq + 4 * e; sv[e] = p.tri ? acc[t][e] : xrow[r] - p.mu[r]; }
                Q = Q + tile_q(acc[t], sv, 16 * (rt0 + t) + kq, d);
            }
        }
    }
    {
        const int pt = p0 + pi;
        if (kq == 0 && pt < npts) {
            like_out[pt] = nan_to_ninf(p.logF - 0.5 * Q);
            if (!p.have_prior) prior_out[pt] = 0.0;
        }
    }
}

// Large d (ld > 128): the quadratic form as a tiled product.  One wave owns PT point tiles x RTC row tiles and walks k
// once (B operands shared by the PT tiles, A operands by the RTC row tiles); the grid covers point-tile groups x
// row-tile groups, so 2560 points x 63 row tiles give 1280 waves instead of the 160 of the one-wave-per-point-tile
// form (C5: 905 -> see DESIGN.md).  Row-tile sums q[point][t] go to a scratch array; k_q_finish adds them in ascending
// t, which keeps the contract (Q = q_0 + q_1 + ...) independent of how the tiles were spread.
template <int PT, int RTC>
__global__ __launch_bounds__(256) void k_logp_mvn_mfma_tiled(Params p, const double* __restrict__ pts, int npts, double* __restrict__ qpart)
{
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int pi = l & 15, kq = l >> 4;
    const int ld = p.ld, d = p.d;
    const int KS = (d + 3) >> 2, NRT = (d + 15) >> 4;
    const int nrg = (NRT + RTC - 1) / RTC;
    const int unit = blockIdx.x * 4 + wv;                 // (point group, row group), row group fastest: neighbours share A rows
    const int pg = unit / nrg, rg = unit - pg * nrg;
    const int p0 = pg * 16 * PT;
    if (p0 >= npts) return;
    const int rt0 = rg * RTC;
    const double* xrow[PT];
#pragma unroll
    for (int u = 0; u < PT; ++u) xrow[u] = pts + (size_t)min(p0 + 16 * u + pi, npts - 1) * ld;
    dz_double4 acc[PT][RTC];
#pragma unroll
    for (int u = 0; u < PT; ++u)
#pragma unroll
        for (int t = 0; t < RTC; ++t) acc[u][t] = dz_double4{0.0, 0.0, 0.0, 0.0};
    const int ks0 = p.tri ? 4 * rt0 : 0;
    double a_n[PT], b_n[RTC];
    {
        const int c = 4 * ks0 + kq;
#pragma unroll
        for (int u = 0; u < PT; ++u) a_n[u] = xrow[u][c] - p.mu[c];
#pragma unroll
        for (int t = 0; t < RTC; ++t) b_n[t] = p.Mt[(size_t)c * ld + 16 * min(rt0 + t, NRT - 1) + pi];
    }
    for (int ks = ks0; ks < KS; ++ks) {
        double a[PT], b[RTC];
#pragma unroll
        for (int u = 0; u < PT; ++u) a[u] = a_n[u];
#pragma unroll
        for (int t = 0; t < RTC; ++t) b[t] = b_n[t];
        if (ks + 1 < KS) {                                // operands of the next k-step are in flight during this one's MFMAs
            const int c = 4 * (ks + 1) + kq;
#pragma unroll
            for (int u = 0; u < PT; ++u) a_n[u] = xrow[u][c] - p.mu[c];
#pragma unroll
            for (int t = 0; t < RTC; ++t) b_n[t] = p.Mt[(size_t)c * ld + 16 * min(rt0 + t, NRT - 1) + pi];
        }
#pragma unroll
        for (int t = 0; t < RTC; ++t) {
            const int rt = rt0 + t;
            if (rt < NRT && (!p.tri || ks >= 4 * rt)) {
#pragma unroll
                for (int u = 0; u < PT; ++u) acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[t], a[u], acc[u][t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int pt = p0 + 16 * u + pi;
#pragma unroll
        for (int t = 0; t < RTC; ++t) {
            if (rt0 + t < NRT) {
                double sv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int r = 16 * (rt0 + t) + kq + 4 * e; sv[e] = p.tri ? acc[u][t][e] : xrow[u][r] - p.mu[r]; }
                const double qv = tile_q(acc[u][t], sv, 16 * (rt0 + t) + kq, d);
                if (kq == 0 && pt < npts) qpart[(size_t)(rt0 + t) * npts + pt] = qv;          // [row tile][point]: k_q_finish reads it coalesced
            }
        }
    }
}
// Large d, many points: the quadratic form as an LDS-tiled product.  A block of 4 waves owns 64 points x 64 rows; it
// walks k in chunks of 16, staging the point chunk (transposed to [k][point], mean subtracted) and the matrix chunk
// ([k][row], as stored) in LDS, double buffered; wave (wm, wn) owns 2 point tiles x 2 row tiles = four accumulators, each
// summed over k in ascending order by one MFMA chain -- the contract's order.  Row-tile sums go to the scratch array
// of k_q_finish.  Triangular factor: a block starts at k = 64*bn and a row tile joins at k = 16*rt.
template <int PT>      // point tiles per wave: the block owns BM = 32 PT points (64 or 128) x 64 rows
__global__ __launch_bounds__(256) void k_logp_mvn_gemm(Params p, const double* __restrict__ pts, int npts, double* __restrict__ qpart, int ncu)
{
    constexpr int BM = 32 * PT, BN = 64, BK = 16, LDA = BM + 2, LDB = BN;
    __shared__ __attribute__((aligned(16))) double As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) double Bs[2][BK * LDB];
    extern __shared__ __attribute__((aligned(16))) double mus[];      // the mean (ld doubles), staged once -- unless it is all zero
    const int tid = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int pi = l & 15, kq = l >> 4;
    const int wm = wv >> 1, wn = wv & 1;
    const int ld = p.ld, d = p.d;
    const int KS = (d + 3) >> 2, NRT = (d + 15) >> 4;
    const int nbn = (NRT * 16 + BN - 1) / BN;
    // All blocks are resident at once and land on the CUs round-robin, so nothing rebalances the triangular factor's
    // uneven blocks (bn = 0 walks all of k, the last bn almost none).  Blocks are therefore numbered from a list sorted
    // heaviest first, taken alternately from its two ends per round of ncu blocks: every CU gets heavy + light.
    // (Tried, no gain each: an XCD-aware numbering -- an XCD keeps a fixed set of point blocks, matrix panels shared by neighbours --;
    //  wave priorities for the long blocks; padded matrix rows in LDS; the per-MFMA predicates hoisted out of the steady state; three
    //  LDS stages with the next chunk's operands read during the MFMAs (+3 us); blocks numbered heaviest first with 3..6 resident per
    //  CU so that the dispatcher balances (+3..5 us); resident blocks pulling units from an atomic counter (+16 us: 2.5 k atomics on
    //  one address).  What bounds it -- tools/micro/mfma_mem_overlap.hip: while a SIMD's FP64 matrix pipe is busy, data returning to
    //  its registers (LDS reads, global loads, LDS-direct loads alike) makes no progress, so a chunk's operand traffic adds to its
    //  MFMA time instead of hiding behind it: knocking out the global loads saves 16 us, the LDS reads 10, the LDS stores 7, the MFMAs
    //  19 of the launch's 55; at 20 k points, where balance no longer matters, the three tile heights reach 46 / 50 / 46 TFLOP/s.)
    const int nbm = (int)gridDim.x / nbn;
    int li;
    {
        const int b = blockIdx.x, q = b / ncu, c = b - q * ncu;
        li = (q & 1) ? (int)gridDim.x - 1 - ((q >> 1) * ncu + c) : (q >> 1) * ncu + c;
    }
    const int bn = li / nbm, bm = li - bn * nbm;
    const int p0 = bm * BM;
    const int kc0 = p.tri ? (BN * bn) / BK : 0, nkc = (4 * KS + BK - 1) / BK;       // chunks of 16 k
    // loader roles -- every wave-wide load covers whole 128-byte lines (eight lanes x 16 bytes per line).  A: a point's chunk of
    // 16 k is one line: thread -> (point tid / 8 + 32 a, k's 2 (tid % 8), +1); B: a k row of the block's 64 matrix rows is four
    // lines: thread -> (k row tid / 32 + 8 h, rows 2 (tid % 32), +1)
    const int a_pt = tid >> 3, a_k = 2 * (tid & 7);
    const int b_k = tid >> 5, b_r = 2 * (tid & 31);
    const double* arow[PT];
#pragma unroll
    for (int a = 0; a < PT; ++a) arow[a] = pts + (size_t)min(p0 + a_pt + 32 * a, npts - 1) * ld;
    // rows of pts, mu and Mt are zero padded to ld (a multiple of 16 >= 4*KS), so whole 16-byte groups can be read.
    // Two register sets: a chunk's loads are issued two chunks ahead (one chunk of MFMAs is ~0.4 us, less than a memory
    // round trip) and always issued (index clamped), so that the wait in front of the LDS store covers exactly the older set.
    // The mean is subtracted when the chunk goes to LDS, not at the load (subtracting there made the loop wait for the loads
    // before the MFMAs instead of after them), and it comes from LDS, not with every chunk.
    struct Stage { double2 ra[PT], rb[2]; };
    const bool mz = p.mu_zero != 0;
    if (!mz) { for (int i = tid; i < ld; i += 256) mus[i] = p.mu[i]; }
    auto gload = [&](int kc, Stage& R) {
        // no predicates (a select after a load makes the loop wait for it at once): k < 16 nkc <= ld always; a row group past
        // ld (last bn only) is read from the last valid group instead -- those output rows are dropped below (rt0 + t < NRT)
        kc = min(kc, nkc - 1);
        const int k = kc * BK + a_k;
#pragma unroll
        for (int a = 0; a < PT; ++a) R.ra[a] = *reinterpret_cast<const double2*>(arow[a] + k);
        const int r = min(BN * bn + b_r, ld - 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) R.rb[h] = *reinterpret_cast<const double2*>(p.Mt + (size_t)(kc * BK + b_k + 8 * h) * ld + r);
    };
    auto lstore = [&](int buf, int kc, const Stage& R) {
        const int k = min(kc, nkc - 1) * BK + a_k;
#pragma unroll
        for (int a = 0; a < PT; ++a) {                             // (x - 0.0 == x, bit for bit)
            As[buf][a_k * LDA + a_pt + 32 * a] = mz ? R.ra[a].x : R.ra[a].x - mus[k];
            As[buf][(a_k + 1) * LDA + a_pt + 32 * a] = mz ? R.ra[a].y : R.ra[a].y - mus[k + 1];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) *reinterpret_cast<double2*>(&Bs[buf][(b_k + 8 * h) * LDB + b_r]) = R.rb[h];
    };
    dz_double4 acc[PT][2];
#pragma unroll
    for (int u = 0; u < PT; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[u][t] = dz_double4{0.0, 0.0, 0.0, 0.0};
    const int rt0 = (BN / 16) * bn + 2 * wn;                       // this wave's first row tile
    // the chunk's operand reads first, then its MFMAs back to back (read-then-use per k-step left an LDS round trip in
    // front of every pair of MFMAs); every accumulator is one chain over k in ascending order -- the contract's order
    auto compute = [&](int kc, int buf) {
        double a[BK / 4][PT], b[BK / 4][2];
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
#pragma unroll
            for (int u = 0; u < PT; ++u) a[q][u] = As[buf][(4 * q + kq) * LDA + 16 * PT * wm + 16 * u + pi];
#pragma unroll
            for (int t = 0; t < 2; ++t) b[q][t] = Bs[buf][(4 * q + kq) * LDB + 32 * wn + 16 * t + pi];
        }
#pragma unroll
        for (int q = 0; q < BK / 4; ++q) {
            const int ks = kc * (BK / 4) + q;
#pragma unroll
            for (int t = 0; t < 2; ++t)
                if (ks < KS && rt0 + t < NRT && (!p.tri || ks >= 4 * (rt0 + t))) {
#pragma unroll
                    for (int u = 0; u < PT; ++u) acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[q][t], a[q][u], acc[u][t], 0, 0, 0);
                }
        }
    };
    Stage R0, R1;
    gload(kc0, R0);
    __syncthreads();                                               // the staged mean
    lstore(0, kc0, R0);
    gload(kc0 + 1, R0); gload(kc0 + 2, R1);
    __syncthreads();
    for (int kc = kc0; kc < nkc; kc += 2) {
        compute(kc, 0);
        lstore(1, kc + 1, R0);                                     // chunk kc+1 (loaded two chunks ago)
        gload(kc + 3, R0);
        __syncthreads();
        if (kc + 1 >= nkc) break;
        compute(kc + 1, 1);
        lstore(0, kc + 2, R1);                                     // chunk kc+2
        gload(kc + 4, R1);
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < PT; ++u) {
        const int pt = p0 + 16 * PT * wm + 16 * u + pi;
        const double* xs = pts + (size_t)min(pt, npts - 1) * ld;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (rt0 + t < NRT) {
                double sv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int r = 16 * (rt0 + t) + kq + 4 * e; sv[e] = p.tri ? acc[u][t][e] : xs[r] - p.mu[r]; }
                const double qv = tile_q(acc[u][t], sv, 16 * (rt0 + t) + kq, d);
                if (kq == 0 && pt < npts) qpart[(size_t)(rt0 + t) * npts + pt] = qv;          // [row tile][point]: k_q_finish reads it coalesced
            }
        }
    }
}

#ifndef DZ_TEMPLATES_ONLY   // plain (non-template) kernels: defined once, in dz_engine.hip's translation unit
// tlogp [generation][chain] -> [chain][generation] for the chain-by-chain download (dz_get_trace_chains)
__global__ void k_transpose_logp(const double* __restrict__ src, int nl, int64_t g0, int ng, double* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // over dst: c * ng + g
    if (i >= (int64_t)nl * ng) return;
    const int64_t c = i / ng, g = i - c * ng;
    dst[i] = src[(g0 + g) * nl + c];
}
__global__ void k_q_finish(Params p, const double* __restrict__ qpart, int npts, int nrt, double* prior_out, double* like_out)
{
    const int pt = blockIdx.x * blockDim.x + threadIdx.x;
    if (pt >= npts) return;
    // ascending t (the contract's order); eight loads in flight at a time -- one load per add left a memory round trip in
    // front of every term (18 us for 63 terms)
    double Q = 0.0;
    int t = 0;
    for (; t + 8 <= nrt; t += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = qpart[(size_t)(t + j) * npts + pt];
#pragma unroll
        for (int j = 0; j < 8; ++j) Q = Q + v[j];
    }
    for (; t < nrt; ++t) Q = Q + qpart[(size_t)t * npts + pt];
    like_out[pt] = nan_to_ninf(p.logF - 0.5 * Q);
    if (!p.have_prior) prior_out[pt] = 0.0;
}

// prior only (wave per point), used beside the MFMA likelihood kernel when priors are not flat
// log of the priors' scale parameters (scipy norm / uniform `scale`), evaluated once instead of per point and dimension
__global__ void k_prior_consts(const double* __restrict__ pb, int n, double* __restrict__ plogb)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) plogb[j] = dlog(pb[j]);
}
#endif  // DZ_TEMPLATES_ONLY
template <int NCH>
__global__ __launch_bounds__(256) void k_prior_only(Params p, const double* __restrict__ pts, int npts, double* prior_out)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= npts) return;
    double x[NCH][2];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        x[it][0] = 0.0; x[it][1] = 0.0;
        if (jj < p.ld) { const double2 t = *reinterpret_cast<const double2*>(pts + (size_t)pt * p.ld + jj); x[it][0] = t.x; x[it][1] = t.y; }
    }
    const double prior = prior_of_point<NCH>(p, x, lane);
    if (lane == 0) prior_out[pt] = nan_to_ninf(prior);
}

// log-likelihood of one point under the Gaussian mixture (identity covariances, mixturemodel.py:37-48); the point sits in
// the wave's registers, lh is scratch for the J component terms (any memory the wave owns)
template <int NCH>
DZ_DEV double mix_like_wave(const Params& p, const double (&x)[NCH][2], int lane, double* __restrict__ lh)
{
    double mx = -__builtin_huge_val();
    for (int j = 0; j < p.J; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int jd = 128 * it + 2 * lane + s;
                if (jd < p.d) { const double t = x[it][s] - p.mu[(size_t)j * p.ld + jd]; acc = fma(t, t, acc); }
            }
        const double S = wave_bfly(acc);
        const double v = -0.5 * S + p.mixF[j];
        if (lane == 0) lh[j] = v;
        if (v > mx) mx = v;
    }
    double dens = 0.0;
    for (int j = 0; j < p.J; ++j) dens = dens + dexp(lh[j] - mx);
    return nan_to_ninf(dlog(dens) + mx);
}

// Gaussian mixture, identity covariances (examples/mixturemodel/mixturemodel.py:37-48)
template <int NCH>
__global__ __launch_bounds__(256) void k_logp_mix(Params p, const double* __restrict__ pts, int npts, double* prior_out, double* like_out)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= npts) return;
    double x[NCH][2];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        x[it][0] = 0.0; x[it][1] = 0.0;
        if (jj < p.ld) { const double2 t = *reinterpret_cast<const double2*>(pts + (size_t)pt * p.ld + jj); x[it][0] = t.x; x[it][1] = t.y; }
    }
    const double prior = prior_of_point<NCH>(p, x, lane);
    double lh[32]; double mx = -__builtin_huge_val();
    for (int j = 0; j < p.J; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int jd = 128 * it + 2 * lane + s;
                if (jd < p.d) { const double t = x[it][s] - p.mu[(size_t)j * p.ld + jd]; acc = fma(t, t, acc); }
            }
        const double S = wave_bfly(acc);
        lh[j] = -0.5 * S + p.mixF[j];
        if (lh[j] > mx) mx = lh[j];
    }
    double dens = 0.0;
    for (int j = 0; j < p.J; ++j) dens = dens + dexp(lh[j] - mx);
    if (lane == 0) { prior_out[pt] = nan_to_ninf(prior); like_out[pt] = nan_to_ninf(dlog(dens) + mx); }
}

// host-likelihood path: add the built-in prior to what the callback returned, map NaN to -inf
template <int NCH>
__global__ __launch_bounds__(256) void k_prior_add(Params p, const double* __restrict__ pts, int npts, double* prior_io, double* like_io)
{
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    if (pt >= npts) return;
    double x[NCH][2];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        x[it][0] = 0.0; x[it][1] = 0.0;
        if (jj < p.ld) { const double2 t = *reinterpret_cast<const double2*>(pts + (size_t)pt * p.ld + jj); x[it][0] = t.x; x[it][1] = t.y; }
    }
    const double prior = prior_of_point<NCH>(p, x, lane);
    if (lane == 0) { prior_io[pt] = nan_to_ninf(p.have_prior ? prior_io[pt] + prior : prior_io[pt]); like_io[pt] = nan_to_ninf(like_io[pt]); }
}

// ------------------------------------------------------------------------------------------
// MT ratio (:305-323) / single-try ratio (:325-334), metrop_select (:980-998), state update
// (:336-347), trace (core.py:114-116), record_history (:919-938), set_current_position_arr
// (:424-449).  One wave per chain.
// ------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(1024) void k_accept(Params p, uint32_t g, int64_t zbase, int c0, int nc, int64_t trace_slot, int append, int publish, int prep_next)
{
    const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (wave >= nc) return;
    double xn[NCH][2]; DrawSrc dn; ChainCtl cn;
    accept_chain<NCH>(p, g, zbase, c0 + wave, threadIdx.x & 63, trace_slot, append, publish, prep_next, p.ctl, p.ctl_next, p.draws_next, xn, dn, cn);
}

// Large d (k_propose_stream's domain), a generation that neither appends nor publishes, another one following: the Metropolis step of
// generation g and the proposal set of generation g+1 in one launch.  One block per chain, one wave per try: wave 0 makes the step
// (accept_chain: state, trace, the draws and decisions of generation g+1) and leaves the new state, the draws and the decisions in
// LDS; behind the barrier wave i makes try i around it (stream_try) -- the set's rows in P are free by then, wave 0 has read the
// selected one.  M: the archive rows generation g+1 may sample (no append in between: the same as generation g's).
template <int NCH>
__global__ __launch_bounds__(1024) void k_accept_propose(Params p, uint32_t g, int64_t zbase, int c0, int nc, int64_t trace_slot, uint32_t M)
{
    extern __shared__ __attribute__((aligned(16))) double xs[];           // [ld] state | 64 uint4 draws | ChainCtl
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int c = c0 + blockIdx.x;
    uint4* sdraw = reinterpret_cast<uint4*>(xs + p.ld);
    ChainCtl* sctl = reinterpret_cast<ChainCtl*>(sdraw + 64);
    if (wv == 0) {
        double xn[NCH][2]; DrawSrc dn; ChainCtl cn;
        accept_chain<NCH>(p, g, zbase, c, lane, trace_slot, 0, 0, 1, p.ctl, p.ctl_next, p.draws_next, xn, dn, cn);
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int jj = 128 * it + 2 * lane;
            if (jj < p.ld) { xs[jj] = xn[it][0]; xs[jj + 1] = xn[it][1]; }
        }
        sdraw[lane] = dn.mine;
        if (lane == 0) *sctl = cn;
    }
    __syncthreads();
    DrawSrc dsrc; dsrc.have = p.nslots <= 64; dsrc.mine = sdraw[lane];
    {
        const unsigned zero = threadIdx.x >> 12;        // (VALU-defined registers for v_readlane, see load_draws)
        dsrc.mine.x += zero; dsrc.mine.y += zero; dsrc.mine.z += zero; dsrc.mine.w += zero;
    }
    const ChainCtl ct = *sctl;
    stream_try<true>(p, 0, g + 1, M, (uint32_t)(p.off + c), wv, lane, xs, dsrc, ct, p.P + ((size_t)c * p.k + wv) * p.ld, p.p_slogp + (size_t)c * p.k);
}

// uniform draws of generation g for local chains [c0, c0+nc): one lane per (chain, slot); the lanes of
// slot 0 also derive the chain's control decisions (needs the CURRENT crossover / gamma-level probabilities)
// Temperature swap of parallel tempering (core.py:185-221): one random pair per generation, after every chain's step
// and the end-of-generation updates.  One wave; lanes share the row exchange.  Stream SWAP (=4) of the random contract.
template <int NCH>
__global__ __launch_bounds__(64) void k_pt_swap(Params p, uint32_t g, int64_t trace_slot, int published)
{
    const int lane = threadIdx.x & 63;
    const u32x4 w = philox(p.k0, p.k1, 0u, stream_id(4u, 0u, 0u), 0u, g);
    const uint32_t a = mulhi_idx(w.x, (uint32_t)p.N);
    uint32_t b = mulhi_idx(w.y, (uint32_t)p.N - 1u);
    if (b >= a) b++;                                                               // np.random.choice(nchains, 2, replace=False) :185
    const double u = u53(w.z, w.w);
    const double T1 = p.Tc[a], T2 = p.Tc[b], l1 = p.llike[a], l2 = p.llike[b];
    const double alpha = ((T1 * l2) + (T2 * l1)) - ((T1 * l1) + (T2 * l2));         // :195
    const bool acc = dlog(u) < alpha;                                              // :197
    if (acc) {
        double* xa = p.X + (size_t)a * p.ld; double* xb = p.X + (size_t)b * p.ld;
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int jj = 128 * it + 2 * lane;
            if (jj < p.ld) {
                const double2 ta = *reinterpret_cast<const double2*>(xa + jj), tb = *reinterpret_cast<const double2*>(xb + jj);
                *reinterpret_cast<double2*>(xa + jj) = tb; *reinterpret_cast<double2*>(xb + jj) = ta;
                if (published) {
                    // the next generation's jumps are measured from the states astep is handed, i.e. after the swap (Dream.py:371-378,
                    // core.py:204-215): the published copy that serves as that baseline follows the exchange (the column statistics
                    // of this generation were taken before it)
                    double* ca = p.cp_new + (size_t)a * p.ld + jj; double* cb = p.cp_new + (size_t)b * p.ld + jj;
                    const double2 qa = *reinterpret_cast<const double2*>(ca), qb = *reinterpret_cast<const double2*>(cb);
                    *reinterpret_cast<double2*>(ca) = qb; *reinterpret_cast<double2*>(cb) = qa;
                }
            }
        }
        if (lane == 0) {
            const double pa = p.lprior[a], pb = p.lprior[b];
            p.llike[a] = l2; p.llike[b] = l1; p.lprior[a] = pb; p.lprior[b] = pa;
        }
    }
    if (lane == 0 && trace_slot >= 0) { int32_t* q = p.tswap + 3 * trace_slot; q[0] = (int32_t)a; q[1] = (int32_t)b; q[2] = acc ? 1 : 0; }
}

// Sequential sums with the loads of a batch in flight together: a plain `for` over dependent adds of freshly loaded values waits for
// one memory round trip per element (64 rows: 18 us, measured); the order of the additions is untouched.
constexpr int SUM_BATCH = 64;      // (a strip of 64 chains, the 64 strips of 4096 chains: one memory round trip)
template <int SQ>      // 0: sum of v; 1: sum of (v - m)^2 by fma; 2: by multiply-then-add (numpy's own order, single-chain stepping)
DZ_DEV double strided_sum(const double* __restrict__ q, size_t stride, int n, double m)
{
    double ps = 0.0;
    int c = 0;
    for (; c + SUM_BATCH <= n; c += SUM_BATCH) {
        double v[SUM_BATCH];
#pragma unroll
        for (int u = 0; u < SUM_BATCH; ++u) v[u] = q[(size_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < SUM_BATCH; ++u) {
            if (SQ == 0) ps = ps + v[u];
            else { const double t = v[u] - m; ps = SQ == 1 ? fma(t, t, ps) : ps + t * t; }
        }
    }
    for (; c < n; ++c) {
        const double x = q[(size_t)c * stride];
        if (SQ == 0) ps = ps + x;
        else { const double t = x - m; ps = SQ == 1 ? fma(t, t, ps) : ps + t * t; }
    }
    return ps;
}
#ifndef DZ_TEMPLATES_ONLY   // plain (non-template) kernels: defined once, in dz_engine.hip's translation unit
// Dream.py:281-282: `while np.all(np.isfinite(np.array(log_ps))==False)` -- marks the chains whose k tries are all impossible
// (log_ps = T log_likes + log_priors, :279); the host draws their proposal sets again (redraw_impossible_sets).
__global__ void k_redo_flags(Params p, int c0, int nc, uint8_t* __restrict__ redo)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const int c = c0 + t;
    bool any = false;
    for (int i = 0; i < p.k; ++i) any = any || is_finite(p.p_prior[c * p.k + i] + chain_T(p, c) * p.p_like[c * p.k + i]);
    redo[c] = any ? 0 : 1;
}
// the redrawn sets, packed: set j (k rows of ld doubles) of chain list[j] -> dst + j k ld, so that only they are evaluated again
__global__ void k_gather_sets(Params p, const int32_t* __restrict__ list, int n, double* __restrict__ dst)
{
    const size_t per = (size_t)p.k * p.ld / 2, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= per * n) return;
    const size_t j = t / per, r = t % per;
    reinterpret_cast<double2*>(dst + j * 2 * per)[r] = reinterpret_cast<const double2*>(p.P + (size_t)list[j] * 2 * per)[r];
}
// ... and their log densities back to the chains' slots
__global__ void k_scatter_logp(Params p, const int32_t* __restrict__ list, int n, const double* __restrict__ prior, const double* __restrict__ like)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * p.k) return;
    const int j = t / p.k, i = t % p.k, c = list[j];
    p.p_prior[c * p.k + i] = prior[t]; p.p_like[c * p.k + i] = like[t];
}

__global__ void k_draws(Params p, uint32_t g, int c0, int nc, uint4* __restrict__ out, ChainCtl* __restrict__ ctl)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc * p.nslots) return;
    const int c = c0 + t / p.nslots, slot = t % p.nslots;
    const uint32_t gc = (uint32_t)(p.off + c);
    const u32x4 w = slot_counter_draw(p, slot, gc, g);
    out[(size_t)c * p.nslots + slot] = make_uint4(w.x, w.y, w.z, w.w);
    if (slot == 0) {
        const Ctrl u = draw_ctrl(p.k0, p.k1, gc, g);
        const StepFlags f = step_flags_chain(p, u, c);
        ChainCtl o; o.snk = f.snk ? 1 : 0; o.cr_idx = f.cr_idx; o.delta = f.delta; o.glev = f.glev; o.u_sel = u.u_sel; o.u_acc = u.u_acc;
        ctl[c] = o;
    }
}

// Peer exchange (dz_peer_attach): waits until every other rank's copy engine has delivered its rows of exchange number `need` -- the
// rank's flag word, written behind the rows on the same copy stream, has reached `need`.  One wave, lane r polls rank r's flag with
// system-scope loads (the words are written by another GPU's DMA engine).  Runs as a kernel of its own IN FRONT of the kernels that read
// the rows, so that those start behind a kernel boundary (their caches hold nothing of the freshly written lines).  stats (host-mapped):
// [0] ticks of the 100 MHz clock spent waiting, [1] gates passed, [2] set to 1 + the missing rank when `timeout_ticks` run out.
__global__ __launch_bounds__(64) void k_peer_gate(const unsigned long long* __restrict__ flags, int world, int rank, unsigned long long need,
                                                   unsigned long long timeout_ticks, unsigned long long* __restrict__ stats)
{
    const int r = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    bool ok = (r >= world) || (r == rank);
    unsigned long long t1 = t0;
    while (!__all(ok)) {
        if (!ok) ok = __hip_atomic_load(flags + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= need;
        t1 = wall_clock64();
        if (t1 - t0 > timeout_ticks) break;
        __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long bad = __ballot(!ok);
    if (r == 0) {
        stats[0] += t1 - t0; stats[1] += 1ull;
        if (bad) stats[2] = 1ull + (unsigned long long)__ffsll((long long)bad) - 1ull;
    }
}

// 64-bit checksum of elements [0, rows) x [0, d) of a row-major [rows, ld] array: the sum modulo 2^64, over all elements, of
// mix64(bits(x) + (r d + j + 1) 0x9E3779B97F4A7C15), mix64 = the splitmix64 finaliser.  Addition commutes, so lanes, waves and blocks add
// in whatever order they finish; every position enters the hash, so equal sums of two replicas mean equal rows in equal places
// (dz_history_checksum: the replicated archives of a sharded run compared after the run, bench.py "replicas_identical").
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_checksum(const double* __restrict__ a, long long rows, int d, int ld, unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0ull;
    const long long nrow_stride = (long long)gridDim.x * (blockDim.x / 64);
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); r < rows; r += nrow_stride)
        for (int j = lane; j < d; j += 64) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(a[(size_t)r * ld + j]);
            acc += mix64(bits + ((unsigned long long)r * (unsigned long long)d + (unsigned long long)j + 1ull) * 0x9E3779B97F4A7C15ull);
        }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) atomicAdd(out, acc);
}

// copy rows [nl,ld] (used for publishing start positions and the sharded exchange staging)
__global__ void k_copy_rows(const double* __restrict__ src, double* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ------------------------------------------------------------------------------------------
// Adaptation during burn-in: estimate_crossover_probabilities :451-499,
// estimate_gamma_level_probs :501-540, on the replicated positions of ALL N chains.
// np.std(axis=0) :476 in two-level order: strips of 64 rows, then strips in order.
// ------------------------------------------------------------------------------------------
// pass 0: partial[s][j] = sum rows of strip s; pass 1: sum of squared deviations from mean[j]
// strip = 64 (lockstep generations); single-chain stepping (schedule S1) sums all N rows in row order, squares by multiply-then-add --
// numpy's own order for np.std(axis=0): strip = N, plain = 1
__global__ void k_strip_partial(const double* __restrict__ pos, int N, int d, int ld, const double* __restrict__ mean, int pass, double* __restrict__ partial, int strip, int plain)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= d) return;
    const int r0 = s * strip, r1 = min(N, r0 + strip);
    const double* q = pos + (size_t)r0 * ld + j;
    double ps;
    if (pass == 0) ps = strided_sum<0>(q, (size_t)ld, r1 - r0, 0.0);
    else if (plain) ps = strided_sum<2>(q, (size_t)ld, r1 - r0, mean[j]);
    else ps = strided_sum<1>(q, (size_t)ld, r1 - r0, mean[j]);
    partial[(size_t)s * ld + j] = ps;
}
// single-chain stepping: the chains of [c0, c0 + nc) that have just updated the shared probabilities adopt them as their own copy
// (Dream.py:375 / :383 through :497 / :538; at the end of the burn-in :409-415 -- there binc / bing are set as well); init != 0: all of
// the range's chains take the shared vectors (a chain's first step, Dream.py:134 / :143)
__global__ void k_own_probs(Params p, int c0, int nc, const int* __restrict__ binc, const int* __restrict__ bing, int init, double* __restrict__ own_cr, double* __restrict__ own_g)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nc) return;
    const int c = c0 + t, gcn = p.off + c;
    if (init || binc[gcn] >= 0) for (int m = 0; m < p.ncr; ++m) own_cr[(size_t)c * p.ncr + m] = p.cr_probs[m];
    if (init || bing[gcn] >= 0) for (int m = 0; m < p.ngamma; ++m) own_g[(size_t)c * p.ngamma + m] = p.g_probs[m];
}
// The column means and pass 1 in one launch: every block makes the means it needs itself -- mean[j] = (sum_s partial0[s][j]) / N, the strips
// in order --, then its strip's sum of squared deviations
__global__ void k_strip_dev(const double* __restrict__ pos, int N, int d, int ld, const double* __restrict__ partial0, int nstrips, double* __restrict__ partial1,
                            double* __restrict__ mean_out, int strip, int plain)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= d) return;
    const double tot = strided_sum<0>(partial0 + j, (size_t)ld, nstrips, 0.0);
    const double m = tot / (double)N;
    if (s == 0) mean_out[j] = m;
    const int r0 = s * strip, r1 = min(N, r0 + strip);
    const double* q = pos + (size_t)r0 * ld + j;
    partial1[(size_t)s * ld + j] = plain ? strided_sum<2>(q, (size_t)ld, r1 - r0, m) : strided_sum<1>(q, (size_t)ld, r1 - r0, m);
}
#endif  // DZ_TEMPLATES_ONLY

// Bins and normalised squared jumps (:481, :527): one wave per GLOBAL chain, 16 chains per block, of which [gc0, gc0 + ngc) take part
// (everything in a lockstep generation; one chain under Dream.astep).  The standard deviations come from the strip sums of pass 1, added
// in strip order by every block for itself (block 0 also stores them): sd[j] = sqrt((sum_s partial1[s][j]) / N),
// for the crossover statistic with 0 -> 1e-12 (:479).  The chain's control draws and the gamma-unity draws of its last proposal call are
// made lane-parallel (lane s = slot s: ONE Philox call per wave instead of 2 + k).
constexpr int JUMP_CHAINS = 16;
template <int NCH>
__global__ __launch_bounds__(1024) void k_jump(Params p, uint32_t g, int gc0, int ngc, const double* __restrict__ partial1, int nstrips,
                                               double* __restrict__ sd_out, double* __restrict__ sdc_out,
                                               double* __restrict__ dl, double* __restrict__ dlg, int* __restrict__ binc, int* __restrict__ bing)
{
    __shared__ double s_sdc[128 * NCH], s_sdg[128 * NCH];
    for (int j = threadIdx.x; j < p.d; j += 1024) {
        const double tot = strided_sum<0>(partial1 + j, (size_t)p.ld, nstrips, 0.0);
        const double v = sqrt(tot / (double)p.N);
        s_sdg[j] = v; s_sdc[j] = v == 0.0 ? 1e-12 : v;
        if (blockIdx.x == 0) { sd_out[j] = v; sdc_out[j] = v == 0.0 ? 1e-12 : v; }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int gcn = blockIdx.x * JUMP_CHAINS + (threadIdx.x >> 6);
    if (gcn >= p.N) return;
    if (gcn < gc0 || gcn >= gc0 + ngc) {                                   // (Dream.astep: every other chain contributes nothing this time)
        if (lane == 0) { binc[gcn] = -1; bing[gcn] = -1; }
        return;
    }
    // lanes 0, 1: control stream idx 0, 1 (set_snooker / set_CR, set_DEpair / set_gamma_level); lanes 2 .. 2 + n - 1: the gamma-unity draws
    // of the tries of the LAST generate_proposal_points call (:371, :705 / :730: the reference set's, or the single try's)
    const int phase = p.k > 1 ? 1 : 0, n = p.k > 1 ? p.k - 1 : 1;
    u32x4 w = u32x4{0, 0, 0, 0};
    if (lane < 2 + n) w = lane < 2 ? philox(p.k0, p.k1, (uint32_t)lane, stream_id(K_CTRL, 0, 0), (uint32_t)gcn, g)
                                   : philox(p.k0, p.k1, 0u, stream_id(K_PT, (uint32_t)(lane - 2), (uint32_t)phase), (uint32_t)gcn, g);
    Ctrl u;
    u.u_snk = u53(__shfl(w.x, 0, 64), __shfl(w.y, 0, 64)); u.u_cr = u53(__shfl(w.z, 0, 64), __shfl(w.w, 0, 64));
    u.u_de = u53(__shfl(w.x, 1, 64), __shfl(w.y, 1, 64)); u.u_glev = u53(__shfl(w.z, 1, 64), __shfl(w.w, 1, 64));
    u.u_sel = 0.0; u.u_acc = 0.0;
    const StepFlags f = step_flags_chain(p, u, gcn - p.off);              // (own copies exist only on an unsharded engine: local == global)
    const bool gu = !f.snk && __any(lane >= 2 && lane < 2 + n && u53(w.x, w.y) < p.pgu);
    const bool at_end = (int)g == p.burnin;
    const bool window = g > 10 && (int)g < p.burnin;
    const bool do_c = p.adapt_cr && (at_end || (window && !gu));               // :371, :395
    const bool do_g = p.adapt_g && (at_end || (window && !gu && !f.snk));      // :381, :391
    double accC = 0.0, accG = 0.0;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        if (jj < p.ld) {
            const double2 a = *reinterpret_cast<const double2*>(p.cp_new + (size_t)gcn * p.ld + jj);
            const double2 b = *reinterpret_cast<const double2*>(p.cp_prev + (size_t)gcn * p.ld + jj);
#pragma unroll
            for (int s = 0; s < 2; ++s) if (jj + s < p.d) {
                const double df = (s ? a.y : a.x) - (s ? b.y : b.x);
                const double t = df / s_sdc[jj + s]; accC = fma(t, t, accC);
                const double t2 = df / s_sdg[jj + s]; accG = fma(t2, t2, accG);
            }
        }
    }
    const double CG = wave_bfly2(accC, accG);
    const double dC = nan_to_num(readlane_f64(CG, 0)), dG = nan_to_num(readlane_f64(CG, 32));
    if (lane == 0) {
        binc[gcn] = do_c ? (f.snk ? p.ncr - 1 : f.cr_idx) : -1;               // :374-378
        bing[gcn] = do_g ? f.glev - 1 : -1;
        dl[gcn] = dC; dlg[gcn] = dG;
    }
}

// ------------------------------------------------------------------------------------------
// Adaptation of a LOCKSTEP generation, reduction contract v3 (oracle: adapt_lockstep).
//   delta_m[m] += sum_j D[m][j] / sd_j^2,  D[m][j] = sum over the chains of bin m of (x_new - x_prev)_j^2:
// the per-bin column sums of squared jumps do not need sd, so they are taken in the same pass over the positions as the sums sd
// itself needs (S1_j = sum (x - s_j), S2_j = sum (x - s_j)^2 around the shift s = previous position of global chain 0) -- ONE
// grid-wide reduction where the chain-by-chain form (k_strip_* / k_jump / k_adapt_update, kept for Dream.astep's schedule S1)
// needs three.  Sums over chains: units of 16 consecutive global chains in chain order from 0.0 (adapt_unit_sums: one block of
// the persistent kernels, or k_adapt_partials), groups of 16 units in order, then the groups in order (k_adapt_totals);
// k_adapt_apply makes sd, the weights 1 / sd^2, the nb dot products and the new probabilities.
// Partial rows: PR[unit][q][ld], q = 0: S1, 1: S2, 2 + m: crossover bin m, 2 + ncr + m: gamma-level bin m; PC[unit][ncr + ngamma]: counts.
// ------------------------------------------------------------------------------------------
DZ_DEV int adapt_nq(const Params& p) { return 2 + p.ncr + p.ngamma; }

// the bins of GLOBAL chain gcn at generation g (:371-383, :385-401), by its wave: lanes 0, 1 make the control stream's idx 0, 1
// (set_snooker / set_CR, set_DEpair / set_gamma_level), lanes 2 .. 2 + n - 1 the gamma-unity draws of the tries of the LAST
// generate_proposal_points call (:705 / :730: the reference set's, or the single try's) -- ONE Philox call per wave
DZ_DEV void adapt_bins(const Params& p, uint32_t g, int gcn, int lane, int& binc, int& bing, const double* cr_probs, const double* g_probs)
{
    const int phase = p.k > 1 ? 1 : 0, n = p.k > 1 ? p.k - 1 : 1;
    u32x4 w = u32x4{0, 0, 0, 0};
    if (lane < 2 + n) w = lane < 2 ? philox(p.k0, p.k1, (uint32_t)lane, stream_id(K_CTRL, 0, 0), (uint32_t)gcn, g)
                                   : philox(p.k0, p.k1, 0u, stream_id(K_PT, (uint32_t)(lane - 2), (uint32_t)phase), (uint32_t)gcn, g);
    Ctrl u;
    u.u_snk = u53(__shfl(w.x, 0, 64), __shfl(w.y, 0, 64)); u.u_cr = u53(__shfl(w.z, 0, 64), __shfl(w.w, 0, 64));
    u.u_de = u53(__shfl(w.x, 1, 64), __shfl(w.y, 1, 64)); u.u_glev = u53(__shfl(w.z, 1, 64), __shfl(w.w, 1, 64));
    u.u_sel = 0.0; u.u_acc = 0.0;
    const StepFlags f = step_flags_from(p, u, cr_probs, g_probs);          // (lockstep: every chain decides with the shared probabilities)
    const bool gu = !f.snk && __any(lane >= 2 && lane < 2 + n && u53(w.x, w.y) < p.pgu);
    const bool at_end = (int)g == p.burnin;
    const bool window = g > 10 && (int)g < p.burnin;
    const bool do_c = p.adapt_cr && (at_end || (window && !gu));               // :371, :395
    const bool do_g = p.adapt_g && (at_end || (window && !gu && !f.snk));      // :381, :391
    binc = do_c ? (f.snk ? p.ncr - 1 : f.cr_idx) : -1;                         // :374-378
    bing = do_g ? f.glev - 1 : -1;
}

// The sums of ONE unit (16 consecutive global chains, nc of them present) by the threads of a block: xn / xp = the chains' new /
// previous positions, [16][ldx] / [16][ldxp] (LDS or global), bin(isg, c) = chain c's crossover (isg = false) / gamma-level bin or -1.
// Thread (q, j) adds its column's terms in chain order.
template <class BIN>
DZ_DEV void adapt_unit_sums(const Params& p, const double* xn, int ldx, const double* xp, int ldxp, int nc, BIN bin, const double* __restrict__ shift,
                            double* __restrict__ pr /* [nq][ld] */, double* __restrict__ pc /* [ncr + ngamma] */, int tid, int nthreads)
{
    const int d = p.d, nq = adapt_nq(p);
    for (int idx = tid; idx < nq * d; idx += nthreads) {
        const int q = idx / d, j = idx - q * d;
        double us = 0.0;
        if (q < 2) {
            const double sj = shift[j];
            for (int c = 0; c < nc; ++c) { const double v = xn[(size_t)c * ldx + j] - sj; us = q == 0 ? us + v : fma(v, v, us); }
        } else {
            const int m = q - 2; const bool isg = m >= p.ncr; const int mm = isg ? m - p.ncr : m;
            for (int c = 0; c < nc; ++c) if (bin(isg, c) == mm) { const double df = xn[(size_t)c * ldx + j] - xp[(size_t)c * ldxp + j]; us = fma(df, df, us); }
        }
        pr[(size_t)q * p.ld + j] = us;
    }
    if (tid < p.ncr + p.ngamma) {
        const bool isg = tid >= p.ncr; const int mm = isg ? tid - p.ncr : tid;
        int cnt = 0;
        for (int c = 0; c < nc; ++c) cnt += bin(isg, c) == mm ? 1 : 0;
        pc[tid] = (double)cnt;
    }
}

// sd, the weights 1 / sd^2, the bins' dot products and the new probabilities (:476-493, :522-536) from the totals, by ONE wave: lane b keeps
// bin b's accumulators (crossover bins first, then the gamma-level bins).  Two halves (round 6: with an adapt_lag the first is made right behind
// the generation, the second when the update is due):
//   adapt_dots_wave: sd and the weights from the column totals, then per bin b with a count the dot product sum_j D[b][j] w_j (the lane's
//   dimensions by fma in ascending j, the 64 partial sums by the xor butterfly), nan_to_num -- lane b returns bin b's, others 0;
//   adapt_scalar_step: delta_m[b] += dot, ncr_updates[b] += count, and the renormalisation of a kind whose bins got anything once every one
//   of its bins has a non-zero delta (:487-493 / :531-536) -- lane b's (delta, n, probability) in registers.
template <int NCH>
DZ_DEV double adapt_dots_wave(const Params& p, const double* __restrict__ TOT, double my_cnt, int lane)
{
    const int d = p.d, ld = p.ld, ncr = p.ncr, nb = ncr + p.ngamma;
    const double Nd = (double)p.N;
    double wc[NCH][2], wg[NCH][2];
#pragma unroll
    for (int it = 0; it < NCH; ++it)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int j = 128 * it + 2 * lane + s;
            wc[it][s] = 0.0; wg[it][s] = 0.0;
            if (j < d) {
                const double a = TOT[j] / Nd;
                double var = fma(-a, a, TOT[(size_t)ld + j] / Nd);
                if (!(var > 0.0)) var = 0.0;
                const double sd = sqrt(var), sdc = sd == 0.0 ? 1e-12 : sd;      // :479 (crossover only)
                wc[it][s] = 1.0 / (sdc * sdc); wg[it][s] = 1.0 / (sd * sd);
            }
        }
    double mine = 0.0;
    for (int b = 0; b < nb; ++b) {
        const double cnt = readlane_f64(my_cnt, b);
        if (!(cnt > 0.0)) continue;                                           // (wave-uniform)
        const bool isg = b >= ncr;
        const double* D = TOT + (size_t)(2 + b) * ld;
        double acc = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int j = 128 * it + 2 * lane + s;
                if (j < d) acc = fma(D[j], isg ? wg[it][s] : wc[it][s], acc);
            }
        const double tot = nan_to_num(wave_bfly(acc));
        if (lane == b) mine = tot;
    }
    return mine;
}
// lane b < nb: bin b's dot product and count (0 where the bin got nothing) into its accumulators; the new probability
DZ_DEV void adapt_scalar_step(const Params& p, double my_dot, double my_cnt, double& my_delta, double& my_n, double& my_prob, int lane)
{
    const int ncr = p.ncr, nb = ncr + p.ngamma;
    const double Nd = (double)p.N;
    const bool mine = lane < nb, isg_l = lane >= ncr;
    if (mine && my_cnt > 0.0) { my_delta = my_delta + my_dot; my_n += my_cnt; }
    const bool anyc = __any(lane < ncr && my_cnt > 0.0), anyg = __any(mine && isg_l && my_cnt > 0.0);
    const bool allc = !__any(lane < ncr && my_delta == 0.0), allg = !__any(mine && isg_l && my_delta == 0.0);
    const double pm = (my_delta / my_n) * Nd;
    double Sc = 0.0, Sg = 0.0;
    for (int m = 0; m < ncr; ++m) Sc = Sc + readlane_f64(pm, m);
    for (int m = ncr; m < nb; ++m) Sg = Sg + readlane_f64(pm, m);
    const bool renorm = isg_l ? (anyg && allg) : (anyc && allc);
    if (renorm) my_prob = pm / (isg_l ? Sg : Sc);
}
// offsets of bin b's probability / delta / count in the shared state cr_probs|cr_delta|cr_n|g_probs|g_delta|g_n
DZ_DEV void adapt_state_offsets(int ncr, int ng, int lane, int& o_probs, int& o_delta, int& o_n)
{
    const bool isg_l = lane >= ncr; const int m_l = isg_l ? lane - ncr : lane;
    o_probs = isg_l ? 3 * ncr + m_l : m_l; o_delta = isg_l ? 3 * ncr + ng + m_l : ncr + m_l; o_n = isg_l ? 3 * ncr + 2 * ng + m_l : 2 * ncr + m_l;
}
// Both halves at once (adapt_lag 0: the previous generation's totals, applied by the next launch's prologue or by k_adapt_apply).  sh_in: the
// state before (cr_probs|cr_delta|cr_n|g_probs|g_delta|g_n); probs_out: [ncr + ngamma] the new probabilities (LDS of a persistent kernel) or
// null; sh_out: the whole new state or null (may be sh_in).
template <int NCH>
DZ_DEV void adapt_apply_wave(const Params& p, const double* __restrict__ TOT, const double* __restrict__ CNT, const double* sh_in, double* probs_out, double* sh_out, int lane)
{
    const int ncr = p.ncr, ng = p.ngamma, nb = ncr + ng;
    const bool mine = lane < nb;
    int o_probs, o_delta, o_n;
    adapt_state_offsets(ncr, ng, lane, o_probs, o_delta, o_n);
    double my_delta = mine ? sh_in[o_delta] : 1.0, my_n = mine ? sh_in[o_n] : 1.0;
    double my_prob = mine ? sh_in[o_probs] : 0.0;
    const double my_cnt = mine ? CNT[lane] : 0.0;
    const double my_dot = adapt_dots_wave<NCH>(p, TOT, my_cnt, lane);
    adapt_scalar_step(p, my_dot, my_cnt, my_delta, my_n, my_prob, lane);
    if (mine) {
        if (probs_out) probs_out[lane] = my_prob;
        if (sh_out) { sh_out[o_probs] = my_prob; sh_out[o_delta] = my_delta; sh_out[o_n] = my_n; }
    }
}
// adapt_lag >= 1: the pending updates of generations [pend0, pend1) -- dot products and counts in ring slots of `R1` -- applied in order by one
// wave while it walks the n generations g0 .. g0 + n - 1 the caller is about to run: generation g decides with everything through generation
// g - 1 - lag (everything there is once g > burnin: the hand-over).  probs_tab: [n][nbp] the probabilities of each of them (LDS) or null;
// sh_out: the state after the last of them or null (may be sh_in).  Returns nothing; every lane ends with the same walk.
DZ_DEV void adapt_pending_apply(const Params& p, const double* __restrict__ DOT, const double* __restrict__ CNTR, int nbp, int R1, long long pend0, long long pend1,
                                long long g0, int n, int lag, int burnin, const double* sh_in, double* probs_tab, double* sh_out, int lane)
{
    const int ncr = p.ncr, ng = p.ngamma, nb = ncr + ng;
    const bool mine = lane < nb;
    int o_probs, o_delta, o_n;
    adapt_state_offsets(ncr, ng, lane, o_probs, o_delta, o_n);
    double my_delta = mine ? sh_in[o_delta] : 1.0, my_n = mine ? sh_in[o_n] : 1.0;
    double my_prob = mine ? sh_in[o_probs] : 0.0;
    long long next = pend0;
    for (int i = 0; i < n; ++i) {
        const long long g = g0 + i;
        const long long through = g > (long long)burnin ? pend1 - 1 : g - 1 - lag;
        while (next < pend1 && next <= through) {
            const int slot = (int)(next % R1);
            const double cnt = mine ? CNTR[(size_t)slot * nbp + lane] : 0.0, dot = mine ? DOT[(size_t)slot * nbp + lane] : 0.0;
            adapt_scalar_step(p, dot, cnt, my_delta, my_n, my_prob, lane);
            ++next;
        }
        if (probs_tab && mine) probs_tab[(size_t)i * nbp + lane] = my_prob;
    }
    if (sh_out && mine) { sh_out[o_probs] = my_prob; sh_out[o_delta] = my_delta; sh_out[o_n] = my_n; }
}

#ifndef DZ_TEMPLATES_ONLY   // plain (non-template) kernels: defined once, in dz_engine.hip's translation unit
// The general form: the units' sums from the published positions (the multi-kernel path, blocks of fewer than 16 chains, and sharded runs
// whose ranks do not own whole groups: then from the replicated positions of ALL N chains); block = one unit, wave w = global chain
// 16 unit + w (its bins), then thread (q, j) the sums.  unit0: the first GLOBAL unit of this launch (a rank that owns whole groups makes
// only its own units' sums: rows PR[0 ..] = units unit0 ..); shift: the row the column sums are taken around -- the previous published
// position of global chain 0 -- or null: row 0 of p.cp_prev.
__global__ __launch_bounds__(1024) void k_adapt_partials(Params p, uint32_t g, double* __restrict__ PR, double* __restrict__ PC, int unit0, const double* __restrict__ shift)
{
    __shared__ int s_bc[16], s_bg[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, unit = unit0 + blockIdx.x;
    const int gcn = 16 * unit + wv;
    int bc = -1, bg = -1;
    if (gcn < p.N) adapt_bins(p, g, gcn, lane, bc, bg, p.cr_probs, p.g_probs);
    if (lane == 0) { s_bc[wv] = bc; s_bg[wv] = bg; }
    __syncthreads();
    const int nc = min(16, p.N - 16 * unit), nq = adapt_nq(p);
    adapt_unit_sums(p, p.cp_new + (size_t)16 * unit * p.ld, p.ld, p.cp_prev + (size_t)16 * unit * p.ld, p.ld, nc, [&](bool isg, int c) { return isg ? s_bg[c] : s_bc[c]; }, shift ? shift : p.cp_prev,
                    PR + (size_t)blockIdx.x * nq * p.ld, PC + (size_t)blockIdx.x * (p.ncr + p.ngamma), threadIdx.x, 1024);
}

// adapt_lag >= 1, several burn-in generations per launch by a kernel that does not make its blocks' unit sums (Publish::multi == 2: blocks of 12, 8
// or 4 chains, split generations, 128 < d, multitry off): the sums of the launch's n generations g0 .. g0 + n - 1 from the RING of published
// positions -- block (unit, i): generation g = g0 + i, new positions = slot g mod (R1 + 1), previous = slot (g - 1) mod (R1 + 1) (generation
// -1: the start positions, seeded by the host), the bins from the probabilities generation g decided with (PG[i], left by the launch's block
// 0), the shift = global chain 0's position after generation g - R1 -- into the rings' slot g mod R1.  Unit 0 also leaves chain 0's new
// position in x0ring (slot g mod 2 R1: no generation of this launch reads a slot this launch writes).
__global__ __launch_bounds__(1024) void k_adapt_partials_ring(Params p, uint32_t g0, int R1, const double* __restrict__ pos, long long pos_stride, const double* __restrict__ PG, int nbp,
                                                              double* __restrict__ PR, double* __restrict__ PC, long long pr_stride, long long pc_stride,
                                                              double* __restrict__ x0ring, const double* __restrict__ x0start, int unit0 = 0)
{
    // (unit0: sharded -- this rank's first GLOBAL unit; its units' sums go to rows 0 .. of the ring slot, as k_adapt_partials leaves them)
    __shared__ int s_bc[16], s_bg[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, unit = unit0 + blockIdx.x;
    const uint32_t g = g0 + blockIdx.y;
    const int R = R1 + 1;
    const int gcn = 16 * unit + wv;
    const double* pr_g = PG + (size_t)blockIdx.y * nbp;
    int bc = -1, bg = -1;
    if (gcn < p.N) adapt_bins(p, g, gcn, lane, bc, bg, pr_g, pr_g + p.ncr);
    if (lane == 0) { s_bc[wv] = bc; s_bg[wv] = bg; }
    __syncthreads();
    const int nc = min(16, p.N - 16 * unit), nq = adapt_nq(p);
    const double* xn = pos + (size_t)(g % (uint32_t)R) * pos_stride + (size_t)16 * unit * p.ld;
    const double* xp = pos + (size_t)((g + (uint32_t)R - 1u) % (uint32_t)R) * pos_stride + (size_t)16 * unit * p.ld;
    const long long hs = (long long)g - (long long)R1;
    const double* shift = hs < 0 ? x0start : x0ring + (size_t)(hs % (2 * R1)) * p.ld;
    const int slot = (int)(g % (uint32_t)R1);
    adapt_unit_sums(p, xn, p.ld, xp, p.ld, nc, [&](bool isg, int c) { return isg ? s_bg[c] : s_bc[c]; }, shift,
                    PR + (size_t)slot * pr_stride + (size_t)blockIdx.x * nq * p.ld, PC + (size_t)slot * pc_stride + (size_t)blockIdx.x * (p.ncr + p.ngamma), threadIdx.x, 1024);
    if (unit == 0)
        for (int j = threadIdx.x; j < p.ld; j += 1024) x0ring[(size_t)(g % (uint32_t)(2 * R1)) * p.ld + j] = xn[j];
}

// ---- sharded crossover burn-in (round 5): the ranks exchange their GROUPS' sums, not their positions.  Contract v3 adds the units' sums
// in groups of 16 units (256 consecutive global chains) in order, then the groups in order; a rank that owns whole groups (chain offset and
// local chain count multiples of 256) makes its groups' sums from its own units' sums -- the same additions in the same order as
// k_adapt_totals' middle stage -- and every rank then adds ALL groups in order from the exchanged records: the same totals, bit for bit, at
// every world size, for (2 + nCR + ngamma) ld + 16 doubles per group instead of 256 ld per group of positions (4096 chains per GPU,
// 100-D, nCR = 3: 86 KB per rank and generation instead of 3.7 MB).  (What replaces the shared current_positions array of core.py:296 /
// Dream.py:447-449 under sharding; Dream.py:451-499 is the arithmetic.)
// A rank's record (rec doubles): [groups][nq][ld] sums | [groups][nbp] bin counts (nbp = nb rounded up to 16) | [ld] the new position of
// the rank's first chain (rank 0's is global chain 0's: the shift of the NEXT generation's column sums).
// thread (group, column): the group's units in order from 0.0
// (adapt_lag >= 1, a sharded launch of several burn-in generations: grid.y = the launch's generations g0 .. -- generation g0 + i reads the units' sums
//  of ring slot (g0 + i) mod R1 (strides pr_stride / pc_stride), takes x0 from slot (g0 + i) mod x0_R of the ring `x0` (x0_R = 0: `x0` is the row) and
//  writes record i of this rank, rec + i rec_stride)
__global__ __launch_bounds__(256) void k_adapt_groups(const double* __restrict__ PR, const double* __restrict__ PC, int nunits, int nq, int d, int ld, int nb, int nbp,
                                                      const double* __restrict__ x0, double* __restrict__ rec,
                                                      long long g0 = 0, int R1 = 1, long long pr_stride = 0, long long pc_stride = 0, long long rec_stride = 0, int x0_R = 0)
{
    {
        const long long g = g0 + blockIdx.y;
        PR += (size_t)(g % R1) * pr_stride; PC += (size_t)(g % R1) * pc_stride; rec += (size_t)blockIdx.y * rec_stride;
        if (x0_R) x0 += (size_t)(g % x0_R) * ld;
    }
    const int ngroups = (nunits + 15) / 16, ncol = nq * d;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const size_t ustride = (size_t)nq * ld;
    if (idx < ngroups * ncol) {
        const int gq = idx / ncol, col = idx - gq * ncol, q = col / d, j = col - q * d;
        const double* src = PR + (size_t)16 * gq * ustride + (size_t)q * ld + j;
        const int nu = min(16, nunits - 16 * gq);
        double v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = i < nu ? src[(size_t)i * ustride] : 0.0;
        double gs = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i < nu) gs = gs + v[i];
        rec[(size_t)gq * ustride + (size_t)q * ld + j] = gs;
        return;
    }
    int r = idx - ngroups * ncol;
    if (r < ngroups * nbp) {           // the bins' counts (small integers: exact)
        const int gq = r / nbp, b = r - gq * nbp;
        double c = 0.0;
        if (b < nb) for (int i = 0; i < 16 && 16 * gq + i < nunits; ++i) c += PC[(size_t)(16 * gq + i) * nb + b];
        rec[(size_t)ngroups * ustride + (size_t)gq * nbp + b] = c;
        return;
    }
    r -= ngroups * nbp;
    if (r < ld) rec[(size_t)ngroups * ustride + (size_t)ngroups * nbp + r] = x0[r];
}
// ... and every rank's totals from all ranks' records: thread column adds the groups in global order (rank by rank) from 0.0 -- k_adapt_totals'
// last stage --; the counts; the next generation's shift (rank 0's first chain)
// (grid.y = the launch's generations: generation g0 + i reads record i of every rank -- GS + i gen_stride, the ranks `rec` apart -- and writes the totals'
//  ring slot (g0 + i) mod R1 and chain 0's position into slot (g0 + i) mod x0_R of the ring `shift_out`; x0_R = 0: the pointers as they are)
__global__ __launch_bounds__(256) void k_group_totals(const double* __restrict__ GS, int world, size_t rec, int gl, int nq, int d, int ld, int nb, int nbp,
                                                      double* __restrict__ TOT, double* __restrict__ CNT, double* __restrict__ shift_out,
                                                      long long g0 = 0, int R1 = 1, long long gen_stride = 0, long long tot_stride = 0, long long cnt_stride = 0, int x0_R = 0)
{
    {
        const long long g = g0 + blockIdx.y;
        GS += (size_t)blockIdx.y * gen_stride; TOT += (size_t)(g % R1) * tot_stride; CNT += (size_t)(g % R1) * cnt_stride;
        if (x0_R) shift_out += (size_t)(g % x0_R) * ld;
    }
    const int ncol = nq * d;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const size_t ustride = (size_t)nq * ld;
    if (idx < ncol) {
        const int q = idx / d, j = idx - q * d;
        double t = 0.0;
        for (int r = 0; r < world; ++r) {
            const double* src = GS + (size_t)r * rec + (size_t)q * ld + j;
            for (int g0 = 0; g0 < gl; g0 += 16) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = g0 + i < gl ? src[(size_t)(g0 + i) * ustride] : 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) if (g0 + i < gl) t = t + v[i];
            }
        }
        TOT[(size_t)q * ld + j] = t;
        return;
    }
    int r2 = idx - ncol;
    if (r2 < nb) {
        double c = 0.0;
        for (int r = 0; r < world; ++r) for (int gq = 0; gq < gl; ++gq) c += GS[(size_t)r * rec + (size_t)gl * ustride + (size_t)gq * nbp + r2];
        CNT[r2] = c;
        return;
    }
    r2 -= nb;
    if (r2 < ld) shift_out[r2] = GS[(size_t)gl * ustride + (size_t)gl * nbp + r2];
}

// Single block of 1024 threads.  Step 1: the chains' jumps and bins are staged in LDS with coalesced loads, 4096 chains at a time (a
// thread reading its strip straight from memory makes 128 line requests of its own: 34 us for 4096 chains, measured), and thread
// (strip s, bin b) adds the jumps of the strip's 64 chains that fell into the bin, in chain order (crossover bins first, then the
// gamma-level bins) -- the contract's inner sum; step 2: thread b adds the strips in order, updates delta / n of its bin; step 3: the
// probabilities are renormalised (:487-493 / :531-536).
constexpr int ADAPT_SM = 1024;
constexpr int ADAPT_CHUNK = 4096;       // chains staged per round: 4096 (8 + 8 + 4 + 4) bytes = 96 KB of LDS
__global__ __launch_bounds__(1024) void k_adapt_update(Params p, const double* __restrict__ dl, const double* __restrict__ dlg, const int* __restrict__ binc, const int* __restrict__ bing,
                                                       double* __restrict__ sm)      // scratch: [nstrips][nb] sums, then [nstrips][nb] counts (as doubles: exact)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int PADDED = ADAPT_CHUNK + ADAPT_CHUNK / 64;          // a strip's 64 values, then one pad: the strips start in different banks
    double* s_dl = lds; double* s_dlg = lds + PADDED;
    int* s_bc = reinterpret_cast<int*>(lds + 2 * PADDED); int* s_bg = s_bc + PADDED;
    __shared__ int any[2];
    __shared__ double sm_lds[2 * ADAPT_SM];        // the strips' sums and counts stay in LDS when they fit (4096 chains x 4 bins: 512 doubles)
    const int t = threadIdx.x, nb = p.ncr + p.ngamma, nstrips = (p.N + 63) / 64;
    if (2 * nstrips * nb <= 2 * ADAPT_SM) sm = sm_lds;
    if (t < 2) any[t] = 0;
    for (int c0 = 0; c0 < p.N; c0 += ADAPT_CHUNK) {
        const int cn = min(ADAPT_CHUNK, p.N - c0);
        __syncthreads();
        for (int i = t; i < cn; i += 1024) { const int q = i + (i >> 6); s_dl[q] = dl[c0 + i]; s_dlg[q] = dlg[c0 + i]; s_bc[q] = binc[c0 + i]; s_bg[q] = bing[c0 + i]; }
        __syncthreads();
        const int ns = (cn + 63) / 64;
        for (int w = t; w < ns * nb; w += 1024) {
            const int s = w / nb, b = w - s * nb;
            const bool isg = b >= p.ncr; const int m = isg ? b - p.ncr : b;
            const double* dd = (isg ? s_dlg : s_dl) + s * 65; const int* bb = (isg ? s_bg : s_bc) + s * 65;
            const int n64 = min(64, cn - s * 64);
            double ps = 0.0; int cnt = 0;
            if (n64 == 64) {
#pragma unroll 16
                for (int c = 0; c < 64; ++c) { const bool in = bb[c] == m; ps = in ? ps + dd[c] : ps; cnt += in ? 1 : 0; }
            } else for (int c = 0; c < n64; ++c) if (bb[c] == m) { ps = ps + dd[c]; cnt++; }
            const int gs = c0 / 64 + s;
            sm[(size_t)gs * nb + b] = ps; sm[(size_t)nstrips * nb + (size_t)gs * nb + b] = (double)cnt;
        }
    }
    __threadfence_block();
    __syncthreads();
    if (t < nb) {
        const bool isg = t >= p.ncr; const int m = isg ? t - p.ncr : t;
        double tot, cnt;
        if (sm == sm_lds) {
            tot = 0.0; cnt = 0.0;
            for (int s2 = 0; s2 < nstrips; ++s2) { tot = tot + sm[s2 * nb + t]; cnt += sm[nstrips * nb + s2 * nb + t]; }
        } else {
            tot = strided_sum<0>(sm + t, (size_t)nb, nstrips, 0.0);                               // (the strips in order; loads in batches)
            cnt = strided_sum<0>(sm + (size_t)nstrips * nb + t, (size_t)nb, nstrips, 0.0);       // (small integers: exact)
        }
        if (cnt > 0.0) {
            double* delta = isg ? p.g_delta : p.cr_delta; double* n = isg ? p.g_n : p.cr_n;
            delta[m] = delta[m] + tot; n[m] += cnt; atomicOr(&any[isg ? 1 : 0], 1);
        }
    }
    __syncthreads();
    if (t < 2 && any[t]) {     // :487-493 / :531-536
        const int nbb = t ? p.ngamma : p.ncr;
        double* probs = t ? p.g_probs : p.cr_probs; const double* delta = t ? p.g_delta : p.cr_delta; const double* n = t ? p.g_n : p.cr_n;
        bool all = true;
        for (int m = 0; m < nbb; ++m) if (delta[m] == 0.0) all = false;
        if (all) {
            double S = 0.0;
            for (int m = 0; m < nbb; ++m) { probs[m] = (delta[m] / n[m]) * (double)p.N; S = S + probs[m]; }
            for (int m = 0; m < nbb; ++m) probs[m] = probs[m] / S;
        }
    }
}

// Totals of the units' sums.  Block = 16 columns (q, j) of the nq d (one 128-byte segment of every unit's row: 32 blocks pull the
// 1 MB of 256 units x 500 columns through 32 CUs' paths to L2); thread (i, column) fetches unit i of every group (16 loads in flight),
// thread (group, column) adds the group's 16 units in order, thread column adds the groups in order.  One more block adds the counts.
// (round 6, adapt_lag >= 1) blockIdx.y = the i-th generation of a launch: generation g0 + i's sums and totals live in ring slot (g0 + i) mod R1
// (the strides between slots in doubles; R1 = 1, one row of blocks: the single set of buffers of adapt_lag 0)
__global__ __launch_bounds__(256) void k_adapt_totals(const double* __restrict__ PR, const double* __restrict__ PC, int nunits, int nq, int d, int ld, int nb,
                                                      double* __restrict__ TOT /* [nq][ld] */, double* __restrict__ CNT /* [nb] */,
                                                      long long g0, int R1, long long pr_stride, long long pc_stride, long long tot_stride, int cnt_stride)
{
    __shared__ double s_v[16][16][17];
    __shared__ double s_g[16][17];
    {
        const int slot = (int)((g0 + (long long)blockIdx.y) % R1);
        PR += (size_t)slot * pr_stride; PC += (size_t)slot * pc_stride; TOT += (size_t)slot * tot_stride; CNT += (size_t)slot * cnt_stride;
    }
    const int tid = threadIdx.x, cl = tid & 15, hi = tid >> 4;
    if (blockIdx.x == gridDim.x - 1) {       // the bins' counts (small integers: exact in any order): wave w = bins w, w + 4, ...
        const int lane = tid & 63, wv = tid >> 6;
        for (int b2 = wv; b2 < nb; b2 += 4) {
            double c = 0.0;
            for (int u = lane; u < nunits; u += 64) c += PC[(size_t)u * nb + b2];
            c = wave_bfly(c);
            if (lane == 0) CNT[b2] = c;
        }
        return;
    }
    const int col = blockIdx.x * 16 + cl;
    const bool on = col < nq * d;
    const int q = on ? col / d : 0, j = on ? col - q * d : 0;
    const double* src = PR + (size_t)q * ld + j;
    const size_t ustride = (size_t)nq * ld;
    const int ngroups = (nunits + 15) / 16;
    double t = 0.0;
    for (int G0 = 0; G0 < ngroups; G0 += 16) {
        double v[16];
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) { const int u = 16 * (G0 + gg) + hi; v[gg] = (on && u < nunits) ? src[(size_t)u * ustride] : 0.0; }
#pragma unroll
        for (int gg = 0; gg < 16; ++gg) s_v[gg][hi][cl] = v[gg];
        __syncthreads();
        {
            const int nu = min(16, nunits - 16 * (G0 + hi));             // (thread (group hi, column cl))
            double gs = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) if (i < nu) gs = gs + s_v[hi][i][cl];
            s_g[hi][cl] = gs;
        }
        __syncthreads();
        if (tid < 16) for (int gg = 0; gg < 16 && G0 + gg < ngroups; ++gg) t = t + s_g[gg][tid];
        __syncthreads();
    }
    if (tid < 16 && on) TOT[(size_t)q * ld + j] = t;
}

// the update on its own (adapt_apply_wave), in place on the engine's current state: behind the multi-kernel path's generations, sharded
// runs, and whenever totals are still pending when a launch that cannot apply them follows
template <int NCH>
__global__ __launch_bounds__(64) void k_adapt_apply(Params p, const double* __restrict__ TOT, const double* __restrict__ CNT)
{
    adapt_apply_wave<NCH>(p, TOT, CNT, p.cr_probs, nullptr, p.cr_probs, (int)threadIdx.x);
}

// adapt_lag >= 1: the first half of the update -- the bins' dot products with the weights 1 / sd^2 (adapt_dots_wave) -- right behind the totals
// of generations g0 .. g0 + gridDim.x - 1 (one wave each; ring slots as in k_adapt_totals), DOT [slot][nbp]; the second half
// (adapt_pending_apply) runs when the update is due: in the prologue of a persistent launch, or as k_adapt_apply_pending
template <int NCH>
__global__ __launch_bounds__(64) void k_adapt_dots(Params p, const double* __restrict__ TOT, const double* __restrict__ CNT, double* __restrict__ DOT,
                                                   long long g0, int R1, long long tot_stride, int nbp)
{
    const int slot = (int)((g0 + (long long)blockIdx.x) % R1), lane = threadIdx.x, nb = p.ncr + p.ngamma;
    const double my_cnt = lane < nb ? CNT[(size_t)slot * nbp + lane] : 0.0;
    const double dot = adapt_dots_wave<NCH>(p, TOT + (size_t)slot * tot_stride, my_cnt, lane);
    if (lane < nbp) DOT[(size_t)slot * nbp + lane] = lane < nb ? dot : 0.0;
}
// ... on its own, in place on the engine's current state: everything that is due before generation g (the multi-kernel path, launches that
// are not fused, the hand-over at the end of the burn-in, the end of dz_step)
__global__ __launch_bounds__(64) void k_adapt_apply_pending(Params p, const double* __restrict__ DOT, const double* __restrict__ CNTR, int nbp, int R1, long long pend0, long long pend1,
                                                            long long g, int lag)
{
    adapt_pending_apply(p, DOT, CNTR, nbp, R1, pend0, pend1, g, 1, lag, p.burnin, p.cr_probs, nullptr, p.cr_probs, (int)threadIdx.x);
}

// ------------------------------------------------------------------------------------------
// Gelman_Rubin (convergence.py:3-20) from the device-resident trace
// ------------------------------------------------------------------------------------------
// (sample order and chain order of the sums are those of the oracle's restatement -- the loads of a batch travel together)
__global__ void k_chain_moments(const double* __restrict__ tX, int nl, int d, int ld, long long tcap, int nsamples, double* __restrict__ mean, double* __restrict__ var)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= d) return;
    const int nb = nsamples / 2, n2 = nsamples - nb;
    const double* x = tX + ((size_t)c * tcap + nb) * ld + j;
    const double m = strided_sum<0>(x, (size_t)ld, n2, 0.0) / (double)n2;
    const double v = strided_sum<2>(x, (size_t)ld, n2, m);
    mean[(size_t)c * d + j] = m; var[(size_t)c * d + j] = v / (double)n2;
}
// block = 16 dimensions x 64 strip slots: thread (dimension, slot) adds the strips slot, slot + 64, ... of 64 chains each (in chain order),
// then thread (dimension, 0) adds the strips in order -- W, the mean of the chain means, B in three such passes (round 3's form: one
// thread per dimension walking all chains three times, two waves for the whole job)
__global__ __launch_bounds__(1024) void k_rhat(const double* __restrict__ mean, const double* __restrict__ var, int nch, int d, int nsamples, double* __restrict__ rhat)
{
    extern __shared__ __attribute__((aligned(16))) double sp[];            // [strips][16]
    __shared__ double s_mm[16];
    const int jl = threadIdx.x & 15, slot = threadIdx.x >> 4, j = blockIdx.x * 16 + jl;
    const bool on = j < d;
    const int ns = (nch + 63) / 64;
    auto strips = [&](const double* src, int mode, double m) {
        for (int s = slot; s < ns; s += 64) {
            const int n = min(64, nch - 64 * s);
            const double* q = src + (size_t)64 * s * d + j;
            sp[s * 16 + jl] = !on ? 0.0 : (mode == 0 ? strided_sum<0>(q, (size_t)d, n, 0.0) : strided_sum<2>(q, (size_t)d, n, m));
        }
        __syncthreads();
        double t = 0.0;
        if (slot == 0) for (int s = 0; s < ns; ++s) t = t + sp[s * 16 + jl];
        __syncthreads();
        return t;
    };
    const double W = strips(var, 0, 0.0) / (double)nch;
    const double mm = strips(mean, 0, 0.0) / (double)nch;
    if (slot == 0) s_mm[jl] = mm;
    __syncthreads();
    const double B = strips(mean, 2, s_mm[jl]) / (double)nch;
    if (slot == 0 && on) {
        const double var_est = W * (1.0 - 1.0 / (double)nsamples) + B;
        rhat[j] = sqrt(var_est / W);
    }
}
#endif  // DZ_TEMPLATES_ONLY

}  // namespace dz
