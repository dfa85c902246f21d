// dz_megakernel_d2.h -- k_generations_d2: the persistent generation kernel for 128 < d <= 256 (round 5; the reference example's own d = 200,
// pydream/examples/ndim_gaussian/dream_ex_ndim_gaussian.py:29).
//
// Same scheme as k_generations (dz_megakernel.h): a block owns its chains through a whole thin-cycle, proposal points live only in LDS as
// MFMA point tiles, four block barriers per generation.  What differs at these dimensions:
//   * a lane owns four dimensions (two 128-dimension chunks, NCH = 2): the tries are the multi-kernel path's propose_set<2> -- the same
//     arithmetic, the same per-lane summation order, hence the same bits -- writing LDS rows instead of HBM rows;
//   * the matrix does NOT fit LDS next to the point tiles (packed triangle at d = 200: 186 KB; the tiles of 16 chains x 5 tries: 129 KB): the
//     likelihood units read their A operand (16 matrix rows x 4 k per MFMA, 128-byte segments) from L2 -- the matrix is a few hundred KB,
//     resident in every XCD's 4 MB -- while the B operand (the points) comes from LDS as before; the four waves of a SIMD hide each other's
//     L2 round trips;
//   * the chain states stay in HBM (one row read and, when a move is accepted, written per generation).
// One wave per chain; 16 chains per block while the point tiles fit (d <= ~218 at 5 tries), else 8.  Eligibility (host, mega_d2_eligible):
// MVN likelihood, 128 < ld <= 256, multitry 3..15, flat priors, no boundaries, one DE pair; everything else at these dimensions keeps the
// multi-kernel path.  Inside the crossover burn-in a launch covers one generation and publishes the positions; the adaptation sums come
// from k_adapt_partials.
#pragma once
#include "dz_megakernel.h"

namespace dz {

// K1: multitry off (the reference's default, Dream.py:271-275 and :326-334) -- one proposal per generation, no reference set, the single-try
// snooker formula and the current point's term log |x - z|^(d-1).
// PB: per-dimension priors (SampledParam norm / uniform, parameters.py:37-47), hard boundaries (Dream.py:733-791), several DE pairs
// (set_DEpair :571-583) -- the multi-kernel path's full proposal code and prior evaluation, constants from global memory.
// WPC (round 6): waves per chain -- 2 at 8 chains per block, where the point tiles of 16 chains no longer fit LDS (d > ~228 at 5 tries): the
// tries of a phase are dealt to the chain's two waves, all 16 waves share the likelihood units, the chain's first wave selects and makes
// the Metropolis step (as in k_generations: two more barriers per generation keep the base point and the new state consistent between them).
// SP (round 6): where the point tiles of 16 chains x k tries do not fit LDS but those of k - 1 tries do (229..256 dimensions at 5 tries), the
// proposal set goes through the tiles in TWO passes -- tries 0 .. k-2, their likelihoods, then try k-1 over the rows of try 0 -- and the reference
// set takes all of them, the selected proposal staying in registers until the Metropolis step.  A selected try whose rows were overwritten
// (try 0) is generated again from the same counters: the same bits (one extra try for a fifth of the chain-generations at 5 tries).
template <int NRT, bool TRI, int CH, bool K1 = false, bool PB = false, int WPC = 1, bool SP = false>
__global__ __launch_bounds__(64 * CH * WPC) void k_generations_d2(const Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0, int64_t zappend, int seg0, Publish pub)
{
    static_assert(WPC == 1 || !K1, "one try per generation: one wave per chain");
    static_assert(!SP || (!K1 && ((CH == 16 && WPC == 1) || (CH == 8 && WPC == 2))), "the two-pass proposal set: 16 chains x 1 wave or 8 chains x 2 waves per block, multi-try");
    constexpr int NCH = NRT > 8 ? 2 : 1, NT = 64 * CH * WPC;    // (NRT = 8, ld = 128: one chunk -- the kernel also serves 113..128 dimensions, where the
                                                                //  matrix in LDS leaves room for the point tiles of 8 chains only)
    double* const publish = pub.to;
    const Params& p = *pp;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int d = p.d, k = K1 ? 1 : p.k, ld = p.ld;
    const bool RG = pub.multi == 2;      // several burn-in generations per launch, the positions of each into the ring (Publish, dz_kernels.h)
    const MegaLayout L = mega_layout(d, k, NRT, p.ncr, p.ngamma, TRI, false, CH, false, false, true, RG ? pub.lag + 1 : 0, SP);
    constexpr int nph = K1 ? 1 : 2;
    double* Pt = smem + L.off_P;
    double* qb = smem + L.off_q;
    double* sP = smem + L.off_sP; double* sS = smem + L.off_sS; double* sL = smem + L.off_sL;
    double* rP = smem + L.off_rP; double* rS = smem + L.off_rS;
    double* mus = smem + L.off_mu;
    double* probs = smem + L.off_pr;
    double* st = smem + L.off_st;
    double* dec = smem + L.off_dec;
    double* gts = smem + L.off_gt;
    const double* Mg = TRI ? p.Mtp : p.Mt;           // the matrix, in global memory (L2): packed triangle / transposed square [ld][ld]
    const int LDMg = ld;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cl = WPC == 1 ? wv : wv / WPC;                            // chain inside the block; this wave's number among the chain's waves (rotated per chain:
    const int sub = WPC == 1 ? 0 : ((wv % WPC) + cl) % WPC;             //  the first waves of the chains spread over the SIMDs)
    const int cg = pub.c0 + blockIdx.x * CH + cl;
    const bool active = cg < pub.c1;
    const int c = min(cg, pub.c1 - 1);
    const uint32_t gc = (uint32_t)(p.off + c);
    const int tstride = CH * L.LDP;                                      // try i of chain cl: Pt + (CH i + cl) LDP
    double* region = Pt + (size_t)cl * L.LDP;

    for (int i = threadIdx.x; i < L.rows * L.LDP; i += NT) Pt[i] = 0.0;
    for (int i = threadIdx.x; i < 4 * ((d + 3) / 4) + 4; i += NT) mus[i] = i < d ? p.mu[i] : 0.0;
    if (RG) {
        if (wv == 0) adapt_pending_apply(p, pub.DOT, pub.CNTR, pub.nbp, pub.lag + 1, pub.pend0, pub.pend1, (long long)g0, ngen, pub.lag, pub.burnin, pub.sh, smem + L.off_tab,
                                         blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else if (pub.TOT) {
        if (wv == 0) adapt_apply_wave<NCH>(p, pub.TOT, pub.CNT, pub.sh, probs, blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else {
        if ((int)threadIdx.x < p.ncr) probs[threadIdx.x] = pub.sh[threadIdx.x];
        if ((int)threadIdx.x < p.ngamma) probs[p.ncr + threadIdx.x] = pub.sh[3 * p.ncr + threadIdx.x];
    }
    for (int i = threadIdx.x; i < p.ngamma * d; i += NT) gts[i] = p.gtab[(size_t)(i / d) * p.depairs * d + (i % d)];
    if (lane == 0 && sub == 0) { st[4 * cl] = p.lprior[c]; st[4 * cl + 1] = p.llike[c]; st[4 * cl + 2] = 0.0; dec[8 * cl + 7] = chain_T(p, c); }
    __syncthreads();
    if (RG && blockIdx.x == 0 && pub.c0 == 0)
        for (int i = threadIdx.x; i < ngen * pub.nbp; i += NT) pub.PG[i] = smem[L.off_tab + i];

    // More than 15 tries (round 6; the reference takes any integer, Dream.py:155-161): a generation's draw slots (3 control + two per try of both
    // sets) exceed the wave's 64 lanes -- the wave then holds ONE phase's slots at a time (DrawSrc::base: k npt <= 64 of them), the three
    // control slots are evaluated where they are read; selection and multi-try ratio over 32 + 32 lanes (mt_select_vals<true>, mt_log_ratio<true>).
    const bool bigk = p.nslots > 64;
    const int base1 = 3 + k * p.npt;                    // first slot of the reference set's tries
    auto generation_draws = [&](uint32_t g_, int base_ = 0) {          // lane s holds slot base_ + s of the chain's wave-uniform draws of generation g_
        DrawSrc q; q.have = true; q.mine = make_uint4(0, 0, 0, 0); q.base = base_;
        if (lane + base_ < p.nslots) { const u32x4 w = slot_counter_draw(p, lane + base_, gc, g_); q.mine = make_uint4(w.x, w.y, w.z, w.w); }
        return q;
    };
    // Q = q_0 + q_1 + ... of point row `pt` in ascending row tile (the MVN contract): the reads in batches of eight, then the ordered adds
    auto q_sum = [&](int pt) {
        double Q = 0.0;
#pragma unroll
        for (int t0 = 0; t0 < NRT; t0 += 8) {
            double qt[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) qt[j] = t0 + j < NRT ? qb[pt * NRT + t0 + j] : 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (t0 + j < NRT) Q = Q + qt[j];
        }
        return Q;
    };
    DrawSrc dsn = generation_draws(g0, bigk ? 3 : 0);
    uint32_t M = M0;                                                        // (appends inside the launch: k_generations)
    int next_app = zappend >= 0 ? seg0 - 1 : -1;

    for (int gi = 0; gi < ngen; ++gi) {
        const uint32_t g = g0 + (uint32_t)gi;
        const bool last = gi == ngen - 1;
        const bool app = gi == next_app;
        DrawSrc ds = dsn;
        double bsel[NCH][2];                                                          // (SP) the selected proposal, from the selection to the Metropolis step
#pragma unroll
        for (int it = 0; it < NCH; ++it) { bsel[it][0] = 0.0; bsel[it][1] = 0.0; }
        for (int phase = 0; phase < nph; ++phase) {
            if (bigk && phase == 1) ds = generation_draws(g, base1);                  // the reference set's slots take the proposal set's place
            // ---- phase 0: k proposals around the chain's state (generate_proposal_points :258-264) into the chain's rows of tiles 0..k-1;
            //      phase 1: the selected proposal moves to tile 0 and k-1 reference points around it (:295-299) take tiles 1..k-1
            StepFlags f;
            double base[NCH][2];
            if (phase == 0) {
                Ctrl u;
                const u32x4 w0 = bigk ? slot_counter_draw(p, 0, gc, g) : uniform_draw(p, ds, 0, gc, g), w1 = bigk ? slot_counter_draw(p, 1, gc, g) : uniform_draw(p, ds, 1, gc, g),
                            w2 = bigk ? slot_counter_draw(p, 2, gc, g) : uniform_draw(p, ds, 2, gc, g);
                u.u_snk = u53(w0.x, w0.y); u.u_cr = u53(w0.z, w0.w); u.u_de = u53(w1.x, w1.y); u.u_glev = u53(w1.z, w1.w);
                u.u_sel = u53(w2.x, w2.y); u.u_acc = u53(w2.z, w2.w);
                const double* const pr_g = RG ? smem + L.off_tab + (size_t)gi * pub.nbp : probs;      // the probabilities this generation decides with
                f = step_flags_from(p, u, pr_g, pr_g + p.ncr);                       // Dream.py:246-256
                if (lane == 0 && sub == 0) {
                    double* dc = dec + 8 * cl;
                    dc[0] = u.u_sel; dc[1] = u.u_acc; dc[2] = f.snk ? 1.0 : 0.0; dc[3] = (double)f.cr_idx; dc[4] = (double)f.glev;
                    if (PB) dc[5] = (double)f.delta;
                }
                load_row<NCH>(p.X + (size_t)c * ld, ld, lane, base);
            } else {
                const double* dc = dec + 8 * cl;
                f.snk = dc[2] != 0.0; f.cr_idx = (int)dc[3]; f.delta = PB ? (int)dc[5] : 1; f.glev = (int)dc[4];
                const double u_sel = dc[0];
                double lp = -__builtin_huge_val();
                if (lane < k) {                                                      // likelihoods of the chain's k points, mt_choose_proposal_pt (:291)
                    const double lk = SP ? sL[cl * k + lane] : nan_to_ninf(p.logF - 0.5 * q_sum(lane * CH + cl));      // (SP: made pass by pass, below)
                    if (!SP && sub == 0) sL[cl * k + lane] = lk;
                    lp = sP[cl * k + lane] + dec[8 * cl + 7] * lk;
                }
                bool fin;
                const int sel = k > 16 ? mt_select_vals<true>(k, lp, u_sel, lane, &fin) : mt_select_vals(k, lp, u_sel, lane, &fin);
                if (lane == 0 && sub == 0) st[4 * cl + 2] = (double)(sel | (fin ? 256 : 0));
                int srow = sel;
                if (SP) {     // where the selected try's point is: the last try sits in row 0; try 0 was overwritten by it and is generated again (same counters, same bits)
                    if (sel == k - 1) srow = 0;
                    else if (sel == 0 && sub == 0) {
                        const bool snk0 = __builtin_amdgcn_readfirstlane((int)f.snk) != 0;
                        const double* grow0 = PB ? gamma_row(p, f.glev, f.delta) : gts + (size_t)(__builtin_amdgcn_readfirstlane(f.glev) - 1) * d;
                        double xb[NCH][2];
                        load_row<NCH>(p.X + (size_t)c * ld, ld, lane, xb);
                        if (PB) propose_set<NCH, false, true, 0>(p, 0, g, M, c, gc, 0, 1, k, lane, xb, grow0, snk0, f.cr_idx, f.delta, f.glev, ds, region, tstride, sS + cl * k, nullptr, sP + cl * k, nullptr);
                        else propose_set<NCH, false, false, 1>(p, 0, g, M, c, gc, 0, 1, k, lane, xb, grow0, snk0, f.cr_idx, 1, f.glev, ds, region, tstride, sS + cl * k, nullptr, sP + cl * k, nullptr);
                    }
                    if (sel == 0) srow = 0;
                    if (WPC > 1) __syncthreads();                                    // (the chain's other wave reads the point its first wave has just made again)
                }
                const double* row = region + (size_t)srow * tstride;
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    const int jj = 128 * it + 2 * lane;
                    base[it][0] = jj < d ? row[jj] : 0.0; base[it][1] = jj + 1 < d ? row[jj + 1] : 0.0;
                }
                if (WPC > 1) __syncthreads();                                        // every wave of the chain holds the base point before any try row is rewritten
                if (SP) {
#pragma unroll
                    for (int it = 0; it < NCH; ++it) { bsel[it][0] = base[it][0]; bsel[it][1] = base[it][1]; }
                } else
                if (sub == 0) {
#pragma unroll
                    for (int it = 0; it < NCH; ++it) {                               // the selected proposal now sits in tile 0 (each lane moves its own values)
                        const int jj = 128 * it + 2 * lane;
                        if (jj < d) region[jj] = base[it][0];
                        if (jj + 1 < d) region[jj + 1] = base[it][1];
                    }
                }
            }
            const bool snk_s = __builtin_amdgcn_readfirstlane((int)f.snk) != 0;
            const double* grow = PB ? gamma_row(p, f.glev, f.delta) : gts + (size_t)(__builtin_amdgcn_readfirstlane(f.glev) - 1) * d;
            const int n = k - phase;
            const int i0 = WPC == 1 ? 0 : (sub * n) / WPC, i1 = WPC == 1 ? n : ((sub + 1) * n) / WPC;     // this wave's tries
            double* slp = phase ? rS + cl * (k - 1) : sS + cl * k;
            double* prp = phase ? rP + cl * (k - 1) : sP + cl * k;
            // tries [a, b) of the set into rows (try - r0) of the chain's column of tiles
            auto propose = [&](int a, int b, int r0) {
                if (snk_s) __builtin_amdgcn_s_setprio(3);      // a snooker set is the longest path to the block's barrier
                double* out = region + (size_t)((SP ? 0 : phase) - r0) * tstride;
                if (a < b) {
                    if (PB) propose_set<NCH, false, true, 0>(p, phase, g, M, c, gc, a, b, n, lane, base, grow, snk_s, f.cr_idx, f.delta, f.glev, ds,
                                                             out, tstride, slp, K1 ? st + 4 * cl + 3 : nullptr, prp, nullptr);
                    else
                    propose_set<NCH, false, false, K1 ? 2 : 1>(p, phase, g, M, c, gc, a, b, n, lane, base, grow, snk_s, f.cr_idx, 1, f.glev, ds,
                                                               out, tstride, slp, K1 ? st + 4 * cl + 3 : nullptr, prp, nullptr);   // (k = 1: log |x - z|^(d-1) of the current point, :328-329)
                }
                if (snk_s) __builtin_amdgcn_s_setprio(0);
            };
            auto units = [&](int row0, int ntl) {   // mt_evaluate_logps :278, :302 -- the (point tile, row tile) units, A operand from L2
                if (TRI) {
                    if (p.mu_zero) mfma_units_d2<NRT, true>(p, Mg, Pt, mus, qb, row0, ntl, wv, CH * WPC, lane, L.LDP);
                    else mfma_units_d2<NRT, false>(p, Mg, Pt, mus, qb, row0, ntl, wv, CH * WPC, lane, L.LDP);
                } else {
                    if (p.mu_zero) mfma_units<NRT, TRI, true>(p, Mg, Pt, mus, qb, row0, ntl, wv, CH * WPC, lane, LDMg, L.LDP);
                    else mfma_units<NRT, TRI, false>(p, Mg, Pt, mus, qb, row0, ntl, wv, CH * WPC, lane, LDMg, L.LDP);
                }
            };
            if (SP && phase == 0) {      // the proposal set in two passes over the rows of k - 1 tries: tries 0 .. k-2, then try k-1 over try 0's rows
                const int a0 = WPC == 1 ? 0 : (sub * (k - 1)) / WPC, a1 = WPC == 1 ? k - 1 : ((sub + 1) * (k - 1)) / WPC;
                propose(a0, a1, 0);
                __syncthreads();
                units(0, ((k - 1) * CH + 15) / 16);
                __syncthreads();
                if (lane < k - 1 && sub == 0) sL[cl * k + lane] = nan_to_ninf(p.logF - 0.5 * q_sum(lane * CH + cl));
                if (sub == WPC - 1) propose(k - 1, k, k - 1);
                __syncthreads();
                units(0, 1);                                                 // (8 chains per block: the tile's other half holds try 1 -- its sums come out the same again)
                __syncthreads();
                if (lane == k - 1 && sub == 0) sL[cl * k + lane] = nan_to_ninf(p.logF - 0.5 * q_sum(cl));
                if (WPC > 1) __syncthreads();                                // (both waves of the chain select from sL)
                continue;
            }
            propose(i0, i1, 0);
            if (phase == nph - 1 && !last) dsn = generation_draws(g + 1u, bigk ? 3 : 0);
            __syncthreads();                                                         // points visible
            {
                const int row0 = (phase && !SP) ? CH : 0, ntl = (n * CH + 15) / 16;
                units(row0, ntl);
            }
            __syncthreads();                                                         // q visible
        }
        // ---- Metropolis step (:305-347), trace (core.py:114-116), record_history (:919-938)
        if (sub == 0) {
            const double* dc = dec + 8 * cl;
            const double u_acc = dc[1];
            const bool snk = dc[2] != 0.0;
            const int cr_idx = (int)dc[3];
            const double lpri = st[4 * cl], llik = st[4 * cl + 1], Tch = dc[7];
            const int sf = (int)st[4 * cl + 2]; const int sel = sf & 255; const bool fin = (sf & 256) != 0;
            double val = -__builtin_huge_val();
            if (K1) {                                                           // single try: the proposal's density straight from the q sums (:271-275)
                val = nan_to_ninf(p.logF - 0.5 * q_sum(cl));                         // (every lane: the same sums)
                sL[cl] = val;
            } else if (lane < k) {
                val = sP[cl * k + lane] + Tch * sL[cl * k + lane];                                       // :279
                if (snk) val = val + sS[cl * k + lane];                                                  // :307
            } else if (lane >= mt_boff(k) && lane < mt_boff(k) + k) {
                const int i = lane - mt_boff(k);
                if (i < k - 1) val = Tch * nan_to_ninf(p.logF - 0.5 * q_sum(((SP ? 0 : 1) + i) * CH + cl)) + rP[cl * (k - 1) + i];     // :303
                else val = Tch * llik + lpri;                                                            // :877-879
                if (snk) { const double sr = i < k - 1 ? rS[cl * (k - 1) + i] : 0.0; val = (val + sr) + sS[cl * k + i]; }   // :312-313
            }
            double lu, ratio;
            if (K1) {
                const double q_logp = Tch * val + sP[cl], last_logp = Tch * llik + lpri;                 // :274, :243
                if (snk) ratio = nan_to_num((q_logp + sS[cl]) - (last_logp + st[4 * cl + 3]));           // :326-332
                else ratio = nan_to_num(q_logp) - nan_to_num(last_logp);                                 // :334
                lu = dlog(u_acc);
            } else {
                ratio = k > 16 ? mt_log_ratio<true>(k, val, u_acc, lane, &lu) : mt_log_ratio(k, val, u_acc, lane, &lu);
                if (!fin) ratio = -__builtin_huge_val();                             // DESIGN.md deviation D1
            }
            const bool accept = is_finite(ratio) && (lu < ratio);                    // :993
            bool diff = false;
            double2 xn[NCH];
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                const int jj = 128 * it + 2 * lane;
                double2 xo = {0.0, 0.0};
                if (jj < ld) xo = *reinterpret_cast<const double2*>(p.X + (size_t)c * ld + jj);
                xn[it] = xo;
                if (accept) {      // the selected proposal
                    if (SP) { xn[it].x = jj < d ? bsel[it][0] : 0.0; xn[it].y = jj + 1 < d ? bsel[it][1] : 0.0; }
                    else { xn[it].x = jj < d ? region[jj] : 0.0; xn[it].y = jj + 1 < d ? region[jj + 1] : 0.0; }
                }
                diff = diff || (xn[it].x != xo.x) || (xn[it].y != xo.y);
            }
            const bool moved = __any(diff);                                          // core.py:120
            const double npri = accept ? sP[cl * k + sel] : lpri, nlik = accept ? sL[cl * k + sel] : llik;   // :345-347
            if (active) {
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    const int jj = 128 * it + 2 * lane;
                    if (jj < ld) {
                        if (accept) gstore2(p.X + (size_t)c * ld + jj, xn[it]);
                        if (trace_slot0 >= 0) gstore2(p.tX + ((size_t)c * p.tcap + (size_t)(trace_slot0 + gi)) * ld + jj, xn[it]);
                        if (app) gstore2(p.Z + ((size_t)zappend + (M - M0) + gc) * ld + jj, xn[it]);                         // record_history :933-936
                        if (publish) gstore2(publish + (RG ? (size_t)(g % (uint32_t)(pub.lag + 2)) * pub.pos_stride : (size_t)0) + (size_t)gc * ld + jj, xn[it]);                          // set_current_position_arr :447-449
                    }
                }
                if (lane == 0) {
                    if (trace_slot0 >= 0) {
                        const size_t o = (size_t)(trace_slot0 + gi) * p.nl + c;
                        p.tlogp[o] = Tch * nlik + npri;                              // core.py:115 / :178
                        p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = cr_idx; p.tsnk[o] = snk ? 1 : 0;
                    }
                    if (last) { p.lprior[c] = npri; p.llike[c] = nlik; }
                }
            }
            if (lane == 0) { st[4 * cl] = npri; st[4 * cl + 1] = nlik; }
        }
        // (one wave per chain: no barrier here -- the next generation's first phase only touches each wave's own chain's rows and scalars;
        //  the state row in HBM was written by this wave and is read by this wave)
        if (WPC > 1) __syncthreads();                                                // the chain's other wave reads the new state (same CU: its stores are drained before the barrier)
        if (app) { next_app += p.thin; M += (uint32_t)p.N; }
    }
}

}  // namespace dz
