// dz_megakernel_w4.h -- k_generations_w4: the persistent generation kernel for SMALL populations (fewer than 8 chains per CU:
// BASELINE configs[1], 1024 chains on 256 CUs), where a block holds 4 chains with FOUR waves per chain and a generation is bound by the
// LENGTH of one chain's dependency chain, not by instruction throughput (round 4: 15.9 us per generation whether 256, 512 or 1024 chains).
//
// What k_generations<.., 4, 4, lean> does with its four waves per chain: the tries of a set are dealt to the waves, everything else --
// decisions, selection -- is computed redundantly by all four, and the Metropolis step by the first while the other three wait.  Cycle
// stamps (profiles/r03_stamps_1024chains.txt): Metropolis step 5.9 k cycles -> first try round 4.6 k + a second round for the fifth try
// 4.2 k -> likelihood -> selection 2.7 k -> barrier 2.4 k -> tries 2.6 k -> likelihood: 34.7 k cycles, of which 20 k are those serial parts.
//
// Here the parts of a DE try that do not depend on the point it is made around -- the dimension draws (Philox, Box-Muller pair), the crossover
// mask and its count d', gamma, the archive rows' difference, e gamma (Z_a - Z_b) and zeta: all of generate_proposal_points (Dream.py:688-726)
// except the last two additions -- are made AHEAD by the chain's waves 1..3 ("pre-tries", at most two per wave, kept in registers) while wave 0
//   * makes the Metropolis step of the generation before (the proposal set of generation g + 1 around a state that is still being decided), and
//   * makes the selection among the proposals (the reference set around a point that is still being chosen);
// when the base point is known a try is finished by two additions and a select per dimension (de_finish).  Same operations in the same order
// as propose_point (dz_kernels.h), hence the same bits: prop = (x + (e gamma) (Z_a - Z_b)) + zeta where U < CR, x elsewhere.
// Snooker sets (one chain-generation in ten) project onto x - z and cannot be made ahead: they run as before, after the base point, over all
// four waves.  Eligibility (host): multitry 3..6 (two pre-tries per wave cover the set), flat priors, no boundaries, one DE pair, the
// chains' states in LDS; everything else at 4 chains per block keeps k_generations<.., 4, 4, ..>.
#pragma once
#include "dz_megakernel.h"

namespace dz {

// the base-independent half of a DE try: per lane the two increments t = (e gamma) (Z_a - Z_b) and the two zeta terms; the crossover mask
// as two wave-wide ballots (scalar registers)
struct PreTry { double t0, t1; float z0, z1; uint32_t keep; };      // (zeta's normals stay binary32 until the try is finished; keep: bit s = U < CR for the lane's dimension 2 lane + s)

// = propose_point's DE branch (Dream.py:692-709) up to, not including, the additions to the base point
DZ_DEV void de_pretry(const Params& p, const SetConsts& sc, int phase, uint32_t g, uint32_t gc, int i, int lane, const double* __restrict__ grow,
                      const DrawSrc& dr, const RowPair& R, PreTry& o)
{
    const int d = p.d, j0 = 2 * lane;
    const u32x4 w = philox(p.k0, p.k1, (uint32_t)(j0 >> 1), stream_id(K_DIM, (uint32_t)i, (uint32_t)phase), gc, g);
    float z0, z1;
    normal32_pair(w.z, w.w, z0, z1);
    const bool in0 = j0 < d, in1 = j0 + 1 < d;
    const bool u0 = (w.x & 0xffffu) < sc.thr, u1 = (w.x >> 16) < sc.thr;                      // U_j < CR :700, :704
    const double e0 = uniform16(w.y, sc.ec1, sc.ec0) + 1.0, e1 = uniform16(w.y >> 16, sc.ec1, sc.ec0) + 1.0;      // :696-697
    o.z0 = z0; o.z1 = z1;                                                                      // :694 (times zeta in de_finish)
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(in0), m1 = __builtin_amdgcn_ballot_w64(in1);
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(u0) & m0, b1 = __builtin_amdgcn_ballot_w64(u1) & m1;
    const int dprime = __popcll(b0) + __popcll(b1);                                          // d' :704 / :709
    const u32x4 wg = uniform_draw(p, dr, sc.slot0 + i * sc.npt, gc, g);                      // set_gamma :615
    double gamma = 1.0;
    if (!u53_below(wg.x, wg.y, sc.pgu_thr)) gamma = grow[(dprime == 0 ? d : dprime) - 1];     // gamma_arr[level-1][0][d'-1] :624
    double t0 = e0 * gamma; t0 = t0 * (R.a.x - R.b.x);                                       // chain_differences :692, :714
    double t1 = e1 * gamma; t1 = t1 * (R.a.y - R.b.y);
    o.t0 = t0; o.t1 = t1; o.keep = ((in0 && u0) ? 1u : 0u) | ((in1 && u1) ? 2u : 0u);
}
// ... and the rest, around the base point x (:714 / :717, crossover :720-726), into the try's LDS row
DZ_DEV void de_finish(const PreTry& o, double zeta, double x0, double x1, double* __restrict__ out, int lane, int d)
{
    double q0 = x0 + o.t0; q0 = q0 + zeta * (double)o.z0;
    double q1 = x1 + o.t1; q1 = q1 + zeta * (double)o.z1;
    const double r0 = (o.keep & 1u) ? q0 : x0, r1 = (o.keep & 2u) ? q1 : x1;
    const int jj = 2 * lane;
    if (jj < d) { out[jj] = r0; out[jj + 1] = jj + 1 < d ? r1 : 0.0; }                        // (d odd: the owner of the last dimension rewrites the first pad column with its zero, as propose_point does)
}

template <int NRT, bool TRI>
__global__ __launch_bounds__(1024) void k_generations_w4(const Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0, int64_t zappend, int seg0, Publish pub)
{
    constexpr int CH = 4, WPC = 4, NT = 64 * CH * WPC, NCH = 1;
    double* const publish = pub.to;
    const Params& p = *pp;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int d = p.d, k = p.k, ld = p.ld;
    const bool RG = pub.multi == 2;      // several burn-in generations per launch, the positions of each into the ring (Publish, dz_kernels.h)
    const MegaLayout L = mega_layout(d, k, NRT, p.ncr, p.ngamma, TRI, true, CH, false, false, false, RG ? pub.lag + 1 : 0);
    double* Ms = smem;
    double* Pt = smem + L.off_P;
    double* qb = smem + L.off_q;
    double* sP = smem + L.off_sP; double* sS = smem + L.off_sS; double* sL = smem + L.off_sL;
    double* rP = smem + L.off_rP; double* rS = smem + L.off_rS;
    double* mus = smem + L.off_mu;
    double* probs = smem + L.off_pr;
    double* st = smem + L.off_st;
    double* dec = smem + L.off_dec;
    double* gts = smem + L.off_gt;
    double* Xs = smem + L.off_X;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // chain inside the block; this wave's role among the chain's four (0: selection and Metropolis step; 1..3: tries).  Wave w runs on SIMD
    // w % 4: a chain's four waves sit on FOUR SIMDs (when the block waits for one chain's snooker set, that set has the CU's SIMDs to
    // itself instead of one), and the roles are rotated so that every SIMD hosts one deciding wave and three try waves.
    // (measured, 1024 chains, steady state: all of a chain's waves on one SIMD 352.6 M proposals/s, spread 375.1 -- one block in three holds a
    //  snooker chain and a launch ends with its unluckiest block; with the deciding wave's issue priority, below, 347 against 382, and the
    //  DE-only generations, 12.35 against 13.2 us before it, no longer differ: 12.65 / 12.60)
#ifdef DZ_ONE_SIMD_PER_CHAIN                                             // (experiment switch: a chain's four waves on ONE SIMD; measured slower, EXPERIMENTS.md)
    const int cl = wv & 3, sub = wv >> 2;
#else
    const int cl = wv >> 2, sub = ((wv & 3) + cl) & 3;
#endif
    const int cg = pub.c0 + blockIdx.x * CH + cl;
    const bool active = cg < pub.c1;
    const int c = min(cg, pub.c1 - 1);
    const uint32_t gc = (uint32_t)(p.off + c);
    const int tstride = CH * L.LDP;                                      // try i of chain cl: Pt + (CH i + cl) LDP
    double* region = Pt + (size_t)cl * L.LDP;

    // ---- staging: the matrix, mu, the selection probabilities, the gamma table, the chains' states and log densities (as k_generations)
    if (TRI) {
        const int nvec = L.off_P >> 1;
        const double2* src = reinterpret_cast<const double2*>(p.Mtp);
        double2* dst = reinterpret_cast<double2*>(Ms);
        for (int i = threadIdx.x; i < nvec; i += NT) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < L.off_P; i += NT) {
            const int row = i / L.LDM, col = i - row * L.LDM;
            Ms[i] = (row < d && col < d) ? p.Mt[(size_t)row * ld + col] : 0.0;
        }
    }
    for (int i = threadIdx.x; i < L.rows * L.LDP; i += NT) Pt[i] = 0.0;
    if ((int)threadIdx.x < 4 * ((d + 3) / 4) + 4) mus[threadIdx.x] = (int)threadIdx.x < d ? p.mu[threadIdx.x] : 0.0;
    if (RG) {
        if (wv == 0) adapt_pending_apply(p, pub.DOT, pub.CNTR, pub.nbp, pub.lag + 1, pub.pend0, pub.pend1, (long long)g0, ngen, pub.lag, pub.burnin, pub.sh, smem + L.off_tab,
                                         blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else if (pub.TOT) {
        if (wv == 0) adapt_apply_wave<1>(p, pub.TOT, pub.CNT, pub.sh, probs, blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else {
        if ((int)threadIdx.x < p.ncr) probs[threadIdx.x] = pub.sh[threadIdx.x];
        if ((int)threadIdx.x < p.ngamma) probs[p.ncr + threadIdx.x] = pub.sh[3 * p.ncr + threadIdx.x];
    }
    for (int i = threadIdx.x; i < p.ngamma * d; i += NT) gts[i] = p.gtab[(size_t)(i / d) * p.depairs * d + (i % d)];
    if (lane == 0 && sub == 0) { st[4 * cl] = p.lprior[c]; st[4 * cl + 1] = p.llike[c]; st[4 * cl + 2] = 0.0; dec[8 * cl + 7] = chain_T(p, c); }
    if (sub == 0)
        for (int j = lane; j < L.LDP; j += 64) Xs[cl * L.LDP + j] = j < d ? p.X[(size_t)c * ld + j] : 0.0;
    __syncthreads();
    if (RG && blockIdx.x == 0 && pub.c0 == 0)
        for (int i = threadIdx.x; i < ngen * pub.nbp; i += NT) pub.PG[i] = smem[L.off_tab + i];

    // History appends inside the launch (k_generations): generation index next_app makes the next one into rows zappend + (Mc - M0) + global chain; Mc is the
    // row count generation gcur samples from, Mn the next generation's (rows and pre-tries are requested a generation ahead).
    uint32_t gcur = g0, Mc = M0, Mn = M0;
    int next_app = zappend >= 0 ? seg0 - 1 : -1;
    auto Mof = [&](uint32_t g_) { return g_ == gcur ? Mc : Mn; };
    auto generation_draws = [&](uint32_t g_) {          // lane s holds slot s of the chain's wave-uniform draws of generation g_
        DrawSrc q; q.have = true; q.mine = make_uint4(0, 0, 0, 0);
        if (lane < p.nslots) { const u32x4 w = slot_counter_draw(p, lane, gc, g_); q.mine = make_uint4(w.x, w.y, w.z, w.w); }
        return q;
    };
    auto decide = [&](const DrawSrc& q, uint32_t g_, Ctrl& u) {          // Dream.py:246-256 from the generation's control draws
        const u32x4 w0 = uniform_draw(p, q, 0, gc, g_), w1 = uniform_draw(p, q, 1, gc, g_), w2 = uniform_draw(p, q, 2, gc, g_);
        u.u_snk = u53(w0.x, w0.y); u.u_cr = u53(w0.z, w0.w); u.u_de = u53(w1.x, w1.y); u.u_glev = u53(w1.z, w1.w);
        u.u_sel = u53(w2.x, w2.y); u.u_acc = u53(w2.z, w2.w);
        const double* pr_g = RG ? smem + L.off_tab + (size_t)min((int)(g_ - g0), ngen - 1) * pub.nbp : probs;      // the probabilities generation g_ decides with
        return step_flags_from(p, u, pr_g, pr_g + p.ncr);
    };
    // the tries of a DE set of n that this wave makes ahead: none for the chain's first wave (it is busy with what the set waits for), the
    // set dealt over the other three (n <= 6: at most two each).  A snooker set, made after its base point, is dealt over all four.
    auto de_range = [&](int n, int& a, int& b) { a = sub == 0 ? 0 : ((sub - 1) * n) / 3; b = sub == 0 ? 0 : (sub * n) / 3; };
    auto snk_range = [&](int n, int& a, int& b) { a = (sub * n) / WPC; b = ((sub + 1) * n) / WPC; };
    const uint32_t ldb = 8u * (uint32_t)p.ld;
    RowPair RA, RB, RC;                 // (RC: only the second try of a snooker set)
    RA.a = double2{0.0, 0.0}; RA.b = RA.a; RB = RA; RC = RA;
    PreTry P0, P1;
    P0.t0 = P0.t1 = 0.0; P0.z0 = P0.z1 = 0.0f; P0.keep = 0u; P1 = P0;
    // the rows of this wave's (up to two) pre-tries of set (g_, phase_), requested as early as the draws allow: they travel during the
    // likelihood pass in between
    auto request_rows = [&](const DrawSrc& q, int phase_, uint32_t g_) {
        int a, b; de_range(k - phase_, a, b);
        if (a < b) request_pair<false>(p, q, pt_slot(p, phase_, a, 1), gc, g_, Mof(g_), lane, RA, p.Z, ldb);
        if (a + 1 < b) request_pair<false>(p, q, pt_slot(p, phase_, a + 1, 1), gc, g_, Mof(g_), lane, RB, p.Z, ldb);
    };
    // ... and of a snooker set: the three rows (z and the projected pair, :808-810) of this wave's FIRST try -- the row numbers do not depend on
    // the base point either -- into RA.a, RA.b, RB.a.  Not prefetched they are a full archive-gather latency in front of every snooker set,
    // and with the DE sets finished in a few instructions that latency is what the block's barrier then waits for.
    auto request_snooker_rows = [&](const DrawSrc& q, int phase_, uint32_t g_) {
        int a, b; snk_range(k - phase_, a, b);
        if (a >= b) return;
        const u32x4 w = uniform_draw(p, q, pt_slot(p, phase_, a, 1), gc, g_);
        const uint32_t M = Mof(g_);
        const uint32_t iz = mulhi_idx(w.x, M), i1x = mulhi_idx(w.y, M), i2x = mulhi_idx(w.z, M);
        const uint32_t jb = (uint32_t)min(16 * lane, (int)ldb - 16);
        const char* Zb = reinterpret_cast<const char*>(p.Z);
        RA.a = gload2(reinterpret_cast<const double*>(Zb + (uint64_t)iz * ldb + jb));
        RA.b = gload2(reinterpret_cast<const double*>(Zb + (uint64_t)i1x * ldb + jb));
        RB.a = gload2(reinterpret_cast<const double*>(Zb + (uint64_t)i2x * ldb + jb));
        if (a + 1 < b) {                    // a second try (k = 5: the fifth proposal): its rows as well, so that the two tries can run interleaved
            const u32x4 w2 = uniform_draw(p, q, pt_slot(p, phase_, a + 1, 1), gc, g_);
            const uint32_t jz = mulhi_idx(w2.x, M), j1x = mulhi_idx(w2.y, M), j2x = mulhi_idx(w2.z, M);
            RB.b = gload2(reinterpret_cast<const double*>(Zb + (uint64_t)jz * ldb + jb));
            RC.a = gload2(reinterpret_cast<const double*>(Zb + (uint64_t)j1x * ldb + jb));
            RC.b = gload2(reinterpret_cast<const double*>(Zb + (uint64_t)j2x * ldb + jb));
        }
    };
    // the snooker set's tries [a, b) of this wave (propose_set's snooker branch, dz_kernels.h, with the first try's rows already in flight)
    auto snooker_tries = [&](const DrawSrc& q, int phase_, uint32_t g_, const StepFlags& f_, const double (&base)[NCH][2], const double* grow_, double* slp, double* prp) {
        int a, b; snk_range(k - phase_, a, b);
        if (a >= b) return;
        __builtin_amdgcn_s_setprio(3);          // a snooker set is the longest path to the block's barrier: its waves get issue priority
        double* out = region + (size_t)phase_ * tstride;
        const int n_ = k - phase_;
        double sqv = 1.0;
        RowTerms<NCH> rt0;
        rt0.a[0][0] = RA.a.x; rt0.a[0][1] = RA.a.y; rt0.b[0][0] = RA.b.x - RB.a.x; rt0.b[0][1] = RA.b.y - RB.a.y;          // :819
        if (a + 1 < b) {        // two tries (at most: k <= 6 over four waves), straight-line: two independent dependency chains for the scheduler to interleave
            RowTerms<NCH> rt1;
            rt1.a[0][0] = RB.b.x; rt1.a[0][1] = RB.b.y; rt1.b[0][0] = RC.a.x - RC.b.x; rt1.b[0][1] = RC.a.y - RC.b.y;
            const double sq0 = propose_point<NCH, false, 1>(p, phase_, g_, Mof(g_), c, a, n_, lane, base, grow_, rt0, out + (size_t)a * tstride, nullptr, true, f_.cr_idx, 1, f_.glev, q);
            const double sq1 = propose_point<NCH, false, 1>(p, phase_, g_, Mof(g_), c, a + 1, n_, lane, base, grow_, rt1, out + (size_t)(a + 1) * tstride, nullptr, true, f_.cr_idx, 1, f_.glev, q);
            if (lane == 0) { prp[a] = 0.0; prp[a + 1] = 0.0; }
            sqv = lane == 0 ? sq0 : (lane == 1 ? sq1 : sqv);
        } else {
            const double sq0 = propose_point<NCH, false, 1>(p, phase_, g_, Mof(g_), c, a, n_, lane, base, grow_, rt0, out + (size_t)a * tstride, nullptr, true, f_.cr_idx, 1, f_.glev, q);
            if (lane == 0) prp[a] = 0.0;
            sqv = lane == 0 ? sq0 : sqv;
        }
        snooker_logps(p, sqv, b - a, lane, slp + a);
        __builtin_amdgcn_s_setprio(0);
    };
    auto make_pretries = [&](const DrawSrc& q, int phase_, uint32_t g_, const StepFlags& f_) {
        int a, b; de_range(k - phase_, a, b);
        const SetConsts sc = set_consts(p, phase_, f_.cr_idx);
        const double* grow = gts + (size_t)(__builtin_amdgcn_readfirstlane(f_.glev) - 1) * d;
        if (a < b) de_pretry(p, sc, phase_, g_, gc, a, lane, grow, q, RA, P0);
        if (a + 1 < b) de_pretry(p, sc, phase_, g_, gc, a + 1, lane, grow, q, RB, P1);
    };
    auto finish_pretries = [&](int phase_, double x0, double x1, double* slp, double* prp) {
        int a, b; de_range(k - phase_, a, b);
        double* rows = region + (size_t)phase_ * tstride;
        const double zeta = p.zeta;
        if (a < b) de_finish(P0, zeta, x0, x1, rows + (size_t)a * tstride, lane, d);
        if (a + 1 < b) de_finish(P1, zeta, x0, x1, rows + (size_t)(a + 1) * tstride, lane, d);
        if (lane >= a && lane < b) { slp[lane] = 0.0; prp[lane] = 0.0; }                    // snooker_logp = 0, flat priors: 0
    };

    DrawSrc dsn = generation_draws(g0);
    bool have_pre = false;                                                // this wave holds the pre-tries of the coming proposal set
    Ctrl un;
    StepFlags fn = decide(dsn, g0, un);
    if (fn.snk) request_snooker_rows(dsn, 0, g0);
    else if (sub != 0) request_rows(dsn, 0, g0);

    for (int gi = 0; gi < ngen; ++gi) {
        const uint32_t g = g0 + (uint32_t)gi;
        const bool last = gi == ngen - 1;
        const bool app = gi == next_app;
        gcur = g; Mn = app ? Mc + (uint32_t)p.N : Mc;
        DZ_W0STAMP(0); DZ_WSTAMP(0);
        const DrawSrc ds = dsn;
        const StepFlags f = fn;
        const bool snk_s = __builtin_amdgcn_readfirstlane((int)f.snk) != 0;
        const double* grow = gts + (size_t)(__builtin_amdgcn_readfirstlane(f.glev) - 1) * d;
        if (lane == 0 && sub == 0) {
            double* dc = dec + 8 * cl;
            dc[0] = un.u_sel; dc[1] = un.u_acc; dc[2] = f.snk ? 1.0 : 0.0; dc[3] = (double)f.cr_idx; dc[4] = (double)f.glev;
        }
        // ---- phase 0: k proposals around the chain's state (generate_proposal_points :258-264) into the chain's rows of tiles 0..k-1;
        //      phase 1: the chain's first wave selects (mt_choose_proposal_pt :291) and moves the chosen proposal to tile 0 while the others make
        //      the reference set's pre-tries (:295-299); behind a barrier every wave reads the base point from tile 0.  One copy of the code
        //      serves both phases (the snooker set and the likelihood units are large: two copies of them set the register allocation of
        //      the whole kernel and the DE generations -- nine in ten -- paid for it with spilled registers on their critical path)
        for (int phase = 0; phase < 2; ++phase) {
            if (phase) {
                if (sub == 0) {
                    __builtin_amdgcn_s_setprio(2);                             // the deciding wave is the block's critical path: it outranks the try waves of its SIMD
                    const double u_sel = dec[8 * cl];
                    double lp = -__builtin_huge_val();
                    if (lane < k) {
                        const int pt = lane * CH + cl;
                        double qt[NRT];
#pragma unroll
                        for (int t = 0; t < NRT; ++t) qt[t] = qb[pt * NRT + t];
                        double Q = 0.0;
#pragma unroll
                        for (int t = 0; t < NRT; ++t) Q = Q + qt[t];
                        const double lk = nan_to_ninf(p.logF - 0.5 * Q);
                        sL[cl * k + lane] = lk;
                        lp = sP[cl * k + lane] + dec[8 * cl + 7] * lk;
                    }
                    bool fin;
                    DZ_W0STAMP(11);
                    const int sel = mt_select_vals(k, lp, u_sel, lane, &fin);
                    DZ_W0STAMP(12);
                    if (lane == 0) st[4 * cl + 2] = (double)(sel | (fin ? 256 : 0));
                    const double* row = region + (size_t)sel * tstride;
                    const double b0 = 2 * lane < d ? row[2 * lane] : 0.0, b1 = 2 * lane + 1 < d ? row[2 * lane + 1] : 0.0;
                    if (2 * lane < d) region[2 * lane] = b0;                // (each lane reads its own two values before it writes them: no hazard inside the wave)
                    if (2 * lane + 1 < d) region[2 * lane + 1] = b1;
                    __builtin_amdgcn_s_setprio(0);
                } else if (!snk_s) { make_pretries(ds, 1, g, f); DZ_WSTAMP(11); }
                DZ_W0STAMP(13); DZ_WSTAMP(13);
                __syncthreads();                                           // the selected proposal sits in tile 0
            }
            const int n = k - phase;
            double* slp = phase ? rS + cl * (k - 1) : sS + cl * k;
            double* prp = phase ? rP + cl * (k - 1) : sP + cl * k;
            {
                const double* xr = phase ? region : Xs + cl * L.LDP;
                double base[NCH][2];
                base[0][0] = 2 * lane < d ? xr[2 * lane] : 0.0; base[0][1] = 2 * lane + 1 < d ? xr[2 * lane + 1] : 0.0;
                if (snk_s) snooker_tries(ds, phase, g, f, base, grow, slp, prp);
                else if (sub != 0) {
                    if (phase == 0 && !have_pre) make_pretries(ds, 0, g, f);       // (the first generation of a launch: nothing was made ahead)
                    finish_pretries(phase, base[0][0], base[0][1], slp, prp);
                }
            }
            if (phase == 0) {                                              // the reference set's rows, ahead of the likelihood pass
                if (snk_s) request_snooker_rows(ds, 1, g);
                else if (sub != 0) request_rows(ds, 1, g);
            }
            else if (!last) {                                              // the next generation's draws, decisions and first rows
                dsn = generation_draws(g + 1u);
                fn = decide(dsn, g + 1u, un);
                if (fn.snk) request_snooker_rows(dsn, 0, g + 1u);
                else if (sub != 0) request_rows(dsn, 0, g + 1u);
            }
            DZ_W0STAMP(1 + 4 * phase); DZ_WSTAMP(1 + 4 * phase);
            __syncthreads();                                               // points visible
            DZ_W0STAMP(2 + 4 * phase); DZ_WSTAMP(2 + 4 * phase);
            {   // mt_evaluate_logps :278, :302
                const int row0 = phase ? CH : 0, ntl = (n * CH + 15) / 16;
                // The (point tile, row tile) units are dealt heaviest first, one per wave (at most 14 here).  Wave w runs on SIMD w % 4: dealt in
                // plain order the two heaviest units of every four land on SIMDs 0 and 1 each time (d = 100, triangular factor, 5 tries: 52 / 52 /
                // 39 / 39 MFMAs per SIMD, reference set 34 / 26 / 18 / 13); in snake order over the SIMDs -- every second group of four waves
                // reversed -- 47 / 47 / 44 / 44 and 25 / 22 / 22 / 22.
                const int wu = ((wv >> 2) & 1) ? (wv & ~3) + 3 - (wv & 3) : wv;
                if (p.mu_zero) mfma_units<NRT, TRI, true>(p, Ms, Pt, mus, qb, row0, ntl, wu, CH * WPC, lane, L.LDM, L.LDP);
                else mfma_units<NRT, TRI, false>(p, Ms, Pt, mus, qb, row0, ntl, wu, CH * WPC, lane, L.LDM, L.LDP);
            }
            DZ_W0STAMP(3 + 4 * phase); DZ_WSTAMP(3 + 4 * phase);
            __syncthreads();                                               // q visible
            DZ_W0STAMP(4 + 4 * phase); DZ_WSTAMP(4 + 4 * phase);
        }
        // ---- the chain's first wave: Metropolis step (:305-347), trace (core.py:114-116), record_history (:919-938); the others: the
        //      pre-tries of the next generation's proposal set
        if (sub == 0) {
            __builtin_amdgcn_s_setprio(2);
            const double* dc = dec + 8 * cl;
            const double u_acc = dc[1];
            const bool snk = dc[2] != 0.0;
            const int cr_idx = (int)dc[3];
            const double lpri = st[4 * cl], llik = st[4 * cl + 1], Tch = dc[7];
            const int sf = (int)st[4 * cl + 2]; const int sel = sf & 255; const bool fin = (sf & 256) != 0;
            double val = -__builtin_huge_val();
            if (lane < k) {
                val = sP[cl * k + lane] + Tch * sL[cl * k + lane];                                       // :279
                if (snk) val = val + sS[cl * k + lane];                                                  // :307
            } else if (lane >= 16 && lane < 16 + k) {
                const int i = lane - 16;
                if (i < k - 1) {
                    const int pt = (1 + i) * CH + cl;
                    double qt[NRT];
#pragma unroll
                    for (int t = 0; t < NRT; ++t) qt[t] = qb[pt * NRT + t];
                    double Q = 0.0;
#pragma unroll
                    for (int t = 0; t < NRT; ++t) Q = Q + qt[t];
                    val = Tch * nan_to_ninf(p.logF - 0.5 * Q) + rP[cl * (k - 1) + i];                     // :303
                } else val = Tch * llik + lpri;                                                          // :877-879
                if (snk) { const double sr = i < k - 1 ? rS[cl * (k - 1) + i] : 0.0; val = (val + sr) + sS[cl * k + i]; }   // :312-313
            }
            DZ_W0STAMP(14);
            double lu;
            double ratio = mt_log_ratio(k, val, u_acc, lane, &lu);             // log(u) of :993 rides in the ratio's logarithm pass
            if (!fin) ratio = -__builtin_huge_val();                           // DESIGN.md deviation D1
            const bool accept = is_finite(ratio) && (lu < ratio);              // :993
            DZ_W0STAMP(15);
            const int jj = 2 * lane;
            double* xr = Xs + cl * L.LDP;
            double2 xo = {0.0, 0.0};
            if (jj < d) xo.x = xr[jj];
            if (jj + 1 < d) xo.y = xr[jj + 1];
            double2 xn = xo;
            if (accept) { xn.x = jj < d ? region[jj] : 0.0; xn.y = jj + 1 < d ? region[jj + 1] : 0.0; }   // the selected proposal
            const bool moved = __any((xn.x != xo.x) || (xn.y != xo.y));        // core.py:120
            const double npri = accept ? sP[cl * k + sel] : lpri, nlik = accept ? sL[cl * k + sel] : llik;   // :345-347
            if (accept) { if (jj < d) xr[jj] = xn.x; if (jj + 1 < d) xr[jj + 1] = xn.y; }
            if (active) {
                if (jj < ld) {
                    if (last) gstore2(p.X + (size_t)c * ld + jj, xn);
                    if (trace_slot0 >= 0) gstore2(p.tX + ((size_t)c * p.tcap + (size_t)(trace_slot0 + gi)) * ld + jj, xn);
                    if (app) gstore2(p.Z + ((size_t)zappend + (Mc - M0) + gc) * ld + jj, xn);                         // record_history :933-936
                    if (publish) gstore2(publish + (RG ? (size_t)(g % (uint32_t)(pub.lag + 2)) * pub.pos_stride : (size_t)0) + (size_t)gc * ld + jj, xn);                          // set_current_position_arr :447-449
                }
                if (lane == 0) {
                    if (trace_slot0 >= 0) {
                        const size_t o = (size_t)(trace_slot0 + gi) * p.nl + c;
                        p.tlogp[o] = Tch * nlik + npri;                          // core.py:115; with a temperature ladder core.py:178
                        p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = cr_idx; p.tsnk[o] = snk ? 1 : 0;
                    }
                    if (last) { p.lprior[c] = npri; p.llike[c] = nlik; }
                }
            }
            if (lane == 0) { st[4 * cl] = npri; st[4 * cl + 1] = nlik; }
            __builtin_amdgcn_s_setprio(0);
        } else if (!last) {
            have_pre = !fn.snk;                                                // (wave-uniform)
            if (have_pre) make_pretries(dsn, 0, g + 1u, fn);
            DZ_WSTAMP(14);
        }
        DZ_W0STAMP(9); DZ_WSTAMP(9);
        __syncthreads();                                                   // the chain's other waves read the new state
        if (app) next_app += p.thin;
        Mc = Mn;
    }
}

}  // namespace dz
