// dz_megakernel.h -- k_generations: a whole run of generations in ONE launch.
//
// Between two history appends the chains share nothing (schedule S2): the Z archive is
// read-only and every chain's draws are addressed by (chain, generation).  So a block can own 16 chains and carry
// them through propose -> likelihood -> reference set -> likelihood -> Metropolis step for many generations
// without ever meeting another block:
//   * one block of 8 waves per CU owns 16 chains (each wave serves two of them, one after the other); the
//     16*k proposal points are exactly k MFMA tiles of 16 points and live in an LDS tile, never in HBM;
//   * the likelihood's matrix is staged in LDS once per launch; the 8 waves split the k*NRT
//     (point tile, row tile) units of the batched quadratic form on the FP64 matrix pipe;
//   * four block barriers per generation, no grid-wide synchronisation, no kernel boundaries.  HBM sees the
//     Z-row gathers, the chain state row, the trace row and (last generation of an appending segment) the
//     history row.
// Per chain the k rows of its LDS region are reused: after the selection the chosen proposal is moved to row 0
// and the k-1 reference points take rows 1..k-1, so the second likelihood pass runs over the same k tiles.
// 512 threads per block on purpose: the 1024-thread variant is capped at 128 VGPRs and spilled ~100 of them.
// Arithmetic is the multi-kernel path's, function for function (propose_set, mt_select_vals, mt_log_ratio,
// the MFMA contract), so results are bit-identical to it and to the oracle.
//
// Eligibility (checked on the host, everything else runs the multi-kernel path): MVN likelihood, ld <= 128,
// flat priors, no finite bounds, multitry >= 3, DEpairs = 1, draw slots <= 64, no position publishing (i.e. outside
// the crossover burn-in), LDS budget met.
#pragma once
#include "dz_kernels.h"

namespace dz {

constexpr int MEGA_CHAINS = 16;      // chains per block
constexpr int MEGA_WAVES = 16;       // waves per block

struct MegaLayout { int LDM, LDP, off_P, off_q, off_sP, off_sS, off_sL, off_rP, off_rS, off_mu, off_pr, off_st, total; };

__host__ __device__ inline MegaLayout mega_layout(int d, int k, int nrt, int ncr, int ngamma)
{
    MegaLayout L;
    L.LDM = d + 2;                    // matrix row (k index c): d entries + pad
    L.LDP = d + 1;                    // point row: odd stride keeps the A-layout reads (16 rows x 4 cols) off one bank
    L.off_P = d * L.LDM;
    L.off_q = L.off_P + MEGA_CHAINS * k * L.LDP;
    L.off_sP = L.off_q + MEGA_CHAINS * k * nrt;
    L.off_sS = L.off_sP + MEGA_CHAINS * k;
    L.off_sL = L.off_sS + MEGA_CHAINS * k;
    L.off_rP = L.off_sL + MEGA_CHAINS * k;
    L.off_rS = L.off_rP + MEGA_CHAINS * k;
    L.off_mu = L.off_rS + MEGA_CHAINS * k;
    L.off_pr = L.off_mu + d + 4;                 // crossover / gamma-level probabilities
    L.off_st = L.off_pr + ncr + ngamma;          // per chain: lprior, llike, sel|fin
    L.total = L.off_st + 3 * MEGA_CHAINS;
    return L;
}

// The (point tile, row tile) units of Y = V M^T for `ntiles` tiles of 16 points held in LDS; writes
// q[point][t] = butterfly16 over i of y_{16t+i} s_{16t+i}  (MVN contract, dz_kernels.h).
template <int NRT, bool TRI>
DZ_DEV void mfma_units(const Params& p, const double* __restrict__ Ms, const double* __restrict__ Pt, const double* __restrict__ mus,
                       double* __restrict__ qb, int ntiles, int wv, int l, int LDM, int LDP)
{
    const int d = p.d, KS = (d + 3) >> 2;
    const int pi = l & 15, kq = l >> 4;
    for (int u = wv; u < ntiles * NRT; u += MEGA_WAVES) {
        const int tile = u / NRT, t = u - tile * NRT;
        const double* arow = Pt + (size_t)(tile * 16 + pi) * LDP;
        const int r = 16 * t + pi;
        const bool rok = r < d;
        const double* bcol = Ms + r;
        dz_double4 acc = dz_double4{0.0, 0.0, 0.0, 0.0};
        int ks = TRI ? 4 * t : 0;
        // four k-steps per trip: the eight LDS reads are issued together, then the four MFMAs (ascending k)
        for (; ks + 4 <= KS; ks += 4) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * (ks + q) + kq;
                const bool ok = c < d;
                const int cs = ok ? c : 0;
                a[q] = ok ? arow[cs] - mus[cs] : 0.0;
                b[q] = (ok && rok) ? bcol[(size_t)cs * LDM] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[q], acc, 0, 0, 0);
        }
        for (; ks < KS; ++ks) {
            const int c = 4 * ks + kq;
            const bool ok = c < d;
            const int cs = ok ? c : 0;
            const double a = ok ? arow[cs] - mus[cs] : 0.0;
            const double b = (ok && rok) ? bcol[(size_t)cs * LDM] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int pt = tile * 16 + kq + 4 * e;
            const double y = acc[e];
            const double sv = TRI ? y : (rok ? Pt[(size_t)pt * LDP + r] - mus[r] : 0.0);
            const double q = bfly16(rok ? y * sv : 0.0);
            if (pi == 0) qb[pt * NRT + t] = q;
        }
    }
}

template <int NRT, bool TRI>
__global__ __launch_bounds__(64 * MEGA_WAVES) void k_generations(const Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M, int64_t trace_slot0, int append_last)
{
    const Params& p = *pp;       // read through the scalar cache on demand: keeps the ~70 fields out of the SGPR file
    constexpr int NCH = 1;
    constexpr int CPW = MEGA_CHAINS / MEGA_WAVES;      // chains per wave
    constexpr int NT = 64 * MEGA_WAVES;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int d = p.d, k = p.k, ld = p.ld;
    const MegaLayout L = mega_layout(d, k, NRT, p.ncr, p.ngamma);
    double* Ms = smem;
    double* Pt = smem + L.off_P;
    double* qb = smem + L.off_q;
    double* sP = smem + L.off_sP; double* sS = smem + L.off_sS; double* sL = smem + L.off_sL;
    double* rP = smem + L.off_rP; double* rS = smem + L.off_rS;
    double* mus = smem + L.off_mu;
    double* probs = smem + L.off_pr;
    double* st = smem + L.off_st;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;

    // ---- stage the matrix (rows c < d of Mt, columns r < d), mu, the selection probabilities, the chains' logp
    for (int i = threadIdx.x; i < d * L.LDM; i += NT) {
        const int row = i / L.LDM, col = i - row * L.LDM;
        Ms[i] = col < d ? p.Mt[(size_t)row * ld + col] : 0.0;
    }
    for (int i = threadIdx.x; i < MEGA_CHAINS * k * L.LDP; i += NT) Pt[i] = 0.0;
    if (threadIdx.x < d) mus[threadIdx.x] = p.mu[threadIdx.x];
    if (threadIdx.x < p.ncr) probs[threadIdx.x] = p.cr_probs[threadIdx.x];
    if (threadIdx.x < p.ngamma) probs[p.ncr + threadIdx.x] = p.g_probs[threadIdx.x];
    if (threadIdx.x < MEGA_CHAINS) {
        const int c = min(blockIdx.x * MEGA_CHAINS + (int)threadIdx.x, p.nl - 1);
        st[3 * threadIdx.x] = p.lprior[c]; st[3 * threadIdx.x + 1] = p.llike[c]; st[3 * threadIdx.x + 2] = 0.0;
    }
    __syncthreads();

    for (int gi = 0; gi < ngen; ++gi) {
        const uint32_t g = g0 + (uint32_t)gi;
        const bool last = gi == ngen - 1;
        for (int phase = 0; phase < 2; ++phase) {
            // ---- phase 0: k proposals around the chain's state (generate_proposal_points :258-264) into rows 0..k-1 of
            //      its LDS region; phase 1: the selected proposal moves to row 0 and k-1 reference points around it
            //      (:295-299) take rows 1..k-1.  One copy of the code serves both (instruction cache).
            for (int h = 0; h < CPW; ++h) {
                const int cl = wv * CPW + h;                                       // chain inside the block
                const int c = min(blockIdx.x * MEGA_CHAINS + cl, p.nl - 1);
                const uint32_t gc = (uint32_t)(p.off + c);
                DrawSrc ds; ds.have = true; ds.mine = make_uint4(0, 0, 0, 0);
                if (lane < p.nslots) { const u32x4 w = slot_counter_draw(p, lane, gc, g); ds.mine = make_uint4(w.x, w.y, w.z, w.w); }
                Ctrl u;
                {
                    const u32x4 w0 = uniform_draw(p, ds, 0, gc, g), w1 = uniform_draw(p, ds, 1, gc, g), w2 = uniform_draw(p, ds, 2, gc, g);
                    u.u_snk = u53(w0.x, w0.y); u.u_cr = u53(w0.z, w0.w); u.u_de = u53(w1.x, w1.y); u.u_glev = u53(w1.z, w1.w);
                    u.u_sel = u53(w2.x, w2.y); u.u_acc = u53(w2.z, w2.w);
                }
                const StepFlags f = step_flags_from(p, u, probs, probs + p.ncr);    // Dream.py:246-256
                double gt[NCH][2];
                load_gamma_row<NCH>(p, f.glev, 1, lane, gt);
                double* region = Pt + (size_t)(cl * k) * L.LDP;
                double base[NCH][2];
                if (phase == 0) load_row<NCH>(p.X + (size_t)c * ld, ld, lane, base);
                else {
                    // likelihoods of the chain's k points, mt_choose_proposal_pt (:291)
                    double lp = -__builtin_huge_val();
                    if (lane < k) {
                        const int pt = cl * k + lane;
                        double Q = 0.0;
                        for (int t = 0; t < NRT; ++t) Q = Q + qb[pt * NRT + t];
                        const double lk = nan_to_ninf(p.logF - 0.5 * Q);
                        sL[pt] = lk;
                        lp = sP[pt] + p.T * lk;
                    }
                    bool fin;
                    const int sel = mt_select_vals(k, lp, u.u_sel, lane, &fin);
                    if (lane == 0) st[3 * cl + 2] = (double)(sel | (fin ? 256 : 0));
                    const double* row = region + (size_t)sel * L.LDP;
                    base[0][0] = 2 * lane < d ? row[2 * lane] : 0.0; base[0][1] = 2 * lane + 1 < d ? row[2 * lane + 1] : 0.0;
                    if (2 * lane < d) region[2 * lane] = base[0][0];                 // the selected proposal now sits in row 0
                    if (2 * lane + 1 < d) region[2 * lane + 1] = base[0][1];
                }
                const int n = k - phase;
                propose_set<NCH, false, false, true>(p, phase, g, M, c, gc, 0, n, n, lane, base, gt, f.snk, f.cr_idx, 1, f.glev, ds,
                                                     region + (size_t)phase * L.LDP, L.LDP, (phase ? rS + cl * (k - 1) : sS + cl * k), nullptr,
                                                     (phase ? rP + cl * (k - 1) : sP + cl * k));
            }
            __syncthreads();                                                         // points visible
            mfma_units<NRT, TRI>(p, Ms, Pt, mus, qb, k, wv, lane, L.LDM, L.LDP);     // mt_evaluate_logps :278, :302
            __syncthreads();                                                         // q visible
        }
        // ---- Metropolis step (:305-347), trace (core.py:114-116), record_history (:919-938)
        for (int h = 0; h < CPW; ++h) {
            const int cl = wv * CPW + h;
            const int cg = blockIdx.x * MEGA_CHAINS + cl;
            const bool active = cg < p.nl;
            const int c = min(cg, p.nl - 1);
            const uint32_t gc = (uint32_t)(p.off + c);
            const u32x4 w2 = slot_counter_draw(p, 2, gc, g);
            const double u_acc = u53(w2.z, w2.w);
            const u32x4 w0 = slot_counter_draw(p, 0, gc, g);
            const bool snk = (p.snooker != 0.0) && (u53(w0.x, w0.y) < p.snooker);
            const int cr_idx = invcdf(probs, p.ncr, u53(w0.z, w0.w));
            const double lpri = st[3 * cl], llik = st[3 * cl + 1];
            const int sf = (int)st[3 * cl + 2]; const int sel = sf & 255; const bool fin = (sf & 256) != 0;
            double val = -__builtin_huge_val();
            if (lane < k) {
                val = sP[cl * k + lane] + p.T * sL[cl * k + lane];                                       // :279
                if (snk) val = val + sS[cl * k + lane];                                                  // :307
            } else if (lane >= 16 && lane < 16 + k) {
                const int i = lane - 16;
                if (i < k - 1) {
                    const int pt = cl * k + 1 + i;
                    double Q = 0.0;
                    for (int t = 0; t < NRT; ++t) Q = Q + qb[pt * NRT + t];
                    val = p.T * nan_to_ninf(p.logF - 0.5 * Q) + rP[cl * (k - 1) + i];                     // :303
                } else val = p.T * llik + lpri;                                                          // :877-879
                if (snk) { const double sr = i < k - 1 ? rS[cl * (k - 1) + i] : 0.0; val = (val + sr) + sS[cl * k + i]; }   // :312-313
            }
            double ratio = mt_log_ratio(k, val);
            if (!fin) ratio = -__builtin_huge_val();                                 // DESIGN.md deviation D1
            const bool accept = is_finite(ratio) && (dlog(u_acc) < ratio);           // :993
            const int jj = 2 * lane;
            double2 xo = {0.0, 0.0};
            if (jj < ld) xo = *reinterpret_cast<const double2*>(p.X + (size_t)c * ld + jj);
            double2 xn = xo;
            if (accept) {
                const double* row = Pt + (size_t)(cl * k) * L.LDP;                   // the selected proposal
                xn.x = jj < d ? row[jj] : 0.0; xn.y = jj + 1 < d ? row[jj + 1] : 0.0;
            }
            const bool moved = __any((xn.x != xo.x) || (xn.y != xo.y));              // core.py:120
            const double npri = accept ? sP[cl * k + sel] : lpri, nlik = accept ? sL[cl * k + sel] : llik;   // :345-347
            if (active) {
                if (jj < ld) {
                    if (accept) *reinterpret_cast<double2*>(p.X + (size_t)c * ld + jj) = xn;
                    if (trace_slot0 >= 0) *reinterpret_cast<double2*>(p.tX + ((size_t)(trace_slot0 + gi) * p.nl + c) * ld + jj) = xn;
                    if (last && append_last) *reinterpret_cast<double2*>(p.Z + ((size_t)M + gc) * ld + jj) = xn;      // record_history :933-936
                }
                if (lane == 0) {
                    st[3 * cl] = npri; st[3 * cl + 1] = nlik;
                    if (trace_slot0 >= 0) {
                        const size_t o = (size_t)(trace_slot0 + gi) * p.nl + c;
                        p.tlogp[o] = nlik + npri;                                    // core.py:115
                        p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = cr_idx; p.tsnk[o] = snk ? 1 : 0;
                    }
                    if (last) { p.lprior[c] = npri; p.llike[c] = nlik; }
                }
            }
        }
        // no barrier here: the next generation's first phase only touches each wave's own chains' rows and scalars,
        // and the shared q buffer is not written again before the next barrier
    }
}

}  // namespace dz
