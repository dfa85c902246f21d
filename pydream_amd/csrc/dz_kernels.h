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

constexpr int MAXK = 16;      // multitry limit
constexpr int MAXPAIR = 8;    // DEpairs limit

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
    // likelihood / prior
    const int32_t* pkind; const double *pa, *pb; int have_prior;
    const double *mu, *Mt; double logF; int tri; int J; const double* mixF;
};

struct StepFlags { bool snk; int cr_idx, delta, glev; };

DZ_DEV StepFlags step_flags(const Params& p, const Ctrl& u)
{
    StepFlags f;
    f.snk = (p.snooker != 0.0) && (u.u_snk < p.snooker);                      // set_snooker :542-554
    f.cr_idx = invcdf(p.cr_probs, p.ncr, u.u_cr);                              // set_CR :556-569
    f.delta = p.depairs > 1 ? 1 + (int)floor(u.u_de * (double)p.depairs) : 1;  // set_DEpair :571-583
    f.glev = 1 + invcdf(p.g_probs, p.ngamma, u.u_glev);                        // set_gamma_level :585-599
    return f;
}

// mt_choose_proposal_pt :883-917 (every lane computes the same scalars)
DZ_DEV int mt_select(const Params& p, int c, double u_sel, bool* anyfinite)
{
    double lp[MAXK];
    double mx = -__builtin_huge_val();
    bool fin = false;
    for (int i = 0; i < p.k; ++i) {
        lp[i] = p.p_prior[c * p.k + i] + p.T * p.p_like[c * p.k + i];
        if (i == 0 || lp[i] > mx) mx = lp[i];
        fin = fin || is_finite(lp[i]);
    }
    double S = 0.0;
    for (int i = 0; i < p.k; ++i) { lp[i] = dexp(lp[i] - mx); S = S + lp[i]; }
    double cum = 0.0; int sel = p.k - 1;
    for (int i = 0; i < p.k; ++i) { cum = cum + lp[i] / S; if (u_sel < cum) { sel = i; break; } }
    *anyfinite = fin;
    return sel;
}

DZ_DEV uint32_t mulhi_idx(uint32_t w, uint32_t M) { return (uint32_t)(((uint64_t)w * (uint64_t)M) >> 32); }

// ------------------------------------------------------------------------------------------
// generate_proposal_points :670-796 (+ snooker_update :798-837, sample_from_history :646-668,
// set_gamma :601-626).  grid: ceil(nc*n/4) blocks of 256; one wave per (chain, try).
// ------------------------------------------------------------------------------------------
template <int NCH>
DZ_DEV void propose_point(const Params& p, int phase, uint32_t g, uint32_t M, int c, int i, int n, int lane,
                          const double* __restrict__ base, double* __restrict__ out, double* slogp_out,
                          double* cur_snk_out, bool snk, int cr_idx, int delta, int glev)
{
    const int d = p.d, ld = p.ld;
    const uint32_t gc = (uint32_t)(p.off + c);
    const double CR = (double)(cr_idx + 1) / (double)p.ncr;                   // :146
    const uint32_t s_pt = stream_id(K_PT, (uint32_t)i, (uint32_t)phase), s_dim = stream_id(K_DIM, (uint32_t)i, (uint32_t)phase),
                   s_bnd = stream_id(K_BND, (uint32_t)i, (uint32_t)phase);
    double xb[NCH][2], pr[NCH][2];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        xb[it][0] = 0.0; xb[it][1] = 0.0;
        if (jj < ld) { const double2 t = *reinterpret_cast<const double2*>(base + jj); xb[it][0] = t.x; xb[it][1] = t.y; }
    }
    double slogp = 0.0;
    if (!snk) {
        uint32_t rows[2 * MAXPAIR];
        {   // random.sample(range(M), 2*delta) :662
            uint32_t sorted[2 * MAXPAIR];
            const int nidx = 2 * delta;
            u32x4 w = philox(p.k0, p.k1, 1, s_pt, gc, g);
            for (int t = 0; t < nidx; ++t) {
                if (t && (t & 3) == 0) w = philox(p.k0, p.k1, 1 + (t >> 2), s_pt, gc, g);
                const uint32_t word = (t & 3) == 0 ? w.x : (t & 3) == 1 ? w.y : (t & 3) == 2 ? w.z : w.w;
                uint32_t r = mulhi_idx(word, M - (uint32_t)t);
                int pos = 0;
                for (int s = 0; s < t; ++s) { if (r >= sorted[s]) { r++; pos = s + 1; } else break; }
                for (int s = t; s > pos; --s) sorted[s] = sorted[s - 1];
                sorted[pos] = r; rows[t] = r;
            }
        }
        double df[NCH][2];
#pragma unroll
        for (int it = 0; it < NCH; ++it) {    // chain_differences :692
            const int jj = 128 * it + 2 * lane;
            df[it][0] = 0.0; df[it][1] = 0.0;
            if (jj < ld) {
                double2 a = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[0] * ld + jj);
                double2 b = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[delta] * ld + jj);
                for (int t = 1; t < delta; ++t) {
                    const double2 a2 = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[t] * ld + jj);
                    const double2 b2 = *reinterpret_cast<const double2*>(p.Z + (size_t)rows[delta + t] * ld + jj);
                    a.x = a.x + a2.x; a.y = a.y + a2.y; b.x = b.x + b2.x; b.y = b.y + b2.y;
                }
                df[it][0] = a.x - b.x; df[it][1] = a.y - b.y;
            }
        }
        bool keep[NCH][2]; double e1[NCH][2], zt[NCH][2];
        int cnt = 0;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) {     // zeta, e, U :694-700
                const int j = 128 * it + 2 * lane + s;
                keep[it][s] = false; e1[it][s] = 0.0; zt[it][s] = 0.0;
                if (j < d) {
                    const u32x4 w = philox(p.k0, p.k1, (uint32_t)j, s_dim, gc, g);
                    keep[it][s] = u32d(w.x) < CR;
                    e1[it][s] = (-p.lamb + (p.lamb - (-p.lamb)) * u32d(w.y)) + 1.0;
                    zt[it][s] = p.zeta * (double)normal32(w.z, w.w);
                    cnt += keep[it][s] ? 1 : 0;
                }
            }
        const int dprime = wave_isum(cnt);                                     // :704 / :709
        const u32x4 wg = philox(p.k0, p.k1, 0, s_pt, gc, g);                   // set_gamma :615
        double gamma = 1.0;
        if (!(u53(wg.x, wg.y) < p.pgu))
            gamma = p.gtab[((size_t)(glev - 1) * p.depairs + (delta - 1)) * d + ((dprime == 0 ? d : dprime) - 1)];   // :624
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s = 0; s < 2; ++s) {     // :714 / :717, crossover :720-726
                double t = e1[it][s] * gamma; t = t * df[it][s];
                double q = xb[it][s] + t; q = q + zt[it][s];
                pr[it][s] = keep[it][s] ? q : xb[it][s];
            }
    } else {
        const u32x4 wg = philox(p.k0, p.k1, 0, stream_id(K_PT, 0, (uint32_t)phase), gc, g);
        const double gamma_s = 1.2 + (2.2 - 1.2) * u53(wg.z, wg.w);            // :618
        const u32x4 wi = philox(p.k0, p.k1, 1, s_pt, gc, g);                   // :808-810
        const uint32_t iz = mulhi_idx(wi.x, M), i1 = mulhi_idx(wi.y, M), i2 = mulhi_idx(wi.z, M);
        double v[NCH][2], dzz[NCH][2], zz[NCH][2];
        double accD = 0.0, accS = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it) {
            const int jj = 128 * it + 2 * lane;
            double2 z = {0.0, 0.0}, r1 = {0.0, 0.0}, r2 = {0.0, 0.0};
            if (jj < ld) {
                z = *reinterpret_cast<const double2*>(p.Z + (size_t)iz * ld + jj);
                r1 = *reinterpret_cast<const double2*>(p.Z + (size_t)i1 * ld + jj);
                r2 = *reinterpret_cast<const double2*>(p.Z + (size_t)i2 * ld + jj);
            }
            zz[it][0] = z.x; zz[it][1] = z.y;
            v[it][0] = xb[it][0] - z.x; v[it][1] = xb[it][1] - z.y;            // :813
            dzz[it][0] = r1.x - r2.x; dzz[it][1] = r1.y - r2.y;                // :819
#pragma unroll
            for (int s = 0; s < 2; ++s) if (jj + s < d) { accD = fma(v[it][s], v[it][s], accD); accS = fma(dzz[it][s], v[it][s], accS); }
        }
        const double D = wave_bfly(accD);                                      // :816 / :827
        if (n > 1) {
            const double cc = wave_bfly(accS) / D;                             // :820
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
        const double norm = sqrt(wave_bfly(accN));                             // :823 / :834
        slogp = norm != 0.0 ? dlog(norm) * (double)(d - 1) : 0.0;              // :824 / :835
        if (cur_snk_out && i == 0) { const double nc = sqrt(D); *cur_snk_out = nc != 0.0 ? dlog(nc) * (double)(d - 1) : 0.0; }   // :328-329
    }
    // hard boundaries :733-791
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        if (jj < ld) {
            if (p.hard) {
                const double2 mn = *reinterpret_cast<const double2*>(p.mins + jj);
                const double2 mxx = *reinterpret_cast<const double2*>(p.maxs + jj);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const double lo = s ? mn.y : mn.x, hi = s ? mxx.y : mxx.x;
                    double x = pr[it][s];
                    const bool bl = x < lo, bh = x > hi;
                    if (bl) x = 2 * lo - x;
                    if (bh) x = 2 * hi - x;
                    if (x < lo || x > hi) {
                        const u32x4 w = philox(p.k0, p.k1, (uint32_t)(jj + s), s_bnd, gc, g);
                        x = lo + u32d(w.x) * (hi - lo);
                    }
                    pr[it][s] = x;
                }
            }
            double2 o; o.x = (jj < d) ? pr[it][0] : 0.0; o.y = (jj + 1 < d) ? pr[it][1] : 0.0;
            *reinterpret_cast<double2*>(out + jj) = o;
        }
    }
    if (lane == 0) *slogp_out = slogp;
}

template <int NCH>
__global__ __launch_bounds__(256) void k_propose(Params p, int phase, uint32_t g, uint32_t M, int c0, int nc)
{
    const int n = phase == 0 ? p.k : p.k - 1;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nc * n) return;
    const int lane = threadIdx.x & 63;
    const int c = c0 + wave / n, i = wave % n;
    const Ctrl u = draw_ctrl(p.k0, p.k1, (uint32_t)(p.off + c), g);
    const StepFlags f = step_flags(p, u);
    const double* base; double* out; double* sl;
    if (phase == 0) { base = p.X + (size_t)c * p.ld; out = p.P + ((size_t)c * p.k + i) * p.ld; sl = p.p_slogp + (size_t)c * p.k + i; }
    else {
        bool fin; const int sel = mt_select(p, c, u.u_sel, &fin);
        base = p.P + ((size_t)c * p.k + sel) * p.ld; out = p.R + ((size_t)c * (p.k - 1) + i) * p.ld; sl = p.r_slogp + (size_t)c * (p.k - 1) + i;
    }
    propose_point<NCH>(p, phase, g, M, c, i, n, lane, base, out, sl, (phase == 0 && p.k == 1) ? p.cur_snk + c : nullptr,
                       f.snk, f.cr_idx, f.delta, f.glev);
}

// debug entry: flags supplied by the host (function-level parity tests)
template <int NCH>
__global__ __launch_bounds__(256) void k_propose_debug(Params p, int phase, uint32_t g, uint32_t M, int c, int n,
                                                      const double* base, double* out, double* sl, int snk, int cr_idx, int delta, int glev)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    propose_point<NCH>(p, phase, g, M, c, i, n, threadIdx.x & 63, base, out + (size_t)i * p.ld, sl + i, nullptr, snk != 0, cr_idx, delta, glev);
}

// ------------------------------------------------------------------------------------------
// Model.total_logp (model.py:17-32) on the device.
// ------------------------------------------------------------------------------------------
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
                if (kd == 1) { const double z = (x[it][s] - p.pa[j]) / p.pb[j]; t = (-(z * z) / 2.0 - 0.91893853320467274178) - dlog(p.pb[j]); }
                else if (kd == 2) t = (x[it][s] >= p.pa[j] && x[it][s] <= p.pa[j] + p.pb[j]) ? -dlog(p.pb[j]) : -__builtin_huge_val();
                acc = acc + t;
            }
        }
    return wave_bfly(acc);
}
DZ_DEV double nan_to_ninf(double x) { return x != x ? -__builtin_huge_val() : x; }

// MVN (examples/ndim_gaussian/dream_ex_ndim_gaussian.py:49-52), v1: one wave per point.
// y_r = sum_c M[r][c] v_c (ascending c, fma chain); Q = sum_r y_r s_r (ascending r, fma chain).
// Mt is M transposed, [d][ld], so that lanes (= rows r) read consecutive addresses.
template <int NCH>
__global__ __launch_bounds__(256) void k_logp_mvn(Params p, const double* __restrict__ pts, int npts, double* prior_out, double* like_out)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pt = blockIdx.x * 4 + wv;
    const int ptc = pt < npts ? pt : npts - 1;
    double* vs = lds + (size_t)wv * 2 * p.ld;
    double* ys = vs + p.ld;
    double x[NCH][2];
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        const int jj = 128 * it + 2 * lane;
        x[it][0] = 0.0; x[it][1] = 0.0;
        if (jj < p.ld) {
            const double2 t = *reinterpret_cast<const double2*>(pts + (size_t)ptc * p.ld + jj);
            x[it][0] = t.x; x[it][1] = t.y;
            vs[jj] = (jj < p.d) ? t.x - p.mu[jj] : 0.0;
            vs[jj + 1] = (jj + 1 < p.d) ? t.y - p.mu[jj + 1] : 0.0;
        }
    }
    const double prior = prior_of_point<NCH>(p, x, lane);
    __syncthreads();
    double y[2 * NCH];
#pragma unroll
    for (int rr = 0; rr < 2 * NCH; ++rr) y[rr] = 0.0;
    for (int c = 0; c < p.d; ++c) {
        const double vc = vs[c];
        const double* col = p.Mt + (size_t)c * p.ld;
#pragma unroll
        for (int rr = 0; rr < 2 * NCH; ++rr) {
            const int r = lane + 64 * rr;
            if (r < p.d && (!p.tri || r <= c)) y[rr] = fma(col[r], vc, y[rr]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 2 * NCH; ++rr) { const int r = lane + 64 * rr; if (r < p.ld) ys[r] = y[rr]; }
    __syncthreads();
    double Q = 0.0;
    for (int r = 0; r < p.d; ++r) { const double yr = ys[r]; Q = fma(yr, p.tri ? yr : vs[r], Q); }
    if (pt < npts && lane == 0) { prior_out[pt] = nan_to_ninf(prior); like_out[pt] = nan_to_ninf(p.logF - 0.5 * Q); }
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
__global__ __launch_bounds__(256) void k_accept(Params p, uint32_t g, int64_t M, int c0, int nc, int64_t trace_slot, int append, int publish)
{
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nc) return;
    const int lane = threadIdx.x & 63;
    const int c = c0 + wave, k = p.k, ld = p.ld;
    const uint32_t gc = (uint32_t)(p.off + c);
    const Ctrl u = draw_ctrl(p.k0, p.k1, gc, g);
    const StepFlags f = step_flags(p, u);
    const double last_prior = p.lprior[c], last_like = p.llike[c];
    const double last_logp = p.T * last_like + last_prior;                     // :243, :268
    double ratio; int sel = 0;
    if (k == 1) {
        const double q_logp = p.T * p.p_like[c] + p.p_prior[c];                // :274
        if (f.snk) ratio = nan_to_num((q_logp + p.p_slogp[c]) - (last_logp + p.cur_snk[c]));   // :326-332
        else ratio = nan_to_num(q_logp) - nan_to_num(last_logp);               // :334
    } else {
        bool fin; sel = mt_select(p, c, u.u_sel, &fin);
        double A[MAXK], B[MAXK];
        for (int i = 0; i < k; ++i) A[i] = p.p_prior[c * k + i] + p.T * p.p_like[c * k + i];                    // :279
        for (int i = 0; i < k - 1; ++i) B[i] = p.T * p.r_like[c * (k - 1) + i] + p.r_prior[c * (k - 1) + i];   // :303
        B[k - 1] = p.T * last_like + last_prior;                               // :877-879
        if (f.snk) {                                                           // :306-313
            for (int i = 0; i < k; ++i) {
                const double sp = p.p_slogp[c * k + i];
                const double sr = i < k - 1 ? p.r_slogp[c * (k - 1) + i] : 0.0;
                A[i] = A[i] + sp; B[i] = (B[i] + sr) + sp;
            }
        }
        double m2 = A[0];
        for (int i = 0; i < k; ++i) { if (A[i] > m2) m2 = A[i]; if (B[i] > m2) m2 = B[i]; }   // :320
        double SA = 0.0, SB = 0.0;
        for (int i = 0; i < k; ++i) SA = SA + dexp(A[i] - m2);                 // :321
        for (int i = 0; i < k; ++i) SB = SB + dexp(B[i] - m2);                 // :322
        ratio = nan_to_num(dlog(SA / SB));                                     // :323
        if (!fin) ratio = -__builtin_huge_val();                               // DESIGN.md deviation D1 (:282-289)
    }
    const bool accept = is_finite(ratio) && (dlog(u.u_acc) < ratio);           // :993
    const double* src = p.P + ((size_t)c * k + sel) * ld;
    double* xrow = p.X + (size_t)c * ld;
    bool diff = false;
    double xn[NCH][2];
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
            if (trace_slot >= 0) *reinterpret_cast<double2*>(p.tX + ((size_t)trace_slot * p.nl + c) * ld + jj) = t;
            if (append) *reinterpret_cast<double2*>(p.Z + ((size_t)M + gc) * ld + jj) = t;       // :933-936
            if (publish) *reinterpret_cast<double2*>(p.cp_new + (size_t)gc * ld + jj) = t;       // :447-449
        }
    }
    if (lane == 0) {
        p.lprior[c] = npri; p.llike[c] = nlik;
        if (trace_slot >= 0) {
            const size_t o = (size_t)trace_slot * p.nl + c;
            p.tlogp[o] = nlik + npri;                                          // core.py:115
            p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = f.cr_idx; p.tsnk[o] = f.snk ? 1 : 0;
        }
    }
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
__global__ void k_strip_partial(const double* __restrict__ pos, int N, int d, int ld, const double* __restrict__ mean, int pass, double* __restrict__ partial)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (j >= d) return;
    const int r0 = s * 64, r1 = min(N, r0 + 64);
    double ps = 0.0;
    if (pass == 0) for (int c = r0; c < r1; ++c) ps = ps + pos[(size_t)c * ld + j];
    else { const double m = mean[j]; for (int c = r0; c < r1; ++c) { const double t = pos[(size_t)c * ld + j] - m; ps = fma(t, t, ps); } }
    partial[(size_t)s * ld + j] = ps;
}
// pass 0: mean[j] = (sum_s partial)/N ; pass 1: sd[j] = sqrt((sum_s partial)/N), sdc = sd with 0 -> 1e-12 (:479)
__global__ void k_strip_finish(const double* __restrict__ partial, int nstrips, int N, int d, int ld, int pass, double* __restrict__ mean, double* __restrict__ sd, double* __restrict__ sdc)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    double tot = 0.0;
    for (int s = 0; s < nstrips; ++s) tot = tot + partial[(size_t)s * ld + j];
    if (pass == 0) mean[j] = tot / (double)N;
    else { const double v = sqrt(tot / (double)N); sd[j] = v; sdc[j] = v == 0.0 ? 1e-12 : v; }
}

// one wave per GLOBAL chain: bins and normalised squared jumps (:481, :527)
template <int NCH>
__global__ __launch_bounds__(256) void k_jump(Params p, uint32_t g, const double* __restrict__ sdc, const double* __restrict__ sdg,
                                              double* __restrict__ dl, double* __restrict__ dlg, int* __restrict__ binc, int* __restrict__ bing)
{
    const int gcn = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gcn >= p.N) return;
    const int lane = threadIdx.x & 63;
    const Ctrl u = draw_ctrl(p.k0, p.k1, (uint32_t)gcn, g);
    const StepFlags f = step_flags(p, u);
    // np.any(self.gamma == 1.0) of the LAST generate_proposal_points call (:371, :705/:730)
    bool gu = false;
    if (!f.snk) {
        const int phase = p.k > 1 ? 1 : 0, n = p.k > 1 ? p.k - 1 : 1;
        for (int i = 0; i < n; ++i) {
            const u32x4 w = philox(p.k0, p.k1, 0, stream_id(K_PT, (uint32_t)i, (uint32_t)phase), (uint32_t)gcn, g);
            gu = gu || (u53(w.x, w.y) < p.pgu);
        }
    }
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
                const double t = df / sdc[jj + s]; accC = fma(t, t, accC);
                const double t2 = df / sdg[jj + s]; accG = fma(t2, t2, accG);
            }
        }
    }
    const double dC = nan_to_num(wave_bfly(accC)), dG = nan_to_num(wave_bfly(accG));
    if (lane == 0) {
        binc[gcn] = do_c ? (f.snk ? p.ncr - 1 : f.cr_idx) : -1;               // :374-378
        bing[gcn] = do_g ? f.glev - 1 : -1;
        dl[gcn] = dC; dlg[gcn] = dG;
    }
}

// single block: thread m < ncr updates crossover bin m, thread ncr+m updates gamma bin m; then renormalise
__global__ void k_adapt_update(Params p, const double* __restrict__ dl, const double* __restrict__ dlg, const int* __restrict__ binc, const int* __restrict__ bing)
{
    __shared__ int any[2];
    const int t = threadIdx.x;
    if (t < 2) any[t] = 0;
    __syncthreads();
    if (t < p.ncr + p.ngamma) {
        const bool isg = t >= p.ncr; const int m = isg ? t - p.ncr : t;
        const double* dd = isg ? dlg : dl; const int* bb = isg ? bing : binc;
        double tot = 0.0; int cnt = 0;
        for (int s = 0; s < p.N; s += 64) {
            double ps = 0.0;
            const int e = min(p.N, s + 64);
            for (int c = s; c < e; ++c) if (bb[c] == m) { ps = ps + dd[c]; cnt++; }
            tot = tot + ps;
        }
        if (cnt) {
            double* delta = isg ? p.g_delta : p.cr_delta; double* n = isg ? p.g_n : p.cr_n;
            delta[m] = delta[m] + tot; n[m] += (double)cnt; atomicOr(&any[isg ? 1 : 0], 1);
        }
    }
    __syncthreads();
    if (t < 2 && any[t]) {     // :487-493 / :531-536
        const int nb = t ? p.ngamma : p.ncr;
        double* probs = t ? p.g_probs : p.cr_probs; const double* delta = t ? p.g_delta : p.cr_delta; const double* n = t ? p.g_n : p.cr_n;
        bool all = true;
        for (int m = 0; m < nb; ++m) if (delta[m] == 0.0) all = false;
        if (all) {
            double S = 0.0;
            for (int m = 0; m < nb; ++m) { probs[m] = (delta[m] / n[m]) * (double)p.N; S = S + probs[m]; }
            for (int m = 0; m < nb; ++m) probs[m] = probs[m] / S;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Gelman_Rubin (convergence.py:3-20) from the device-resident trace
// ------------------------------------------------------------------------------------------
__global__ void k_chain_moments(const double* __restrict__ tX, int nl, int d, int ld, int nsamples, double* __restrict__ mean, double* __restrict__ var)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (j >= d) return;
    const int nb = nsamples / 2, n2 = nsamples - nb;
    const double* x = tX + ((size_t)nb * nl + c) * ld + j;
    const size_t st = (size_t)nl * ld;
    double s = 0.0;
    for (int t = 0; t < n2; ++t) s = s + x[(size_t)t * st];
    const double m = s / (double)n2;
    double v = 0.0;
    for (int t = 0; t < n2; ++t) { const double q = x[(size_t)t * st] - m; v = v + q * q; }
    mean[(size_t)c * d + j] = m; var[(size_t)c * d + j] = v / (double)n2;
}
__global__ void k_rhat(const double* __restrict__ mean, const double* __restrict__ var, int nch, int d, int nsamples, double* __restrict__ rhat)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d) return;
    double W = 0.0, mm = 0.0;
    for (int c = 0; c < nch; ++c) { W = W + var[(size_t)c * d + j]; mm = mm + mean[(size_t)c * d + j]; }
    W = W / (double)nch; mm = mm / (double)nch;
    double B = 0.0;
    for (int c = 0; c < nch; ++c) { const double q = mean[(size_t)c * d + j] - mm; B = B + q * q; }
    B = B / (double)nch;
    const double var_est = W * (1.0 - 1.0 / (double)nsamples) + B;
    rhat[j] = sqrt(var_est / W);
}

}  // namespace dz
