// dz_megakernel.h -- k_generations: a whole run of generations in ONE launch.
//
// Between two history appends the chains share nothing (schedule S2): the Z archive is
// read-only and every chain's draws are addressed by (chain, generation).  So a block can own 16 chains and carry
// them through propose -> likelihood -> reference set -> likelihood -> Metropolis step for many generations
// without ever meeting another block:
//   * one block of 16 waves per CU owns 16 chains, one wave per chain; try i of the 16 chains is point tile i
//     (16 points = one MFMA tile), held in LDS, never in HBM;
//   * the likelihood's matrix is staged in LDS once per launch (the triangular factor packed, tri_row_offset); the
//     16 waves share the (point tile, row tile) units of the batched quadratic form on the FP64 matrix pipe, dealt
//     out heaviest first in snake order so that waves and SIMDs carry the same number of k-steps;
//   * with the packed triangle there is room for the chains' states, the gamma table and each chain's control
//     decisions in LDS as well (XLDS): HBM then sees only the Z-row gathers, the trace rows and (last generation of
//     an appending segment) the history rows;
//   * four block barriers per generation, no grid-wide synchronisation, no kernel boundaries.
// After the selection the chosen proposal is moved to the chain's row of tile 0 and the k-1 reference points take
// tiles 1..k-1; the second likelihood pass covers only those (the chosen proposal's density is already known).
// Arithmetic is the multi-kernel path's, function for function (propose_set, mt_select_vals, mt_log_ratio,
// the MFMA contract), so results are bit-identical to it and to the oracle.
//
// Eligibility (checked on the host, everything else runs the multi-kernel path): MVN likelihood, ld <= 128, multitry 1 or 3..15 (draw
// slots <= 64), a block size whose LDS layout fits (16, 8 or 4 chains).  Priors / hard boundaries / DEpairs > 1 / redraw rounds: the PB
// instantiations.  Inside the crossover burn-in and under parallel tempering a launch covers ONE generation (positions published, the
// block's adaptation sums made, the previous generation's totals applied in the prologue; the swap kernel follows).
#pragma once
#include "dz_kernels.h"

namespace dz {

constexpr int DZ_MAX_REDRAWS_DEV = 64;                                  // == DZ_MAX_REDRAWS (include/dreamzs.h; checked in dz_engine.hip)
constexpr unsigned long long DZ_REDRAW_KEY_STEP_DEV = 0x9E3779B97F4A7C15ull;   // == DZ_REDRAW_KEY_STEP
constexpr int MEGA_CHAINS = 16;      // chains (= waves) per block at full size; 8 or 4 when there are too few chains to give every CU a block

struct MegaLayout { int LDM, LDP, rows, off_P, off_q, off_sP, off_sS, off_sL, off_rP, off_rS, off_mu, off_pr, off_st, off_dec, off_gt, off_X, off_pc, pcn, off_Xo, off_tab, total; };

// point rows of a block: try i of chain c at row i*ch + c; tiles are 16 consecutive rows, from row 0 (k tries) or from
// row ch (the k-1 reference tries); rows past the last point stay zero
__host__ __device__ inline int mega_rows(int k, int ch)
{
    const int a = 16 * ((k * ch + 15) / 16), b = ch + 16 * (((k - 1) * ch + 15) / 16);
    return a > b ? a : b;
}
// pb: room for the per-dimension prior / boundary constants of the full-code instantiations (PBConsts: five double arrays and one int
// array of pcn = 16 nrt entries)
// xo: room for the chains' states as they were at the start of the launch (crossover burn-in: the block's adaptation sums need the jumps)
// nomat: the matrix is NOT staged (k_generations_d2, 128 < d <= 256: it does not fit next to the point tiles and is read from L2)
// ntab: rows of the table of crossover / gamma-level probabilities per generation of the launch (adapt_lag >= 1: several burn-in generations per
// launch, the MG instantiations; rows of (ncr + ngamma + 1) & ~1 doubles)
// sp (k_generations_d2<.., SP>, round 6): the proposal set goes through the point tiles in two passes (tries 0 .. k-2, then the last one), the selected
// proposal stays in registers: k - 1 tries' rows instead of k
__host__ __device__ inline MegaLayout mega_layout(int d, int k, int nrt, int ncr, int ngamma, bool tri, bool xlds, int ch = MEGA_CHAINS, bool pb = false, bool xo = false, bool nomat = false, int ntab = 0,
                                                  bool sp = false)
{
    MegaLayout L;
    const int ks4 = 4 * ((d + 3) / 4);
    L.LDM = d + 2;                    // dense matrix row (k index c): d entries + pad; rows d..ks4-1 are zero
    L.LDP = ks4 + 1;                  // point row: zero padded to the k-steps; odd stride keeps the A-layout reads (16 rows x 4 cols) off one bank
    L.off_P = nomat ? 0 : (tri ? tri_row_offset(ks4) : ks4 * L.LDM + 16);       // (+16: the last row tile's column reads run past the last row's end)
    L.rows = sp ? 16 * (((k - 1) * ch + 15) / 16) : mega_rows(k, ch);
    L.off_q = L.off_P + L.rows * L.LDP;
    L.off_sP = L.off_q + L.rows * nrt;
    L.off_sS = L.off_sP + ch * k;
    L.off_sL = L.off_sS + ch * k;
    L.off_rP = L.off_sL + ch * k;
    L.off_rS = L.off_rP + ch * k;
    L.off_mu = L.off_rS + ch * k;
    L.off_pr = L.off_mu + ks4 + 4;               // (mu zero padded to the k-steps) crossover / gamma-level probabilities
    L.off_st = L.off_pr + ncr + ngamma + 2 - ((ncr + ngamma) & 1);      // (the offsets behind stay even; one word of the gap: the log prior of a point inside every uniform support, PBConsts::inside) per chain: lprior, llike, sel|fin
    L.off_dec = L.off_st + 4 * ch;      // per chain and generation: u_sel, u_acc, snooker, CR index, gamma level
    L.off_gt = L.off_dec + 8 * ch;      // gamma_arr[level-1][0][:]
    L.off_X = L.off_gt + ngamma * d + (d & 1);   // chain states (XLDS)
    L.total = L.off_X + (xlds ? ch * L.LDP : 0);
    L.total += L.total & 1;
    L.off_pc = L.total; L.pcn = 16 * nrt;
    if (pb) L.total += 5 * L.pcn + L.pcn / 2;
    L.total += L.total & 1;
    L.off_Xo = L.total;
    if (xo) L.total += ch * L.LDP + ((ch * L.LDP) & 1);
    L.off_tab = L.total;
    L.total += ntab * ((ncr + ngamma + 1) & ~1);
    return L;
}

// Unit u (0 .. ntl*NRT-1) in dealing order: row tile first (heaviest first for the triangular factor), then point tile.
// Wave w of tw takes the units w, 2tw-1-w, 2tw+w, ... (snake), which evens out the k-steps per wave and per SIMD.
DZ_DEV int mega_unit(int j, int wv, int tw) { return (j & 1) ? tw * j + (tw - 1 - wv) : tw * j + wv; }

// One accumulator's chain over the whole batches of a unit that starts at batch T0 (the packed triangle: row tile t joins at k = 16 t),
// as a software pipeline over half batches (two k-steps): the operands of half batch h + 1 are requested before the MFMAs of h are
// issued, so that a wave that is alone on its SIMD does not wait an LDS round trip in front of every pair of MFMAs.  The start is a
// template parameter (one straight-line body per starting batch: the LDS offsets stay immediates and the waitcnt counts exact); KB is
// NRT - 1 or NRT (d mod 16 in 1..12 or not), so only the last batch is conditional.  Same MFMAs in the same ascending k order.
template <int NRT, bool TRI, bool MZ, int T0>
DZ_DEV void unit_batches(const double* __restrict__ ap, const double* __restrict__ bp, const double* __restrict__ mp, int kq, bool lastb, int LDM,
                         dz_double4& acc)
{
    constexpr int H0 = 2 * T0, HN = 2 * NRT;
    double A[2][2], B[2][2];
    auto ldh = [&](int h, double (&a2)[2], double (&b2)[2]) {
        const int b16 = h >> 1, q0 = 2 * (h & 1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = 16 * b16 + 4 * (q0 + q);
            b2[q] = MZ ? bp[c] : bp[c] - mp[c];
            a2[q] = TRI ? ap[128 * b16 * (b16 + 1) + (4 * (q0 + q) + kq) * 16 * (b16 + 1)] : ap[(size_t)(c + kq) * LDM];
        }
    };
#pragma unroll
    for (int h = H0; h < HN; ++h) {                                // (constant trip count, no early exit: it has to unroll)
        if (h < HN - 2 || lastb) {
            if (h == H0) ldh(h, A[h & 1], B[h & 1]);
            if (h + 1 < HN && (h + 1 < HN - 2 || lastb)) ldh(h + 1, A[(h + 1) & 1], B[(h + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[h & 1][0], B[h & 1][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[h & 1][1], B[h & 1][1], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int NRT, bool TRI, bool MZ, int T>
DZ_DEV void unit_dispatch(int t, const double* __restrict__ ap, const double* __restrict__ bp, const double* __restrict__ mp, int kq, bool lastb, int LDM,
                          dz_double4& acc)
{
    if (t == T) unit_batches<NRT, TRI, MZ, T>(ap, bp, mp, kq, lastb, LDM, acc);
    else if constexpr (T + 1 < NRT) unit_dispatch<NRT, TRI, MZ, T + 1>(t, ap, bp, mp, kq, lastb, LDM, acc);
}

// The (point tile, row tile) units of Y^T = M V^T for the ntl point tiles of 16 rows starting at row row0 of the LDS point area; writes
// q[point][t], the row-tile sum of the MVN contract (dz_kernels.h tile_q_*).  Transposed tile: the matrix is the MFMA's A operand
// (lane l: row 16 t + l%16, k = 4 ks + l/16), the points are its B operand (lane l: point l%16, same k), so a lane ends up with four
// rows of ONE point and the row-tile sum needs two lane-swap steps instead of four butterfly stages per accumulator element.
// wv, tw, ntl and row0 are wave-uniform (the kernel passes its wave number through readfirstlane): the unit index, the row tile and
// every loop bound live in scalar registers, and with the block index b16 unrolled the LDS addresses are one lane offset per unit plus
// immediates.
template <int NRT, bool TRI, bool MZ>
DZ_DEV void mfma_units(const Params& p, const double* __restrict__ Ms, const double* __restrict__ Pt, const double* __restrict__ mus,
                       double* __restrict__ qb, int row0, int ntl, int wv, int tw, int l, int LDM, int LDP)
{
    const int d = p.d, KS = (d + 3) >> 2, KB = KS >> 2;            // KB whole batches of four k-steps (one 16-row block of the matrix)
    const int pi = l & 15, kq = l >> 4;
    const int nun = ntl * NRT;
    const int rcp = ((1 << 20) + ntl - 1) / ntl;                  // u / ntl == (u * rcp) >> 20 for every u < 2^10
    for (int j = 0; tw * j < nun; ++j) {
        const int u = mega_unit(j, wv, tw);
        if (u >= nun) continue;
        const int t = (u * rcp) >> 20, trow = row0 + 16 * (u - t * ntl);           // the tile's first point row
        const double* bp = Pt + (size_t)(trow + pi) * LDP + kq;                    // B operand: point trow + pi, k = 4 ks + kq
        const double* mp = mus + kq;
        // A operand: matrix row r = 16 t + pi of k-row c = 4 ks + kq; packed triangle: k-row c of block b = c / 16 starts at
        // tri_row_offset(c) = 128 b (b + 1) + (c - 16 b) 16 (b + 1)
        const double* ap = Ms + 16 * t + pi;
        dz_double4 acc = dz_double4{0.0, 0.0, 0.0, 0.0};
        // No predicates: point rows and mu are zero padded to the k-steps and the matrix rows c >= d are zero, so out-of-range k
        // terms add exact zeros; output rows r >= d of the packed triangle are exact zeros, those of the dense square read whatever
        // sits there, stay inside their own accumulator rows and are dropped by tile_q.
        // the whole batches b16 = (TRI ? t : 0) .. KB - 1, operand reads one half batch ahead of the MFMAs (unit_batches)
        if (TRI) unit_dispatch<NRT, TRI, MZ, 0>(t, ap, bp, mp, kq, KB == NRT, LDM, acc);
        else unit_batches<NRT, TRI, MZ, 0>(ap, bp, mp, kq, KB == NRT, LDM, acc);
        for (int ks = 4 * KB; ks < KS; ++ks) {                     // the last, partial block: every row tile takes part (ks >= 4 (NRT - 1) >= 4 t)
            const int c = 4 * ks;
            const double av = TRI ? ap[128 * KB * (KB + 1) + (c - 16 * KB + kq) * 16 * (KB + 1)] : ap[(size_t)(c + kq) * LDM];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, MZ ? bp[c] : bp[c] - mp[c], acc, 0, 0, 0);
        }
        double qv;
        if (TRI) qv = tile_q_tri(acc);
        else {
            double sv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = min(16 * t + kq + 4 * e, d - 1);
                sv[e] = MZ ? Pt[(size_t)(trow + pi) * LDP + r] : Pt[(size_t)(trow + pi) * LDP + r] - mus[r];
            }
            qv = tile_q(acc, sv, 16 * t + kq, d);
        }
        if (kq == 0) qb[(trow + pi) * NRT + t] = qv;
    }
}

// The likelihood units of k_generations_d2 (triangular factor; cf. mfma_units, dz_megakernel.h): a unit is a ROW tile and a PAIR of point
// tiles, so that every A operand fetched from L2 (16 matrix rows x 4 k) feeds two MFMAs -- half the L2 reads and half the load
// instructions of one unit per point tile; an odd last tile runs alone.  Each accumulator is still ONE chain over k in ascending order
// (the contract), so the row-tile sums are the bits mfma_units makes.  Units are dealt heaviest first (row tile 0 walks
// all of k) in snake order over the waves.
template <int NRT, bool MZ, int NPT>
DZ_DEV void d2_unit(const double* __restrict__ ap, const double* __restrict__ bp0, const double* __restrict__ bp1, const double* __restrict__ mp,
                    int t, int KS, int KB, int kq, dz_double4& acc0, dz_double4& acc1)
{
#pragma unroll 1
    for (int b16 = t; b16 < KB; ++b16) {                           // (not unrolled: b16 is wave-uniform, the offsets are scalar arithmetic; unrolled, the compiler
        double a[4], b0[4], b1[4];                                 //  hoisted the L2 loads of many batches and spilled ninety registers)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                              // the batch's operand reads are issued together, then its MFMAs (ascending k)
            const int c = 16 * b16 + 4 * q;
            a[q] = ap[128 * b16 * (b16 + 1) + (4 * q + kq) * 16 * (b16 + 1)];
            b0[q] = MZ ? bp0[c] : bp0[c] - mp[c];
            if (NPT == 2) b1[q] = MZ ? bp1[c] : bp1[c] - mp[c];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b0[q], acc0, 0, 0, 0);
            if (NPT == 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b1[q], acc1, 0, 0, 0);
        }
    }
    for (int ks = 4 * KB; ks < KS; ++ks) {                         // the last, partial block: every row tile takes part
        const int c = 4 * ks;
        const double av = ap[128 * KB * (KB + 1) + (c - 16 * KB + kq) * 16 * (KB + 1)];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, MZ ? bp0[c] : bp0[c] - mp[c], acc0, 0, 0, 0);
        if (NPT == 2) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, MZ ? bp1[c] : bp1[c] - mp[c], acc1, 0, 0, 0);
    }
}
template <int NRT, bool MZ>
DZ_DEV void mfma_units_d2(const Params& p, const double* __restrict__ Mg, const double* __restrict__ Pt, const double* __restrict__ mus,
                          double* __restrict__ qb, int row0, int ntl, int wv, int tw, int l, int LDP)
{
    const int d = p.d, KS = (d + 3) >> 2, KB = KS >> 2;
    const int pi = l & 15, kq = l >> 4;
    const int npr = (ntl + 1) >> 1, nun = npr * NRT;
    const int rcp = ((1 << 20) + npr - 1) / npr;                  // u / npr == (u * rcp) >> 20 for every u < 2^10
    for (int j = 0; tw * j < nun; ++j) {
        const int u = mega_unit(j, wv, tw);
        if (u >= nun) continue;
        const int t = (u * rcp) >> 20, pr = u - t * npr, trow = row0 + 32 * pr;
        const bool two = 2 * pr + 1 < ntl;                        // (wave-uniform)
        const double* bp0 = Pt + (size_t)(trow + pi) * LDP + kq;
        const double* bp1 = bp0 + (size_t)16 * LDP;
        const double* mp = mus + kq;
        const double* ap = Mg + 16 * t + pi;
        dz_double4 acc0 = dz_double4{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
        if (two) d2_unit<NRT, MZ, 2>(ap, bp0, bp1, mp, t, KS, KB, kq, acc0, acc1);
        else d2_unit<NRT, MZ, 1>(ap, bp0, bp1, mp, t, KS, KB, kq, acc0, acc1);
        const double q0 = tile_q_tri(acc0);
        if (kq == 0) qb[(trow + pi) * NRT + t] = q0;
        if (two) {
            const double q1 = tile_q_tri(acc1);
            if (kq == 0) qb[(trow + 16 + pi) * NRT + t] = q1;
        }
    }
}

// ---- archive-row prefetch of the persistent kernels.  Measured with cycle stamps (tools/stamps.py): with the rows of try i + 1
// requested at the start of try i, a try of ~2.8 k cycles still waited 1-2 k cycles for them -- the gathers are latency-bound (two
// 1-KB requests in flight per wave), not bandwidth-bound.  So three row buffers rotate (the rows of tries i + 1 and i + 2 are in
// flight during try i; the loop is unrolled by three so that no buffer is ever copied -- a copy would wait for its load), and the
// first two tries of a phase are requested before the PREVIOUS phase's barrier and likelihood pass: row indices depend on
// (chain, generation, phase, try) only, never on the selected point.
struct RowPair { double2 a, b; };

// The two row numbers of a DE try, random.sample(range(M), 2) (:662), from the words of its idx-1 slot
DZ_DEV void sample_pair(uint32_t wx, uint32_t wy, uint32_t M, uint32_t& r0, uint32_t& r1)
{
    r0 = mulhi_idx(wx, M);
    r1 = mulhi_idx(wy, M - 1u);
    r1 += (r1 >= r0) ? 1u : 0u;
}
// What the tries read from the generation's slot draws, made lane-parallel once per generation instead of on the scalar unit once per
// try (DrawSrc::xf): lane s holds slot s.  snk: the chain's move of this generation (wave-uniform).
DZ_DEV void finish_draws(const Params& p, DrawSrc& q, bool snk, uint32_t M, int lane)
{
    const int sl = lane - 3;
    const bool pt = lane >= 3 && lane < p.nslots;
    const bool rows = pt && (sl % p.npt) == 1;            // (DEpairs = 1: npt = 2 -- idx 0 the gamma draws, idx 1 the rows)
    const bool gam = pt && (sl % p.npt) == 0;
    uint32_t r0, r1;
    sample_pair(q.mine.x, q.mine.y, M, r0, r1);
    const uint32_t s0 = mulhi_idx(q.mine.x, M), s1 = mulhi_idx(q.mine.y, M), s2 = mulhi_idx(q.mine.z, M);      // :808-810
    const uint32_t gu = u53_below(q.mine.x, q.mine.y, p.pgu_thr) ? 1u : 0u;                                  // :615
    q.mine.x = rows ? (snk ? s0 : r0) : (gam ? gu : q.mine.x);
    q.mine.y = rows ? (snk ? s1 : r1) : q.mine.y;
    q.mine.z = (rows && snk) ? s2 : q.mine.z;
    q.xf = true;
}

template <bool XF>
DZ_DEV void request_pair(const Params& p, const DrawSrc& ds, int slot, uint32_t gc, uint32_t g, uint32_t M, int lane, RowPair& R, const double* Zb, uint32_t ldb)
{   // slot = pt_slot(p, phase, i, 1); Zb / ldb: the archive and its row pitch in BYTES (fetched once per set, SetConsts)
    const u32x4 w = uniform_draw(p, ds, slot, gc, g);
    uint32_t r0, r1;
    if (XF) { r0 = w.x; r1 = w.y; }                        // (finish_draws)
    else {
        r0 = __builtin_amdgcn_readfirstlane(mulhi_idx(w.x, M));
        r1 = __builtin_amdgcn_readfirstlane(mulhi_idx(w.y, M - 1u));
        if (r1 >= r0) r1++;                                // random.sample(range(M), 2) :662  (wave-uniform: kept on the scalar unit)
    }
    // (no lane predicate: lanes past ld re-read the row's last pair, masked where used)  scalar row base + 32-bit lane offset in bytes
    const uint32_t jb = (uint32_t)min(16 * lane, (int)ldb - 16);
    R.a = gload2(reinterpret_cast<const double*>(reinterpret_cast<const char*>(Zb) + (uint64_t)r0 * ldb + jb));
    R.b = gload2(reinterpret_cast<const double*>(reinterpret_cast<const char*>(Zb) + (uint64_t)r1 * ldb + jb));
}

// DE tries i0..i1-1 of one chain's set (one pair, the common case); A and B already hold the rows of tries i0 and i0 + 1.
template <int LEAN, bool XF>
DZ_DEV void propose_de_pf(const Params& p, int phase, uint32_t g, uint32_t M, int c, uint32_t gc, int i0, int i1, int n, int lane,
                          const double (&xb)[1][2], const double* __restrict__ grow, int cr_idx, int glev, const DrawSrc& ds,
                          double* out, int out_stride, double* sl, double* prior_out, RowPair& A, RowPair& B, RowPair& C, const PBConsts* pc = nullptr)
{
    const SetConsts sc = set_consts(p, phase, cr_idx);
    // (the tries' prior butterflies batched three at a time through wave_bfly4 -- lane partial sums kept across a round -- made the
    //  full-code kernel spill twice as much, 88 -> 180 bytes per lane, and cost 18 %: not kept)
    auto body = [&](int i, const RowPair& R) {
        RowTerms<1> rt;
        rt.a[0][0] = R.a.x - R.b.x; rt.a[0][1] = R.a.y - R.b.y; rt.b[0][0] = 0.0; rt.b[0][1] = 0.0;       // chain_differences :692
        if (!LEAN && pc) {      // full code, constants in LDS: the prior is evaluated on the values the lane has just stored
            double pv[1][2];
            propose_point<1, false, LEAN>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * out_stride, nullptr, false, cr_idx, 1, glev, ds, nullptr, &sc, pc, pv);
            if (prior_out) {
                const double pr = p.have_prior ? prior_try_lds(p, *pc, pv, lane) : 0.0;
                if (lane == 0) prior_out[i] = pr;
            }
            return;
        }
        propose_point<1, false, LEAN>(p, phase, g, M, c, i, n, lane, xb, grow, rt, out + (size_t)i * out_stride, nullptr, false, cr_idx, 1, glev, ds, nullptr, &sc);
        if (!LEAN && prior_out) point_prior<1>(p, out + (size_t)i * out_stride, lane, prior_out + i);
    };
    const int rs = sc.slot0 + 1;                          // pt_slot(phase, i, 1) = rs + i npt
    const double* Zb = p.Z; const uint32_t ldb = 8u * (uint32_t)p.ld;
    if (!LEAN) {      // full code: two row buffers -- the rows of try i + 1 in flight during try i (the caller requested tries i0 and i0 + 1)
        for (int i = i0; i < i1; i += 2) {
            body(i, A);
            if (i + 1 >= i1) break;
            if (i + 2 < i1) request_pair<XF>(p, ds, rs + (i + 2) * sc.npt, gc, g, M, lane, A, Zb, ldb);
            body(i + 1, B);
            if (i + 3 < i1) request_pair<XF>(p, ds, rs + (i + 3) * sc.npt, gc, g, M, lane, B, Zb, ldb);
        }
        if (lane >= i0 && lane < i1) sl[lane] = 0.0;
        return;
    }
    for (int i = i0; i < i1; i += 3) {
        if (i + 2 < i1) request_pair<XF>(p, ds, rs + (i + 2) * sc.npt, gc, g, M, lane, C, Zb, ldb);
        body(i, A);
        if (i + 1 >= i1) break;
        if (i + 3 < i1) request_pair<XF>(p, ds, rs + (i + 3) * sc.npt, gc, g, M, lane, A, Zb, ldb);
        body(i + 1, B);
        if (i + 2 >= i1) break;
        if (i + 4 < i1) request_pair<XF>(p, ds, rs + (i + 4) * sc.npt, gc, g, M, lane, B, Zb, ldb);
        body(i + 2, C);
    }
    if (lane >= i0 && lane < i1) { sl[lane] = 0.0; if (LEAN && prior_out) prior_out[lane] = 0.0; }            // snooker_logp = 0 (flat priors: 0)
}

// PB: per-dimension priors and/or hard boundaries (SampledParam priors, parameters.py:37-47; Dream.py:733-791) -- the full
// propose_point with its prior evaluation; the flat, unbounded case keeps the lean code (2.4 % faster at the headline size).
// K1: multitry off (the reference's default, Dream.py:271-275 and :326-334) -- one proposal per generation, no reference set, the
// snooker move's current-point term; a template flag so that the multi-try kernels carry none of it (it cost them a register spill).
// M0: archive rows the first generation of this launch samples from; zappend: first row of the first history append made inside the launch
// (the rows go to zappend + global chain), by its generation index seg0 - 1, or -1: none -- with dz_config.history_lag the two differ (rows
// written earlier are not sampleable yet), and a launch may hold up to lag + 1 appends, thin generations apart (see the loop).
// publish: during the crossover burn-in (one generation per launch) the new states also go to the published positions
// (set_current_position_arr, Dream.py:364-366, :447-449: [N][ld], row = global chain); null otherwise.
// REDO (with PB, multi-try): a proposal set whose tries are ALL impossible is generated again, with the same decisions, from the key of the
// next redraw round, and evaluated again (Dream.py:281-289) -- a block-level loop around the proposal and likelihood steps of phase 0: only
// the chains that need it propose again, every wave takes part in the barriers and the likelihood units (the other chains' points have not
// changed: their sums come out the same), and the block leaves the loop when none of its chains needs another round (DZ_MAX_REDRAWS caps it).
// MG (round 6, adapt_lag >= 1; blocks of 16 chains with one wave each, the states in LDS): SEVERAL burn-in generations per launch
// (Publish::multi) -- the prologue applies the pending updates in order and leaves the probabilities of each of the launch's generations in an
// LDS table; every generation ends with the block's unit sums into that generation's ring slot (one more barrier per generation: the chains'
// states before the generation come from the Metropolis step's registers, the states after it are the LDS rows).
template <int NRT, bool TRI, bool XLDS, int CH, int WPC, bool PB, bool K1 = false, bool REDO = false, bool MG = false>
__global__ __launch_bounds__(64 * CH * WPC) void k_generations(const Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0, int64_t zappend, int seg0, Publish pub)
{
    double* const publish = pub.to;
    const Params& p = *pp;       // read through the scalar cache on demand: keeps the ~70 fields out of the SGPR file
    constexpr int NCH = 1;
    // WPC waves per chain (1 at 16 chains per block; 2 / 4 at 8 / 4 chains per block, i.e. when there are fewer than 16 chains
    // per CU): the tries of a phase are dealt to the chain's waves, all of them share the likelihood units, the chain's
    // first wave does the Metropolis step.  Every wave of a chain derives the chain's decisions itself (same draws, same
    // arithmetic); two more barriers per generation keep the base point and the new state consistent between them.
    constexpr int NT = 64 * CH * WPC;
    constexpr int LEANV = PB ? 0 : (K1 ? 2 : 1);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int d = p.d, k = K1 ? 1 : p.k, ld = p.ld;
    const bool pbl = PB && p.pb_lds != 0;
    static_assert(!MG || (CH == 16 && WPC == 1 && !K1 && XLDS && !REDO), "several burn-in generations per launch: 16 chains per block, one wave each, states in LDS");
    // (any instantiation; Publish::multi == 2) several burn-in generations per launch WITHOUT the block's own unit sums: the positions of every generation go to
    // the ring of published positions and k_adapt_partials_ring makes the sums behind the launch; the prologue and the table as with MG
    const bool RG = !MG && pub.multi == 2;
    const bool fuse_adapt = !MG && !RG && CH == 16 && WPC == 1 && !K1 && XLDS && pub.PR != nullptr;
    const MegaLayout L = mega_layout(d, k, NRT, p.ncr, p.ngamma, TRI, XLDS, CH, pbl, MG || fuse_adapt, false, (MG || RG) ? pub.lag + 1 : 0);
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
    PBConsts pcs;       // (PB) per-dimension prior and boundary constants, staged below
    {
        double* q = smem + L.off_pc;
        pcs.a = q; pcs.b = q + L.pcn; pcs.logb = q + 2 * L.pcn; pcs.lo = q + 3 * L.pcn; pcs.hi = q + 4 * L.pcn;
        pcs.kind = reinterpret_cast<const int*>(q + 5 * L.pcn);
        pcs.inside = smem + L.off_pr + p.ncr + p.ngamma;
    }
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;       // (scalar: everything derived from the wave number stays on the scalar unit)
    // chain inside the block; this wave's number among the chain's waves.  Four waves per chain (CH = 4): wave w runs on SIMD w % 4, and a
    // chain's waves sit on FOUR SIMDs with the numbers rotated per chain (round 5, as in k_generations_w4: when the block waits for one
    // chain's snooker set, that set has the CU's SIMDs to itself instead of a quarter of one)
#ifdef DZ_ONE_SIMD_PER_CHAIN
    const int cl = WPC == 1 ? wv : wv % CH;
    const int sub = WPC == 1 ? 0 : wv / CH;
#else
    const int cl = WPC == 1 ? wv : wv / WPC;
    const int sub = WPC == 1 ? 0 : ((wv % WPC) + cl) % WPC;
#endif
    const int cg = pub.c0 + blockIdx.x * CH + cl;
    const bool active = cg < pub.c1;
    const int c = min(cg, pub.c1 - 1);
    const uint32_t gc = (uint32_t)(p.off + c);
    const int tstride = CH * L.LDP;                                      // try i of chain cl: Pt + (CH i + cl) LDP
    // DEpairs > 1 (set_DEpair :571-583, several archive-row pairs per try): served by the PB instantiations, which carry the full
    // proposal code -- the general propose_set fetches its own rows, so the one-pair row prefetch is off
    const bool multipair = PB && p.depairs > 1;
    double* region = Pt + (size_t)cl * L.LDP;

    // ---- stage the matrix, mu, the selection probabilities, the gamma table, the chains' logp (and states)
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
    if (threadIdx.x < 4 * ((d + 3) / 4) + 4) mus[threadIdx.x] = threadIdx.x < d ? p.mu[threadIdx.x] : 0.0;
    if (MG || RG) {     // adapt_lag >= 1: the pending updates, in order; the probabilities of each of the launch's generations into the table
        if (wv == 0) adapt_pending_apply(p, pub.DOT, pub.CNTR, pub.nbp, pub.lag + 1, pub.pend0, pub.pend1, (long long)g0, ngen, pub.lag, pub.burnin, pub.sh, smem + L.off_tab,
                                         blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else if (pub.TOT) {      // the previous generation's adaptation totals are still to be applied: every block makes the update for itself (wave 0)
        if (wv == 0) adapt_apply_wave<1>(p, pub.TOT, pub.CNT, pub.sh, probs, blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else {
        if (threadIdx.x < p.ncr) probs[threadIdx.x] = pub.sh[threadIdx.x];
        if (threadIdx.x < p.ngamma) probs[p.ncr + threadIdx.x] = pub.sh[3 * p.ncr + threadIdx.x];
    }
    for (int i = threadIdx.x; i < p.ngamma * d; i += NT) gts[i] = p.gtab[(size_t)(i / d) * p.depairs * d + (i % d)];
    if (pbl && (int)threadIdx.x < L.pcn) {
        const int j = threadIdx.x;
        double* q = smem + L.off_pc;
        const bool hp = p.have_prior && j < d, hb = p.hard && j < d;
        q[j] = hp ? p.pa[j] : 0.0; q[L.pcn + j] = hp ? p.pc2[j] : 1.0; q[2 * L.pcn + j] = hp ? p.plogb[j] : 0.0;
        q[3 * L.pcn + j] = hb ? p.mins[j] : -__builtin_huge_val(); q[4 * L.pcn + j] = hb ? p.maxs[j] : __builtin_huge_val();
        reinterpret_cast<int*>(q + 5 * L.pcn)[j] = hp ? p.pkind[j] : 0;
    }
    if (lane == 0 && sub == 0) { st[4 * cl] = p.lprior[c]; st[4 * cl + 1] = p.llike[c]; st[4 * cl + 2] = 0.0; dec[8 * cl + 7] = chain_T(p, c); }      // (the chain's temperature: Dream.astep's T, core.py:133-136)
    if (XLDS && sub == 0) {
        for (int j = lane; j < L.LDP; j += 64) {
            const double v = j < d ? p.X[(size_t)c * ld + j] : 0.0;
            Xs[cl * L.LDP + j] = v;
            if (fuse_adapt) smem[L.off_Xo + cl * L.LDP + j] = v;
        }
    }
    __syncthreads();
    if (RG && blockIdx.x == 0 && pub.c0 == 0)      // the table for k_adapt_partials_ring (the bins of every generation come from ITS probabilities)
        for (int i = threadIdx.x; i < ngen * pub.nbp; i += NT) pub.PG[i] = smem[L.off_tab + i];
    if (PB && pbl && p.have_prior && p.prior_nonormal) {      // the log prior of a point inside every uniform support: the sum every such try would make
        if (wv == 0) {
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < 2; ++s) { const int j = 2 * lane + s; if (j < d) acc = acc + (pcs.kind[j] == 2 ? -pcs.logb[j] : 0.0); }
            acc = wave_bfly(acc);
            if (lane == 0) smem[L.off_pr + p.ncr + p.ngamma] = acc;
        }
        __syncthreads();
    }

    // A generation's wave-uniform draws: lane s holds slot s (both phases read them; four registers across the likelihood pass are
    // cheaper than a second Philox call).  They are made at the end of the PREVIOUS generation's second proposal phase, together
    // with the requests for the first archive rows the generation will need.
    // (the full-code instantiations read the raw draws: several pairs per try; with one try per generation -- multitry off -- or one or
    //  two tries per wave -- four waves per chain, every one of which would make the pass -- it costs more than it saves: 583 -> 557 M/s
    //  and 320 -> 301 M/s at 1024 chains)
    // (not in the full-code instantiations: measured there at -2.5 % -- the pass's registers -- with three row buffers, +-0 with two)
    constexpr bool XF = !PB && !K1 && WPC == 1;
    auto generation_draws = [&](uint32_t g_, uint32_t M) {
        DrawSrc q; q.have = true; q.mine = make_uint4(0, 0, 0, 0);
        if (lane < p.nslots) { const u32x4 w = slot_counter_draw(p, lane, gc, g_); q.mine = make_uint4(w.x, w.y, w.z, w.w); }
        if (XF && !multipair) {
            const uint32_t hx = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.mine.x), hy = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.mine.y);
            finish_draws(p, q, u53_below(hx, hy, p.snk_thr), M, lane);               // (slot 0 = lane 0: set_snooker's draw)
        }
        return q;
    };
    RowPair RA, RB, RC;
    RA.a = double2{0.0, 0.0}; RA.b = RA.a; RB = RA; RC = RA;
    auto prefetch_first = [&](const DrawSrc& q, int phase_, uint32_t g_, uint32_t M) {          // rows of this wave's first two tries of (g_, phase_)
        const int n_ = k - phase_;
        const int a0 = WPC == 1 ? 0 : (sub * n_) / WPC, a1 = WPC == 1 ? n_ : ((sub + 1) * n_) / WPC;
        if (a0 < a1) request_pair<XF>(p, q, pt_slot(p, phase_, a0, 1), gc, g_, M, lane, RA, p.Z, 8u * (uint32_t)p.ld);
        if (a0 + 1 < a1) request_pair<XF>(p, q, pt_slot(p, phase_, a0 + 1, 1), gc, g_, M, lane, RB, p.Z, 8u * (uint32_t)p.ld);
    };
    auto draws_say_snooker = [&](const DrawSrc& q, uint32_t g_) {                   // set_snooker :542-554 on the integer form of the draw
        const u32x4 w0 = uniform_draw(p, q, 0, gc, g_);
        return u53_below(w0.x, w0.y, p.snk_thr);
    };
    // The full-code instantiations live at the register limit (128 per wave at 16 waves per block: they spill): there the rows of a
    // phase's first tries are requested at the START of the phase, not a phase ahead -- 16 registers less across the likelihood pass and
    // the barriers -- and two row buffers rotate instead of three (propose_de_pf).  Measured together: uniform priors + hard boundaries
    // 508 -> 612 M proposals/s, normal priors 522 -> 572 (spilled registers are reloaded on the generation's critical path).
    constexpr bool PF_AHEAD = !PB;
    DrawSrc dsn = generation_draws(g0, M0);
    if (PF_AHEAD && !multipair && !draws_say_snooker(dsn, g0)) prefetch_first(dsn, 0, g0, M0);
    // History appends inside the launch (history_lag >= 1 on one GPU: the rows an append writes are not sampled before `lag` more appends
    // have been made, so a launch may run on past it): generation index next_app makes the next one, into rows zappend + (M - M0) + global chain; the
    // generations behind it sample p.N more rows (M: this generation's count, Mn: the next one's).
    uint32_t M = M0;
    int next_app = zappend >= 0 ? seg0 - 1 : -1;

    for (int gi = 0; gi < ngen; ++gi) {
        const uint32_t g = g0 + (uint32_t)gi;
        const bool last = gi == ngen - 1;
        const bool app = gi == next_app;
        const uint32_t Mn = app ? M + (uint32_t)p.N : M;
        const double* const pr_g = (MG || RG) ? smem + L.off_tab + (size_t)gi * pub.nbp : probs;      // the probabilities this generation decides with
        double* const publish_g = RG ? publish + (size_t)(g % (uint32_t)(pub.lag + 2)) * pub.pos_stride : publish;      // (the ring of published positions: lag + 2 slots)
        DZ_MSTAMP(0);
        const DrawSrc ds = dsn;
        constexpr int nph = K1 ? 1 : 2;                                              // multitry off: no reference set
        for (int phase = 0; phase < nph; ++phase) {
            // ---- phase 0: k proposals around the chain's state (generate_proposal_points :258-264) into the chain's rows
            //      of tiles 0..k-1; phase 1: the selected proposal moves to tile 0 and k-1 reference points around it
            //      (:295-299) take tiles 1..k-1.  One copy of the code serves both (instruction cache).
            if (phase) DZ_MSTAMP(10);
            StepFlags f;
            double base[NCH][2];
            if (phase == 0) {
                Ctrl u;
                const u32x4 w0 = uniform_draw(p, ds, 0, gc, g), w1 = uniform_draw(p, ds, 1, gc, g), w2 = uniform_draw(p, ds, 2, gc, g);
                u.u_snk = u53(w0.x, w0.y); u.u_cr = u53(w0.z, w0.w); u.u_de = u53(w1.x, w1.y); u.u_glev = u53(w1.z, w1.w);
                u.u_sel = u53(w2.x, w2.y); u.u_acc = u53(w2.z, w2.w);
                f = step_flags_from(p, u, pr_g, pr_g + p.ncr);                       // Dream.py:246-256
                if (lane == 0 && sub == 0) {
                    double* dc = dec + 8 * cl;
                    dc[0] = u.u_sel; dc[1] = u.u_acc; dc[2] = f.snk ? 1.0 : 0.0; dc[3] = (double)f.cr_idx; dc[4] = (double)f.glev;
                    if (PB) dc[5] = (double)f.delta;
                }
                if (XLDS) {
                    const double* xr = Xs + cl * L.LDP;
                    base[0][0] = xr[2 * lane < d ? 2 * lane : 0]; base[0][1] = xr[2 * lane + 1 < d ? 2 * lane + 1 : 0];
                    if (2 * lane >= d) base[0][0] = 0.0;
                    if (2 * lane + 1 >= d) base[0][1] = 0.0;
                } else load_row<NCH>(p.X + (size_t)c * ld, ld, lane, base);
            } else {
                const double* dc = dec + 8 * cl;
                f.snk = dc[2] != 0.0; f.cr_idx = (int)dc[3]; f.delta = PB ? (int)dc[5] : 1; f.glev = (int)dc[4];
                // likelihoods of the chain's k points, mt_choose_proposal_pt (:291)
                const double u_sel = dc[0];
                double lp = -__builtin_huge_val();
                if (lane < k) {
                    const int pt = lane * CH + cl;
                    double qt[NRT];
#pragma unroll
                    for (int t = 0; t < NRT; ++t) qt[t] = qb[pt * NRT + t];          // (all reads in flight, then the ordered sum)
                    double Q = 0.0;
#pragma unroll
                    for (int t = 0; t < NRT; ++t) Q = Q + qt[t];
                    const double lk = nan_to_ninf(p.logF - 0.5 * Q);
                    if (sub == 0) sL[cl * k + lane] = lk;
                    lp = sP[cl * k + lane] + dec[8 * cl + 7] * lk;
                }
                bool fin;
                DZ_MSTAMP(11);
                const int sel = mt_select_vals(k, lp, u_sel, lane, &fin);
                DZ_MSTAMP(12);
                if (lane == 0 && sub == 0) st[4 * cl + 2] = (double)(sel | (fin ? 256 : 0));
                const double* row = region + (size_t)sel * tstride;
                base[0][0] = 2 * lane < d ? row[2 * lane] : 0.0; base[0][1] = 2 * lane + 1 < d ? row[2 * lane + 1] : 0.0;
                if (WPC > 1) __syncthreads();                                        // every wave of the chain holds the base point before any try row is rewritten
                if (sub == 0) {
                    if (2 * lane < d) region[2 * lane] = base[0][0];                 // the selected proposal now sits in tile 0
                    if (2 * lane + 1 < d) region[2 * lane + 1] = base[0][1];
                }
            }
            const bool snk_s = __builtin_amdgcn_readfirstlane((int)f.snk) != 0;      // (the decision is wave-uniform: branch on the scalar unit)
            const double* grow = multipair ? gamma_row(p, __builtin_amdgcn_readfirstlane(f.glev), __builtin_amdgcn_readfirstlane(f.delta))
                                           : gts + (size_t)(__builtin_amdgcn_readfirstlane(f.glev) - 1) * d;
            if (phase) DZ_MSTAMP(13);
            const int n = k - phase;
            const int i0 = WPC == 1 ? 0 : (sub * n) / WPC, i1 = WPC == 1 ? n : ((sub + 1) * n) / WPC;     // this wave's tries
            double* slp = phase ? rS + cl * (k - 1) : sS + cl * k;
            double* prp = phase ? rP + cl * (k - 1) : sP + cl * k;
            DrawSrc dcur = ds;                                                       // (REDO: a redraw round's draws)
            int round = 0; bool mine = true, redrew = false;
            // tries [a, b) of this wave's share; A_ / B_ hold the archive rows of tries a and a + 1 of a one-pair DE set
            auto propose_range = [&](int a, int b, RowPair& A_, RowPair& B_, RowPair& C_) {
                if (!snk_s && multipair) {
                    if (a < b)
                        propose_set<NCH, false, true, 0>(p, phase, g, M, c, gc, a, b, n, lane, base, grow, false, f.cr_idx, f.delta, f.glev, dcur,
                                                          region + (size_t)phase * tstride, tstride, slp, nullptr, prp, pbl ? &pcs : nullptr);
                } else if (!snk_s) {
                    propose_de_pf<LEANV, XF>(p, phase, g, M, c, gc, a, b, n, lane, base, grow, f.cr_idx, f.glev, dcur,
                                       region + (size_t)phase * tstride, tstride, slp, prp, A_, B_, C_, pbl ? &pcs : nullptr);
                } else if (a < b) {
                    // a snooker set is the longest path to the block's barrier (three rows and three reductions per try, one chain in
                    // ten): its wave gets issue priority over the three DE waves it shares a SIMD with
                    __builtin_amdgcn_s_setprio(3);
                    propose_set<NCH, false, false, LEANV>(p, phase, g, M, c, gc, a, b, n, lane, base, grow, true, f.cr_idx, 1, f.glev, dcur,
                                                         region + (size_t)phase * tstride, tstride, slp, K1 ? st + 4 * cl + 3 : nullptr, prp, pbl ? &pcs : nullptr);   // (k = 1: log |x - z|^(d-1) of the current point, :328-329)
                    __builtin_amdgcn_s_setprio(0);
                }
            };
            for (;;) {
                if (mine) {
                    if (((REDO && round > 0) || !PF_AHEAD) && !snk_s && !multipair) prefetch_first(dcur, phase, g, M);      // (round 0's first rows were requested a phase ahead)
                    propose_range(i0, i1, RA, RB, RC);
                    if (PF_AHEAD && phase == 0 && round == 0 && !snk_s && !multipair) prefetch_first(ds, 1, g, M);  // the reference set's first rows, ahead of the likelihood pass
                }
                if (round == 0 && phase == nph - 1 && !last) {                       // the next generation's draws and first rows
                    dsn = generation_draws(g + 1u, Mn);
                    if (PF_AHEAD && !multipair && !draws_say_snooker(dsn, g + 1u)) prefetch_first(dsn, 0, g + 1u, Mn);
                }
                DZ_MSTAMP(1 + 4 * phase);
                __syncthreads();                                                     // points visible
                DZ_MSTAMP(2 + 4 * phase);
                // mt_evaluate_logps :278, :302 (x - 0.0 == x bit for bit, so a zero mean skips the subtraction and its LDS read)
                {
                    const int row0 = phase ? CH : 0, ntl = ((k - phase) * CH + 15) / 16;
                    // (the pair-of-tiles units of k_generations_d2 in here, the matrix still in LDS: 714 against 725 M proposals/s at the headline size -- coarser
                    //  units balance worse over the 16 waves and the rolled batch loop costs address arithmetic; not kept)
                    if (p.mu_zero) mfma_units<NRT, TRI, true>(p, Ms, Pt, mus, qb, row0, ntl, wv, CH * WPC, lane, L.LDM, L.LDP);
                    else mfma_units<NRT, TRI, false>(p, Ms, Pt, mus, qb, row0, ntl, wv, CH * WPC, lane, L.LDM, L.LDP);
                }
                DZ_MSTAMP(3 + 4 * phase);
                __syncthreads();                                                     // q visible
                if (!(REDO && phase == 0)) break;
                // `while np.all(np.isfinite(np.array(log_ps))==False)` (:281-282): log_ps = log_priors + T * log_likes of the chain's k tries
                double lpv = -__builtin_huge_val();
                if (lane < k) {
                    const int pt = lane * CH + cl;
                    double Q = 0.0;
#pragma unroll
                    for (int t = 0; t < NRT; ++t) Q = Q + qb[pt * NRT + t];
                    lpv = sP[cl * k + lane] + dec[8 * cl + 7] * nan_to_ninf(p.logF - 0.5 * Q);
                }
                const bool need = __any(lane < k && is_finite(lpv)) == 0 && round < DZ_MAX_REDRAWS_DEV;
                if (!__syncthreads_or(need ? 1 : 0)) break;
                ++round; mine = need; redrew = redrew || need;
                if (threadIdx.x == 0 && p.redraw_count) atomicAdd(p.redraw_count, 1ull);
                if (mine) {   // round r of the set: point / dimension / boundary streams from the key seed + r * step (:284-289 with the same decisions)
                    const unsigned long long key = (((unsigned long long)p.k1 << 32) | (unsigned long long)p.k0) + (unsigned long long)round * DZ_REDRAW_KEY_STEP_DEV;
                    dcur.rekey = true; dcur.k0 = (uint32_t)key; dcur.k1 = (uint32_t)(key >> 32);
                    dcur.mine = make_uint4(0, 0, 0, 0);
                    if (lane < p.nslots) { const u32x4 w = slot_counter_draw_key(p, lane, gc, g, dcur.k0, dcur.k1); dcur.mine = make_uint4(w.x, w.y, w.z, w.w); }
                    dcur.xf = false;
                    if (XF && !multipair) finish_draws(p, dcur, snk_s, M, lane);
                }
            }
            if (PF_AHEAD && REDO && phase == 0 && redrew && !snk_s && !multipair) prefetch_first(ds, 1, g, M);      // (the redraw rounds used the row buffers)
            DZ_MSTAMP(4 + 4 * phase);
        }
        // ---- Metropolis step (:305-347), trace (core.py:114-116), record_history (:919-938)
        if (sub == 0) {
            const double* dc = dec + 8 * cl;
            const double u_acc = dc[1];
            const bool snk = dc[2] != 0.0;
            const int cr_idx = (int)dc[3];
            const double lpri = st[4 * cl], llik = st[4 * cl + 1], Tch = dec[8 * cl + 7];
            const int sf = (int)st[4 * cl + 2]; const int sel = sf & 255; const bool fin = (sf & 256) != 0;
            double val = -__builtin_huge_val();
            if (K1) {                                                           // single try: the proposal's density straight from the q sums (:271-275)
                double qt[NRT];
#pragma unroll
                for (int t = 0; t < NRT; ++t) qt[t] = qb[cl * NRT + t];
                double Q = 0.0;
#pragma unroll
                for (int t = 0; t < NRT; ++t) Q = Q + qt[t];
                val = nan_to_ninf(p.logF - 0.5 * Q);                                 // (every lane: the same sums)
                sL[cl] = val;
            } else if (lane < k) {
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
            DZ_MSTAMP(14);
            double lu, ratio;
            if (K1) {
                const double q_logp = Tch * val + sP[cl], last_logp = Tch * llik + lpri;                 // :274, :243
                if (snk) ratio = nan_to_num((q_logp + sS[cl]) - (last_logp + st[4 * cl + 3]));           // :326-332
                else ratio = nan_to_num(q_logp) - nan_to_num(last_logp);                                 // :334
                lu = dlog(u_acc);
            } else {
                ratio = mt_log_ratio(k, val, u_acc, lane, &lu);                      // log(u) of :993 rides in the ratio's logarithm pass
                if (!fin) ratio = -__builtin_huge_val();                             // DESIGN.md deviation D1
            }
            const bool accept = is_finite(ratio) && (lu < ratio);                    // :993
            DZ_MSTAMP(15);
            const int jj = 2 * lane;
            double2 xo = {0.0, 0.0};
            if (XLDS) { const double* xr = Xs + cl * L.LDP; if (jj < d) xo.x = xr[jj]; if (jj + 1 < d) xo.y = xr[jj + 1]; }
            else if (jj < ld) xo = *reinterpret_cast<const double2*>(p.X + (size_t)c * ld + jj);
            double2 xn = xo;
            if (accept) { xn.x = jj < d ? region[jj] : 0.0; xn.y = jj + 1 < d ? region[jj + 1] : 0.0; }   // the selected proposal
            const bool moved = __any((xn.x != xo.x) || (xn.y != xo.y));              // core.py:120
            const double npri = accept ? sP[cl * k + sel] : lpri, nlik = accept ? sL[cl * k + sel] : llik;   // :345-347
            if (XLDS && accept) { double* xr = Xs + cl * L.LDP; if (jj < d) xr[jj] = xn.x; if (jj + 1 < d) xr[jj + 1] = xn.y; }
            if (MG) { double* xq = smem + L.off_Xo + cl * L.LDP; if (jj < d) xq[jj] = xo.x; if (jj + 1 < d) xq[jj + 1] = xo.y; }      // the state before this generation (the jump's base)
            if (active) {
                if (jj < ld) {
                    if (XLDS ? last : accept) gstore2(p.X + (size_t)c * ld + jj, xn);
                    if (trace_slot0 >= 0) gstore2(p.tX + ((size_t)c * p.tcap + (size_t)(trace_slot0 + gi)) * ld + jj, xn);
                    if (app) gstore2(p.Z + ((size_t)zappend + (M - M0) + gc) * ld + jj, xn);                         // record_history :933-936
                    if (publish && (!MG || last)) gstore2(publish_g + (size_t)gc * ld + jj, xn);         // set_current_position_arr :447-449
                    if (MG && gc == 0u) gstore2(pub.x0ring + (size_t)(g % (uint32_t)(2 * (pub.lag + 1))) * ld + jj, xn);      // global chain 0 after generation g: a later generation's shift
                }
                if (lane == 0) {
                    if (trace_slot0 >= 0) {
                        const size_t o = (size_t)(trace_slot0 + gi) * p.nl + c;
                        p.tlogp[o] = Tch * nlik + npri;                              // core.py:115; with a temperature ladder core.py:178 (1.0 * x == x)
                        p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = cr_idx; p.tsnk[o] = snk ? 1 : 0;
                    }
                    if (last) { p.lprior[c] = npri; p.llike[c] = nlik; }
                }
            }
            if (lane == 0) { st[4 * cl] = npri; st[4 * cl + 1] = nlik; }
        }
        DZ_MSTAMP(9);
        if (MG) {   // the block's unit sums of THIS generation (contract v3) into its ring slot: the states before (off_Xo) and after it (the chains' LDS rows)
            int bc, bg;
            adapt_bins(p, g, (int)gc, lane, bc, bg, pr_g, pr_g + p.ncr);
            if (lane == 0) { st[4 * cl + 3] = (double)bc; dec[8 * cl + 6] = (double)bg; }
            __syncthreads();
            const int unit = blockIdx.x, R1 = pub.lag + 1, slot = (int)(g % (uint32_t)R1);
            const long long hs = (long long)g - 1 - pub.lag;
            const double* shift = hs < 0 ? pub.x0start : pub.x0ring + (size_t)(hs % (2 * R1)) * ld;
            adapt_unit_sums(p, Xs, L.LDP, smem + L.off_Xo, L.LDP, min(16, p.nl - 16 * unit), [&](bool isg, int c_) { return (int)(isg ? dec[8 * c_ + 6] : st[4 * c_ + 3]); }, shift,
                            pub.PR + (size_t)slot * pub.pr_stride + (size_t)unit * adapt_nq(p) * ld, pub.PC + (size_t)slot * pub.pc_stride + (size_t)unit * (p.ncr + p.ngamma),
                            (int)threadIdx.x, NT);
        }
        // one wave per chain: no barrier here -- the next generation's first phase only touches each wave's own chain's rows
        // and scalars, and the shared q buffer is not written again before the next barrier
        if (WPC > 1) __syncthreads();                                                // the chain's other waves read the new state
        if (app) next_app += p.thin;
        M = Mn;
    }
    // Crossover burn-in (one generation per launch), a block of 16 chains = one unit of the adaptation's column sums (contract v3): the
    // chains' new states sit in LDS, so do the ones they started the launch with (off_Xo); every wave makes its chain's
    // bins, then the block adds its unit's sums (adapt_unit_sums).  Kept outside the generation loop: nothing of it lives in registers there.
    if (fuse_adapt) {
        int bc, bg;
        adapt_bins(p, g0, (int)gc, lane, bc, bg, probs, probs + p.ncr);
        if (lane == 0) { st[4 * cl + 3] = (double)bc; dec[8 * cl + 6] = (double)bg; }      // (st[.. + 3] is the multitry-off kernels' word; dec[.. + 7] holds the temperature)
        __syncthreads();
        const int unit = blockIdx.x;
        adapt_unit_sums(p, Xs, L.LDP, smem + L.off_Xo, L.LDP, min(16, p.nl - 16 * unit), [&](bool isg, int c_) { return (int)(isg ? dec[8 * c_ + 6] : st[4 * c_ + 3]); }, pub.shift,
                        pub.PR + (size_t)unit * adapt_nq(p) * ld, pub.PC + (size_t)unit * (p.ncr + p.ngamma), (int)threadIdx.x, NT);
    }
}

#ifndef DZ_TEMPLATES_ONLY
// ---------------------------------------------------------------------------------------------------------------------
// The same persistent scheme for a likelihood a single wave evaluates on its own (Gaussian mixture): nothing is shared
// between the chains of a block, so there are no tiles and no barriers at all -- every wave carries its chain through the
// generations of the launch independently (its k points in an LDS region it alone touches, its state in registers).
// Eligibility: multitry 1 or >= 3, ld <= 128, up to 32 components; priors / boundaries / several pairs run the <true> instantiation.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int MIXW = 4;       // waves (= chains) per block

__host__ __device__ inline int mega_mix_wave_doubles(int d, int k, int J) { const int n = k * (4 * ((d + 3) / 4) + 1) + 6 * k + k * J + 8; return n + (n & 1); }

// (blocks of MIXW waves; blocks of 16 -- one adaptation unit -- inside the crossover burn-in, where the block adds its unit's sums)
// PB (round 4): per-dimension priors, hard boundaries, several DE pairs -- the full proposal code and the prior evaluation (constants from
// global memory: this kernel has no block-wide staging); the flat, unbounded, one-pair case keeps the lean instantiation.
// MG (round 6, adapt_lag >= 1): SEVERAL burn-in generations per launch (Publish::multi; blocks of 16).  The prologue applies the pending
// updates in order and leaves the probabilities of each of the launch's generations in an LDS table; every generation ends with the block's
// unit sums into that generation's ring slot: the chains' states before and after it and their bins go through a double-buffered LDS stash
// (one block barrier per generation -- the only one in this kernel: a wave can be at most one generation ahead of the slowest, and the
// stash it then writes is the other one).  LDS behind the waves' regions: table [lag + 1][nbp] | stash [2]{before, after}[16][LDP] | bins [2][32] (int).
// LK: the likelihood of the wave's n points (mt_evaluate_logps :278, :302) -- rows [n][LDP] in the wave's LDS region -> out[0 .. n-1]; lh: the wave's
// scratch ([k][J] doubles).  MixLike: the built-in mixture.  A user's device function takes its place in a code object built at run time
// (dz_user_generations.hip.in; pydream_amd.likelihoods.DeviceFunctionLogLike): the same kernel around ANY density a wave can evaluate.
struct MixLike {
    template <int NCH>
    DZ_DEV static void eval(const Params& p, const double* rows, int LDP, int n, int lane, double* lh, double* out)
    {
        const int d = p.d;
        // The squared distances to the J means need the whole wave (one butterfly each); the log-sum-exp of a point is scalar work, so lane i
        // does it for point i and the n points cost one pass of exp / log instead of n (same operations per point as k_logp_mix).
        // (points in groups of GP: GP independent reductions in flight per component, and each mean is read once per group)
        constexpr int GP = 4;
        for (int i0 = 0; i0 < n; i0 += GP) {
            double x0[GP][NCH], x1[GP][NCH];      // (a lane's dimensions 128 it + 2 lane, + 1: in ascending order, as k_logp_mix adds them)
#pragma unroll
            for (int u = 0; u < GP; ++u) {
                const double* row = rows + (size_t)min(i0 + u, n - 1) * LDP;
#pragma unroll
                for (int it = 0; it < NCH; ++it) { const int jj = 128 * it + 2 * lane; x0[u][it] = jj < d ? row[jj] : 0.0; x1[u][it] = jj + 1 < d ? row[jj + 1] : 0.0; }
            }
            for (int j = 0; j < p.J; ++j) {
                const double* mj = p.mu + (size_t)j * p.ld;
                double m0[NCH], m1[NCH];
#pragma unroll
                for (int it = 0; it < NCH; ++it) { const int jj = 128 * it + 2 * lane; m0[it] = jj < d ? mj[jj] : 0.0; m1[it] = jj + 1 < d ? mj[jj + 1] : 0.0; }
                double S[GP];
#pragma unroll
                for (int u = 0; u < GP; ++u) {
                    double acc = 0.0;
#pragma unroll
                    for (int it = 0; it < NCH; ++it) {
                        const int jj = 128 * it + 2 * lane;
                        if (jj < d) { const double t = x0[u][it] - m0[it]; acc = fma(t, t, acc); }
                        if (jj + 1 < d) { const double t = x1[u][it] - m1[it]; acc = fma(t, t, acc); }
                    }
                    S[u] = acc;
                }
                static_assert(GP == 4, "wave_bfly4 sums four values");
                const double R = wave_bfly4(S[0], S[1], S[2], S[3]);          // (rows hold the totals of S[0], S[2], S[1], S[3]: the bits of four wave_bfly)
                {
                    const int row = lane >> 4, u = (row == 1) ? 2 : (row == 2 ? 1 : row);
                    if ((lane & 15) == 0 && i0 + u < n) lh[(i0 + u) * p.J + j] = -0.5 * R + p.mixF[j];
                }
            }
        }
        {
            const int i = lane < n ? lane : 0;
            double mx = -__builtin_huge_val();
            for (int j = 0; j < p.J; ++j) { const double v = lh[i * p.J + j]; if (v > mx) mx = v; }
            double dens = 0.0;
            for (int j = 0; j < p.J; ++j) dens = dens + dexp(lh[i * p.J + j] - mx);
            const double lk = nan_to_ninf(dlog(dens) + mx);
            if (lane < n) out[lane] = lk;
        }
    }
};

template <bool PB, bool MG, class LK, int NCH = 1>
DZ_DEV void generations_wave_body(const Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0, int64_t zappend, int seg0, const Publish& pub)
{
    double* const publish = pub.to;
    const Params& p = *pp;
    static_assert(NCH == 1 || !MG, "the blocks' own unit sums: d <= 128");      // (NCH = 2, 128 < d <= 256: a lane owns four dimensions, 128 it + 2 lane, + 1)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int d = p.d, k = p.k, ld = p.ld;
    const int LDP = 4 * ((d + 3) / 4) + 1;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* region = smem + (size_t)wv * mega_mix_wave_doubles(d, k, p.J);     // [k][LDP] the chain's points
    double* sP = region + (size_t)k * LDP; double* sS = sP + k; double* sL = sS + k; double* rS = sL + k; double* rL = rS + k;
    double* lh = rL + k;                                                    // [k][J] mixture component terms; then [8] decisions
    double* dec = lh + (size_t)k * p.J;
    double* rP = dec + 8;                                                   // [k] (PB) log priors of the reference points
    const int nwv = blockDim.x >> 6;
    // the crossover / gamma-level probabilities the launch decides with: behind the waves' regions; made by wave 0 when the previous
    // generation's adaptation totals are still to be applied (Publish::TOT)
    double* probs = smem + (size_t)nwv * mega_mix_wave_doubles(d, k, p.J);
    const bool RG = !MG && pub.multi == 2;      // (k_generations: the positions of every generation into the ring, the sums behind the launch)
    if (MG || RG) {
        if (wv == 0) adapt_pending_apply(p, pub.DOT, pub.CNTR, pub.nbp, pub.lag + 1, pub.pend0, pub.pend1, (long long)g0, ngen, pub.lag, pub.burnin, pub.sh, probs,
                                         blockIdx.x == 0 ? pub.sh_out : nullptr, lane);
    } else if (pub.TOT) { if (wv == 0) adapt_apply_wave<NCH>(p, pub.TOT, pub.CNT, pub.sh, probs, blockIdx.x == 0 ? pub.sh_out : nullptr, lane); }
    else {
        if ((int)threadIdx.x < p.ncr) probs[threadIdx.x] = pub.sh[threadIdx.x];
        if ((int)threadIdx.x < p.ngamma) probs[p.ncr + threadIdx.x] = pub.sh[3 * p.ncr + threadIdx.x];
    }
    __syncthreads();
    if (RG && blockIdx.x == 0 && pub.c0 == 0)
        for (int i = threadIdx.x; i < ngen * pub.nbp; i += (int)blockDim.x) pub.PG[i] = probs[i];
    const int cg = pub.c0 + blockIdx.x * nwv + wv;
    const bool active = cg < pub.c1;
    const int c = min(cg, pub.c1 - 1);
    const uint32_t gc = (uint32_t)(p.off + c);
    double xs[NCH][2];                                                      // the chain's state lives in registers
    load_row<NCH>(p.X + (size_t)c * ld, ld, lane, xs);
    double* xo_area = probs + (MG ? (pub.lag + 1) * pub.nbp : ((p.ncr + p.ngamma + 1) & ~1));      // (crossover burn-in, blocks of 16: the states the launch started with, [16][LDP]; MG: the stash)
    if (NCH == 1 && !MG && pub.PR) { if (2 * lane < d) xo_area[wv * LDP + 2 * lane] = xs[0][0]; if (2 * lane + 1 < d) xo_area[wv * LDP + 2 * lane + 1] = xs[0][1]; }
    double lpri = p.lprior[c], llik = p.llike[c];
    if (lane == 0) dec[6] = chain_T(p, c);                                  // the chain's temperature (Dream.astep's T)
    // (PB) the prior / boundary constants straight from global memory through the PBConsts interface of the block-staged kernels: the prior
    // is evaluated on the values in registers, and where the supports contain the boundaries' box (Params::prior_const) it is one constant --
    // made here by the wave itself, kept in the word of rP the reference set never uses
    PBConsts pcs;
    pcs.a = p.pa; pcs.b = p.pc2; pcs.logb = p.plogb; pcs.lo = p.mins; pcs.hi = p.maxs; pcs.kind = p.pkind; pcs.inside = rP + (k - 1);
    if (PB && p.have_prior && p.prior_nonormal) {
        double acc = 0.0;
#pragma unroll
        for (int it = 0; it < NCH; ++it)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) { const int j = 128 * it + 2 * lane + s2; if (j < d) acc = acc + (p.pkind[j] == 2 ? -p.plogb[j] : 0.0); }
        acc = wave_bfly(acc);
        if (lane == 0) rP[k - 1] = acc;
    }
    uint32_t M = M0;                                                        // (appends inside the launch: k_generations)
    int next_app = zappend >= 0 ? seg0 - 1 : -1;
    for (int gi = 0; gi < ngen; ++gi) {
        const uint32_t g = g0 + (uint32_t)gi;
        const bool last = gi == ngen - 1;
        const bool app = gi == next_app;
        int sel = 0; bool fin = true;
        const double* const pr_g = (MG || RG) ? probs + (size_t)gi * pub.nbp : probs;      // the probabilities this generation decides with
        double* const publish_g = RG ? publish + (size_t)(g % (uint32_t)(pub.lag + 2)) * pub.pos_stride : publish;
        const double xb0 = xs[0][0], xb1 = xs[0][1];                          // (MG) the state before the generation
        DrawSrc ds; ds.have = true; ds.mine = make_uint4(0, 0, 0, 0);      // the generation's wave-uniform draws, both phases read them
        if (lane < p.nslots) { const u32x4 w = slot_counter_draw(p, lane, gc, g); ds.mine = make_uint4(w.x, w.y, w.z, w.w); }
        const int nph = k == 1 ? 1 : 2;                                    // multitry off: one proposal, no reference set
        for (int phase = 0; phase < nph; ++phase) {
            StepFlags f;
            double base[NCH][2];
            if (phase == 0) {
                Ctrl u;
                const u32x4 w0 = uniform_draw(p, ds, 0, gc, g), w1 = uniform_draw(p, ds, 1, gc, g), w2 = uniform_draw(p, ds, 2, gc, g);
                u.u_snk = u53(w0.x, w0.y); u.u_cr = u53(w0.z, w0.w); u.u_de = u53(w1.x, w1.y); u.u_glev = u53(w1.z, w1.w);
                u.u_sel = u53(w2.x, w2.y); u.u_acc = u53(w2.z, w2.w);
                f = step_flags_from(p, u, pr_g, pr_g + p.ncr);                      // Dream.py:246-256
                if (lane == 0) { dec[0] = u.u_sel; dec[1] = u.u_acc; dec[2] = f.snk ? 1.0 : 0.0; dec[3] = (double)f.cr_idx; dec[4] = (double)f.glev; if (PB) dec[7] = (double)f.delta; }
#pragma unroll
                for (int it = 0; it < NCH; ++it) { base[it][0] = xs[it][0]; base[it][1] = xs[it][1]; }
            } else {
                f.snk = dec[2] != 0.0; f.cr_idx = (int)dec[3]; f.delta = PB ? (int)dec[7] : 1; f.glev = (int)dec[4];
                double lp = -__builtin_huge_val();
                if (lane < k) lp = sP[lane] + dec[6] * sL[lane];                    // :279, mt_choose_proposal_pt :291
                sel = mt_select_vals(k, lp, dec[0], lane, &fin);
                const double* row = region + (size_t)sel * LDP;
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    const int jj = 128 * it + 2 * lane;
                    base[it][0] = jj < d ? row[jj] : 0.0; base[it][1] = jj + 1 < d ? row[jj + 1] : 0.0;
                    if (jj < d) region[jj] = base[it][0];                           // the selected proposal now sits in row 0
                    if (jj + 1 < d) region[jj + 1] = base[it][1];
                }
            }
            const double* grow = gamma_row(p, f.glev, PB ? f.delta : 1);
            const int n = k - phase;
            double* rows = region + (size_t)phase * LDP;
            if (PB) propose_set<NCH, false, true, 0>(p, phase, g, M, c, gc, 0, n, n, lane, base, grow, f.snk, f.cr_idx, f.delta, f.glev, ds,
                                                     rows, LDP, (phase ? rS : sS), (k == 1 ? dec + 5 : nullptr), (phase ? rP : sP), &pcs);
            else propose_set<NCH, false, false, 2>(p, phase, g, M, c, gc, 0, n, n, lane, base, grow, f.snk, f.cr_idx, 1, f.glev, ds,
                                                   rows, LDP, (phase ? rS : sS), (k == 1 ? dec + 5 : nullptr), (phase ? lh : sP));   // (flat priors: in phase 1 the prior slot is scratch)
            LK::template eval<NCH>(p, rows, LDP, n, lane, lh, phase ? rL : sL);                   // mt_evaluate_logps :278, :302 -- by this wave, for its own points
        }
        // ---- Metropolis step (:305-347), trace (core.py:114-116), record_history (:919-938)
        {
            const double u_acc = dec[1];
            const bool snk = dec[2] != 0.0;
            const int cr_idx = (int)dec[3];
            double val = -__builtin_huge_val();
            if (lane < k) {
                val = sP[lane] + dec[6] * sL[lane];                                 // :279
                if (snk) val = val + sS[lane];                                      // :307
            } else if (lane >= 16 && lane < 16 + k) {
                const int i = lane - 16;
                val = i < k - 1 ? dec[6] * rL[i] + (PB ? rP[i] : 0.0) : dec[6] * llik + lpri;      // :303, :877-879
                if (snk) { const double sr = i < k - 1 ? rS[i] : 0.0; val = (val + sr) + sS[i]; }   // :312-313
            }
            double lu, ratio;
            if (k == 1) {
                const double q_logp = dec[6] * sL[0] + sP[0], last_logp = dec[6] * llik + lpri;          // :274, :243
                if (snk) ratio = nan_to_num((q_logp + sS[0]) - (last_logp + dec[5]));                    // :326-332
                else ratio = nan_to_num(q_logp) - nan_to_num(last_logp);                                 // :334
                lu = dlog(u_acc);
            } else {
                ratio = mt_log_ratio(k, val, u_acc, lane, &lu);
                if (!fin) ratio = -__builtin_huge_val();                            // DESIGN.md deviation D1
            }
            const bool accept = is_finite(ratio) && (lu < ratio);                   // :993
            double2 xn[NCH];
            bool diff = false;
#pragma unroll
            for (int it = 0; it < NCH; ++it) {
                const int jj = 128 * it + 2 * lane;
                const double2 xo = {xs[it][0], xs[it][1]};
                xn[it] = xo;
                if (accept) { xn[it].x = jj < d ? region[jj] : 0.0; xn[it].y = jj + 1 < d ? region[jj + 1] : 0.0; }
                diff = diff || (xn[it].x != xo.x) || (xn[it].y != xo.y);
                xs[it][0] = xn[it].x; xs[it][1] = xn[it].y;
            }
            const bool moved = __any(diff);                                          // core.py:120
            const double npri = accept ? sP[sel] : lpri, nlik = accept ? sL[sel] : llik;    // :345-347
            if (active) {
#pragma unroll
                for (int it = 0; it < NCH; ++it) {
                    const int jj = 128 * it + 2 * lane;
                    if (jj < ld) {
                        if (last) *reinterpret_cast<double2*>(p.X + (size_t)c * ld + jj) = xn[it];
                        if (trace_slot0 >= 0) gstore2(p.tX + ((size_t)c * p.tcap + (size_t)(trace_slot0 + gi)) * ld + jj, xn[it]);
                        if (app) gstore2(p.Z + ((size_t)zappend + (M - M0) + gc) * ld + jj, xn[it]);                         // record_history :933-936
                        if (publish && (!MG || last)) gstore2(publish_g + (size_t)gc * ld + jj, xn[it]);         // set_current_position_arr :447-449
                        if (MG && gc == 0u) gstore2(pub.x0ring + (size_t)(g % (uint32_t)(2 * (pub.lag + 1))) * ld + jj, xn[it]);      // global chain 0 after generation g: a later generation's shift
                    }
                }
                if (lane == 0) {
                    if (trace_slot0 >= 0) {
                        const size_t o = (size_t)(trace_slot0 + gi) * p.nl + c;
                        p.tlogp[o] = dec[6] * nlik + npri;                          // core.py:115 / :178
                        p.tmoved[o] = moved ? 1 : 0; p.ttry[o] = sel; p.tcr[o] = cr_idx; p.tsnk[o] = snk ? 1 : 0;
                    }
                    if (last) { p.lprior[c] = npri; p.llike[c] = nlik; }
                }
            }
            lpri = npri; llik = nlik;
        }
        if (MG) {   // the block's unit sums of THIS generation (contract v3) into its ring slot; stash gi & 1
            const int buf = gi & 1;
            double* xo_b = xo_area + (size_t)buf * 32 * LDP; double* xn_b = xo_b + (size_t)16 * LDP;
            int* bins_b = reinterpret_cast<int*>(xo_area + (size_t)64 * LDP) + 32 * buf;
            int bc, bg;
            adapt_bins(p, g, (int)gc, lane, bc, bg, pr_g, pr_g + p.ncr);
            if (2 * lane < d) { xo_b[wv * LDP + 2 * lane] = xb0; xn_b[wv * LDP + 2 * lane] = xs[0][0]; }
            if (2 * lane + 1 < d) { xo_b[wv * LDP + 2 * lane + 1] = xb1; xn_b[wv * LDP + 2 * lane + 1] = xs[0][1]; }
            if (lane == 0) { bins_b[2 * wv] = bc; bins_b[2 * wv + 1] = bg; }
            __syncthreads();
            const int unit = blockIdx.x, R1 = pub.lag + 1, slot = (int)(g % (uint32_t)R1);
            const long long hs = (long long)g - 1 - pub.lag;
            const double* shift = hs < 0 ? pub.x0start : pub.x0ring + (size_t)(hs % (2 * R1)) * ld;
            adapt_unit_sums(p, xn_b, LDP, xo_b, LDP, min(16, p.nl - 16 * unit), [&](bool isg, int c_) { return bins_b[2 * c_ + (isg ? 1 : 0)]; }, shift,
                            pub.PR + (size_t)slot * pub.pr_stride + (size_t)unit * adapt_nq(p) * ld, pub.PC + (size_t)slot * pub.pc_stride + (size_t)unit * (p.ncr + p.ngamma),
                            (int)threadIdx.x, (int)blockDim.x);
        }
        if (app) { next_app += p.thin; M += (uint32_t)p.N; }
    }
    if (NCH == 1 && !MG && pub.PR) {   // crossover burn-in, blocks of 16 chains (one adaptation unit), k >= 3: the new state into the chain's (dead) row 1, its bins into that row's pad
        int bc, bg;
        adapt_bins(p, g0, (int)gc, lane, bc, bg, probs, probs + p.ncr);
        double* sn = region + LDP;
        if (2 * lane < d) sn[2 * lane] = xs[0][0];
        if (2 * lane + 1 < d) sn[2 * lane + 1] = xs[0][1];
        if (lane == 0) { sn[LDP - 1] = (double)bc; sn[LDP] = (double)bg; }          // (the first element of row 2 is dead as well)
        __syncthreads();
        const int W = mega_mix_wave_doubles(d, k, p.J), unit = blockIdx.x;
        const double* s0 = smem + LDP;
        adapt_unit_sums(p, s0, W, xo_area, LDP, min(16, p.nl - 16 * unit), [&](bool isg, int c_) { return (int)s0[(size_t)c_ * W + LDP - 1 + (isg ? 1 : 0)]; }, pub.shift,
                        pub.PR + (size_t)unit * adapt_nq(p) * ld, pub.PC + (size_t)unit * (p.ncr + p.ngamma), (int)threadIdx.x, (int)blockDim.x);
    }
}
template <bool PB, bool MG = false, int NCH = 1>
__global__ __launch_bounds__(1024) void k_generations_mix(const Params* __restrict__ pp, uint32_t g0, int ngen, uint32_t M0, int64_t trace_slot0, int64_t zappend, int seg0, Publish pub)
{
    generations_wave_body<PB, MG, MixLike, NCH>(pp, g0, ngen, M0, trace_slot0, zappend, seg0, pub);
}
#endif  // DZ_TEMPLATES_ONLY

}  // namespace dz
