// dz_device.h -- device-side building blocks of the gfx950 MT-DREAM(ZS) engine:
// the counter-based random contract, the bit-reproducible elementary functions and
// the wave-64 reduction order (DESIGN.md "Random contract", "Elementary functions",
// "Reduction contract").  Compiled with -ffp-contract=off: every fused multiply-add
// is an explicit fma()/fmaf().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DZ_DEV __device__ __forceinline__

namespace dz {

// ---------------------------------------------------------------- random contract
struct u32x4 { uint32_t x, y, z, w; };

// Philox4x32-10, key = 64-bit seed, counter = (idx, stream, chain, generation).
DZ_DEV u32x4 philox(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    // The round keys are recomputed by every call (20 scalar adds).  Left to itself the compiler hoists the whole key
    // schedule out of the callers' loops into 20 SGPRs, which then spill to VGPR lanes: ~40 v_readlane per try in k_propose.
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // (one three-input v_bitop3_b32, truth table 0x96 = a ^ b ^ c, instead of two v_xor_b32: 20 instructions per call)
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

enum : uint32_t { K_CTRL = 0, K_PT = 1, K_DIM = 2, K_BND = 3 };
DZ_DEV uint32_t stream_id(uint32_t kind, uint32_t tr, uint32_t phase) { return kind | (tr << 4) | (phase << 12); }

DZ_DEV double u53(uint32_t hi, uint32_t lo)
{
    return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}
DZ_DEV double u32d(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }

DZ_DEV double u16d(uint32_t h) { return ((double)(h & 0xffffu) + 0.5) * (1.0 / 65536.0); }

// np.random.uniform(low, high) (Dream.py:696) from a 16-bit draw h: low + (high - low)(h + 1/2) 2^-16 as ONE fused multiply-add
// h c1 + c0, c1 = (high - low) 2^-16, c0 = low + (high - low) 2^-17 (both made once on the host: Params::ec1, ec0)
DZ_DEV double uniform16(uint32_t h, double c1, double c0) { return fma((double)(h & 0xffffu), c1, c0); }

// Two independent standard normals (binary32), contract v2 (DESIGN.md "Random contract"; restated in oracle/dreamzs_oracle.c):
// Box-Muller made of +, *, fma, one correctly rounded square root and integer bit operations.
//   radius: u = n 2^-24, n = (w1 >> 8) | 1 odd in [1, 2^24); n = 2^k m with m in [sqrt(1/2), sqrt 2): t = -ln u = (24 - k) ln2 - ln m,
//           ln m = f g(f), f = m - 1; r = sqrt(t)
//   angle:  theta = pi/4 + phi, phi = (pi/2) y, y in [-1/2, 1/2) from the top 23 bits of w2; sqrt2 cos theta = cos phi - sin phi,
//           sqrt2 sin theta = cos phi + sin phi; the quadrant comes from two sign bits (bit 0 of w2 / of w1)
DZ_DEV void normal32_pair(uint32_t w1, uint32_t w2, float& z0, float& z1)
{
    const uint32_t n = (w1 >> 8) | 1u;
    const float nf = (float)n;                                                  // exact (24 bits)
    const uint32_t ix = __float_as_uint(nf) + 0x004afb0du;                      // exponent field = that of n / sqrt(1/2)
    const float Ef = (float)(151 - (int)(ix >> 23));                            // 24 - k
    const float f = __uint_as_float((ix & 0x007fffffu) + 0x3f3504f3u) - 1.0f;
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
    const float tc = __uint_as_float((uint32_t)max((int)__float_as_uint(t), 0));  // a rounding-level negative becomes +0
    // correctly rounded sqrt(tc), tc = 0 or in [2^-24, 17): the hardware approximation (within 1 ulp) and the two-sided correction
    // the compiler's own IEEE expansion uses, without that expansion's scaling of subnormal inputs and its inf / zero cases
    float r = __builtin_amdgcn_sqrtf(tc);
    {
        const float rdn = __uint_as_float(__float_as_uint(r) - 1u), rup = __uint_as_float(__float_as_uint(r) + 1u);
        const float edn = fmaf(-rdn, r, tc), eup = fmaf(-rup, r, tc);
        r = edn <= 0.0f ? rdn : r;
        r = eup > 0.0f ? rup : r;
    }
    const float y = __uint_as_float(0x3f800000u | (w2 >> 9)) - 1.5f;            // exact
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
    const float r0 = __uint_as_float(__float_as_uint(r) | (w2 << 31)), r1 = __uint_as_float(__float_as_uint(r) | (w1 << 31));
    z0 = r0 * (cs - sn); z1 = r1 * (cs + sn);
}

// 16-byte load through the GLOBAL address space.  A pointer read out of a Params held in memory is generic to the compiler,
// which then emits flat_load: that counts on lgkmcnt as well as vmcnt, so every later LDS or scalar-memory wait would also
// wait for archive rows that are requested a whole try ahead on purpose.
typedef double __attribute__((ext_vector_type(2))) dz_d2v;      // (a native vector: double2's copy constructor takes a generic reference)
DZ_DEV double2 gload2(const double* q)
{
    const dz_d2v v = *(const __attribute__((address_space(1))) dz_d2v*)q;
    double2 r; r.x = v.x; r.y = v.y; return r;
}
DZ_DEV void gstore2(double* q, double2 v)
{
    dz_d2v w; w.x = v.x; w.y = v.y;
    *(__attribute__((address_space(1))) dz_d2v*)q = w;
}

// ---------------------------------------------------------------- elementary functions
#define DZ_LN2_HI 6.93147180369123816490e-01
#define DZ_LN2_LO 1.90821492927058770002e-10
#define DZ_DBL_MAX 1.7976931348623157e308

// A double constant that is rebuilt where it is used (two scalar moves) instead of living in a register pair for the whole
// kernel: left alone the compiler hoists the polynomial coefficients below out of the generation loop of the persistent
// kernel and then spills them, and every evaluation waits for a chain of scratch reloads (5 inside one dexp).
DZ_DEV double kd(double c)
{
    unsigned lo = (unsigned)((unsigned long long)__double_as_longlong(c) & 0xffffffffull), hi = (unsigned)((unsigned long long)__double_as_longlong(c) >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// a * b + c with such a constant c as the addend, read straight from its scalar register pair (one v_fma_f64; written as fma(a, b,
// kd(c)) the compiler copies the pair into vector registers first and uses v_fmac: three vector instructions per Horner step)
DZ_DEV double fma_k(double a, double b, double c)
{
#ifdef DZ_NO_FMAK
    return fma(a, b, kd(c));
#endif
    unsigned lo = (unsigned)((unsigned long long)__double_as_longlong(c) & 0xffffffffull), hi = (unsigned)((unsigned long long)__double_as_longlong(c) >> 32);
    asm volatile("" : "+s"(lo), "+s"(hi));
    const unsigned long long cb = ((unsigned long long)hi << 32) | lo;
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(cb));
    return r;
}

DZ_DEV double dexp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return __builtin_huge_val();
    if (x < -745.2) return 0.0;
    const double kf = floor(x * 1.4426950408889634 + 0.5);
    double r = fma(-kf, DZ_LN2_HI, x);
    r = fma(-kf, DZ_LN2_LO, r);
    double p = kd(1.0 / 6227020800.0);
    p = fma_k(p, r, 1.0 / 479001600.0);
    p = fma_k(p, r, 1.0 / 39916800.0);
    p = fma_k(p, r, 1.0 / 3628800.0);
    p = fma_k(p, r, 1.0 / 362880.0);
    p = fma_k(p, r, 1.0 / 40320.0);
    p = fma_k(p, r, 1.0 / 5040.0);
    p = fma_k(p, r, 1.0 / 720.0);
    p = fma_k(p, r, 1.0 / 120.0);
    p = fma_k(p, r, 1.0 / 24.0);
    p = fma_k(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kf;
    if (k < -1000) { p = p * __longlong_as_double((long long)(1023 - 1000) << 52); k += 1000; }
    return p * __longlong_as_double((long long)(k + 1023) << 52);
}

DZ_DEV double dlog(double x)
{
    if (x != x) return x;
    if (x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_huge_val();
    if (x == __builtin_huge_val()) return x;
    int e = 0;
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    if ((b >> 52) == 0) { x = x * 18014398509481984.0; e = -54; b = (unsigned long long)__double_as_longlong(x); }
    e += (int)(b >> 52) - 1023;
    double m = __longlong_as_double((long long)((b & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = kd(1.0 / 23.0);
    p = fma_k(p, z, 1.0 / 21.0);
    p = fma_k(p, z, 1.0 / 19.0);
    p = fma_k(p, z, 1.0 / 17.0);
    p = fma_k(p, z, 1.0 / 15.0);
    p = fma_k(p, z, 1.0 / 13.0);
    p = fma_k(p, z, 1.0 / 11.0);
    p = fma_k(p, z, 1.0 / 9.0);
    p = fma_k(p, z, 1.0 / 7.0);
    p = fma_k(p, z, 1.0 / 5.0);
    p = fma_k(p, z, 1.0 / 3.0);
    const double t = 2.0 * s;
    const double lm = fma(t * z, p, t);
    const double ef = (double)e;
    return fma(ef, DZ_LN2_HI, fma(ef, DZ_LN2_LO, lm));
}

DZ_DEV double nan_to_num(double x)
{
    if (x != x) return 0.0;
    if (x == __builtin_huge_val()) return DZ_DBL_MAX;
    if (x == -__builtin_huge_val()) return -DZ_DBL_MAX;
    return x;
}
DZ_DEV bool is_finite(double x) { return (x == x) && (x != __builtin_huge_val()) && (x != -__builtin_huge_val()); }

// ---------------------------------------------------------------- wave-64 reductions
// lane(j) = (j >> 1) & 63; a lane adds its own dimensions in increasing j; the 64 lane
// partials are combined by an xor butterfly (32,16,...,1).  Every lane ends with the total.
// One DPP row rotation (lanes rotate right by ROR inside each row of 16); no LDS crossbar involved.
template <int ROR>
DZ_DEV double row_ror(double v)
{
    const long long b = __double_as_longlong(v);
    // every lane of a row rotation reads a live lane, so the destination needs no defined previous value
    // (update_dpp with old = 0 costs two extra v_mov per rotation)
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffll), 0x120 + ROR, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x120 + ROR, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// p[i] + p[i ^ off] for off = 8,4,2,1 inside rows of 16 lanes.  After the step with offset 2m the values have
// period 2m within the row, so rotating by m fetches a lane that holds exactly the bits of lane i ^ m.
DZ_DEV double bfly16(double v)
{
    v = v + row_ror<8>(v);
    v = v + row_ror<4>(v);
    v = v + row_ror<2>(v);
    v = v + row_ror<1>(v);
    return v;
}
// x[i] + x[i ^ 32] and x[i] + x[i ^ 16] with gfx950's lane-swap instructions (VALU, no LDS crossbar round trip):
// v_permlane32_swap exchanges the upper half of its first operand with the lower half of its second; with both operands
// equal to x the two results hold (x_lo, x_lo) and (x_hi, x_hi) by halves, so their sum is x[i] + x[i ^ 32] in every lane
// (addition commutes, the bits are those of the xor butterfly).  v_permlane16_swap does the same for odd and even rows.
DZ_DEV double swap32_sum(double v)
{
    const long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane32_swap((int)(b & 0xffffffffll), (int)(b & 0xffffffffll), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((int)(b >> 32), (int)(b >> 32), false, false);
    const double a = __longlong_as_double(((long long)(int)hi[0] << 32) | (unsigned int)lo[0]);
    const double c = __longlong_as_double(((long long)(int)hi[1] << 32) | (unsigned int)lo[1]);
    return a + c;
}
DZ_DEV double swap16_sum(double v)
{
    const long long b = __double_as_longlong(v);
    const auto lo = __builtin_amdgcn_permlane16_swap((int)(b & 0xffffffffll), (int)(b & 0xffffffffll), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((int)(b >> 32), (int)(b >> 32), false, false);
    const double a = __longlong_as_double(((long long)(int)hi[0] << 32) | (unsigned int)lo[0]);
    const double c = __longlong_as_double(((long long)(int)hi[1] << 32) | (unsigned int)lo[1]);
    return a + c;
}
DZ_DEV double wave_bfly(double v)
{
    v = swap32_sum(v);
    v = swap16_sum(v);
    return bfly16(v);       // all four rows now hold the same 16 values
}
// FOUR wave sums for little more than the price of one: the two lane-swap steps are where the values merge -- v_permlane32_swap with
// (a, b) leaves a's two halves in the lower lanes of the two results and b's in the upper lanes, so ONE addition makes a[i] + a[i ^ 32] in
// lanes 0-31 and b[i] + b[i ^ 32] in lanes 32-63; v_permlane16_swap does the same for rows -- and the four DPP row steps then finish the
// four values side by side.  Every value is added along the same tree as in wave_bfly (the same pairs at every step; addition commutes):
// the same bits.  Returns in row 0 (lanes 0-15) the total of a, row 1 of c, row 2 of b, row 3 of d, in every lane of the row.
DZ_DEV void swap32_pair(double a, double b, double& lo_sum)
{
    const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane32_swap((int)(ba & 0xffffffffll), (int)(bb & 0xffffffffll), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((int)(ba >> 32), (int)(bb >> 32), false, false);
    const double x = __longlong_as_double(((long long)(int)hi[0] << 32) | (unsigned int)lo[0]);
    const double y = __longlong_as_double(((long long)(int)hi[1] << 32) | (unsigned int)lo[1]);
    lo_sum = x + y;
}
DZ_DEV void swap16_pair(double a, double b, double& out)
{
    const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((int)(ba & 0xffffffffll), (int)(bb & 0xffffffffll), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((int)(ba >> 32), (int)(bb >> 32), false, false);
    const double x = __longlong_as_double(((long long)(int)hi[0] << 32) | (unsigned int)lo[0]);
    const double y = __longlong_as_double(((long long)(int)hi[1] << 32) | (unsigned int)lo[1]);
    out = x + y;
}
DZ_DEV double wave_bfly4(double a, double b, double c, double d)
{
    double ab, cd, q;
    swap32_pair(a, b, ab);          // lanes 0-31: a, 32-63: b
    swap32_pair(c, d, cd);          // lanes 0-31: c, 32-63: d
    swap16_pair(ab, cd, q);         // rows: a, c, b, d
    return bfly16(q);
}
// the row (0..3) of wave_bfly4's result that holds its u-th argument
DZ_DEV int bfly4_row(int u) { return u == 1 ? 2 : (u == 2 ? 1 : u); }
// two sums: the total of a in lanes 0-31, of b in lanes 32-63
DZ_DEV double wave_bfly2(double a, double b)
{
    double ab;
    swap32_pair(a, b, ab);
    return bfly16(swap16_sum(ab));
}

// value of lane ln (wave-uniform index) -- v_readlane, no LDS crossbar round trip
DZ_DEV double readlane_f64(double v, int ln)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), ln), hi = __builtin_amdgcn_readlane((int)(b >> 32), ln);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// maximum over each row of 16 lanes, in every lane of the row (fmax is exact, so the order is immaterial)
DZ_DEV double rowmax16(double v)
{
    v = fmax(v, row_ror<8>(v));
    v = fmax(v, row_ror<4>(v));
    v = fmax(v, row_ror<2>(v));
    v = fmax(v, row_ror<1>(v));
    return v;
}
DZ_DEV int wave_isum(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// first category m with u < p0 + .. + pm
DZ_DEV int invcdf(const double* __restrict__ p, int n, double u)
{
    double c = 0.0;
    for (int m = 0; m < n; ++m) { c = c + p[m]; if (u < c) return m; }
    return n - 1;
}

struct Ctrl { double u_snk, u_cr, u_de, u_glev, u_sel, u_acc; };
DZ_DEV Ctrl draw_ctrl(uint32_t k0, uint32_t k1, uint32_t gc, uint32_t g)
{
    const uint32_t s = stream_id(K_CTRL, 0, 0);
    Ctrl c;
    u32x4 w = philox(k0, k1, 0, s, gc, g); c.u_snk = u53(w.x, w.y); c.u_cr = u53(w.z, w.w);
    w = philox(k0, k1, 1, s, gc, g); c.u_de = u53(w.x, w.y); c.u_glev = u53(w.z, w.w);
    w = philox(k0, k1, 2, s, gc, g); c.u_sel = u53(w.x, w.y); c.u_acc = u53(w.z, w.w);
    return c;
}

}  // namespace dz
