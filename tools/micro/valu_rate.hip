// Microbenchmark: issue cost of the VALU instructions the proposal code is made of, on gfx950 (cycles per
// instruction and SIMD, 4 waves per SIMD, 8 independent dependency chains per wave).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 32-bit destination, two 32-bit sources
#define K32(NAME, INSN)                                                                                          \
    __global__ void NAME(unsigned* out, long long* cyc, int iters)                                               \
    {                                                                                                            \
        unsigned r[8], b = threadIdx.x * 2654435761u + 12345u;                                                   \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 40503u + i * 7919u + 1u;                                \
        __syncthreads();                                                                                         \
        const long long t0 = __builtin_readcyclecounter();                                                       \
        for (int it = 0; it < iters; ++it) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSN " %0, %0, %1" : "+v"(r[i]) : "v"(b)); \
            }                                                                                                    \
        }                                                                                                        \
        const long long t1 = __builtin_readcyclecounter();                                                       \
        unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                          \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                         \
    }
// 32-bit destination, three 32-bit sources
#define K32_3(NAME, INSN, TAIL)                                                                                  \
    __global__ void NAME(unsigned* out, long long* cyc, int iters)                                               \
    {                                                                                                            \
        unsigned r[8], b = threadIdx.x * 2654435761u + 12345u;                                                   \
        for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 40503u + i * 7919u + 1u;                                \
        __syncthreads();                                                                                         \
        const long long t0 = __builtin_readcyclecounter();                                                       \
        for (int it = 0; it < iters; ++it) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(INSN " %0, %0, %1, %1" TAIL : "+v"(r[i]) : "v"(b)); \
            }                                                                                                    \
        }                                                                                                        \
        const long long t1 = __builtin_readcyclecounter();                                                       \
        unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                          \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                         \
    }
// 64-bit (register pair) destination and sources, NS sources after the destination
#define K64(NAME, ASMSTR)                                                                                        \
    __global__ void NAME(unsigned* out, long long* cyc, int iters)                                               \
    {                                                                                                            \
        double r[8], b = 1.0 + threadIdx.x * 1e-9;                                                               \
        for (int i = 0; i < 8; ++i) r[i] = 1.0 + threadIdx.x * 1e-6 + i * 1e-3;                                  \
        __syncthreads();                                                                                         \
        const long long t0 = __builtin_readcyclecounter();                                                       \
        for (int it = 0; it < iters; ++it) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(r[i]) : "v"(b));        \
            }                                                                                                    \
        }                                                                                                        \
        const long long t1 = __builtin_readcyclecounter();                                                       \
        double s = 0; for (int i = 0; i < 8; ++i) s += r[i];                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(long long)s;                                     \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                         \
    }
// conversions: destination and source of different widths
#define KCVT(NAME, ASMSTR, DT, ST)                                                                               \
    __global__ void NAME(unsigned* out, long long* cyc, int iters)                                               \
    {                                                                                                            \
        DT r[8]; ST b[8];                                                                                        \
        for (int i = 0; i < 8; ++i) { r[i] = (DT)0; b[i] = (ST)(threadIdx.x + i + 1); }                          \
        __syncthreads();                                                                                         \
        const long long t0 = __builtin_readcyclecounter();                                                       \
        for (int it = 0; it < iters; ++it) {                                                                     \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                      \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "=v"(r[i]) : "v"(b[i]));     \
            }                                                                                                    \
        }                                                                                                        \
        const long long t1 = __builtin_readcyclecounter();                                                       \
        double s = 0; for (int i = 0; i < 8; ++i) s += (double)r[i];                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(long long)s;                                     \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                         \
    }

K32(k_add_u32, "v_add_u32")
K32(k_xor_b32, "v_xor_b32")
K32(k_mul_lo_u32, "v_mul_lo_u32")
K32(k_mul_hi_u32, "v_mul_hi_u32")
K32(k_mul_u32_u24, "v_mul_u32_u24")
K32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24")
K32(k_mul_f32, "v_mul_f32")
K32(k_lshlrev, "v_lshlrev_b32")
K32_3(k_alignbit, "v_alignbit_b32", "")
K32_3(k_bitop3, "v_bitop3_b32", " bitop3:0x96")
K32_3(k_fma_f32, "v_fma_f32", "")
K32_3(k_mad_u32_u24, "v_mad_u32_u24", "")
K32_3(k_add3_u32, "v_add3_u32", "")
K32_3(k_xad_u32, "v_xad_u32", "")
K64(k_fma_f64, "v_fma_f64 %0, %0, %1, %1")
K64(k_mul_f64, "v_mul_f64 %0, %0, %1")
K64(k_add_f64, "v_add_f64 %0, %0, %1")
K64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")
K64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
K64(k_lshlrev_b64, "v_lshlrev_b64 %0, 1, %0")
__global__ void k_mad_u64_u32(unsigned* out, long long* cyc, int iters)
{
    unsigned long long r[8]; unsigned b = threadIdx.x * 2654435761u + 12345u;
    for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 40503u + i * 7919u + 1u;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, 0" : "=v"(r[i]) : "v"((unsigned)r[i]), "v"(b) : "s20", "s21");
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    unsigned long long s = 0; for (int i = 0; i < 8; ++i) s ^= r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_mul_lo_hi(unsigned* out, long long* cyc, int iters)
{   // the 32 x 32 -> 64 product as two instructions
    unsigned lo[8], hi[8]; unsigned b = threadIdx.x * 2654435761u + 12345u;
    for (int i = 0; i < 8; ++i) { lo[i] = threadIdx.x * 40503u + i * 7919u + 1u; hi[i] = 0; }
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_hi_u32 %1, %0, %2\n v_mul_lo_u32 %0, %0, %2" : "+v"(lo[i]), "=v"(hi[i]) : "v"(b));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= lo[i] ^ hi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
K64(k_rcp_f64, "v_rcp_f64 %0, %0")
K64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KCVT(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %1", double, unsigned)
KCVT(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1", double, float)
KCVT(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %1", float, unsigned)
KCVT(k_cvt_u32_f32, "v_cvt_u32_f32 %0, %1", unsigned, float)
KCVT(k_cvt_f32_f64, "v_cvt_f32_f64 %0, %1", float, double)
KCVT(k_cvt_i32_f64, "v_cvt_i32_f64 %0, %1", int, double)
KCVT(k_rcp_f32, "v_rcp_f32 %0, %1", float, float)
KCVT(k_sqrt_f32, "v_sqrt_f32 %0, %1", float, float)
KCVT(k_rsq_f32, "v_rsq_f32 %0, %1", float, float)
KCVT(k_log_f32, "v_log_f32 %0, %1", float, float)
KCVT(k_exp_f32, "v_exp_f32 %0, %1", float, float)
KCVT(k_sin_f32, "v_sin_f32 %0, %1", float, float)
KCVT(k_floor_f32, "v_floor_f32 %0, %1", float, float)
KCVT(k_floor_f64, "v_floor_f64 %0, %1", double, double)
KCVT(k_mov_b32, "v_mov_b32 %0, %1", unsigned, unsigned)
KCVT(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf", unsigned, unsigned)
KCVT(k_readlane, "v_readlane_b32 s20, %1, 3\n v_mov_b32 %0, s20", unsigned, unsigned)

typedef void (*kern_t)(unsigned*, long long*, int);
struct Ent { const char* name; kern_t f; int per; };

int main()
{
    unsigned* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 4000, wpb = 16;
    const Ent tab[] = {
        {"v_add_u32", k_add_u32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_lshlrev_b32", k_lshlrev, 1}, {"v_alignbit_b32", k_alignbit, 1},
        {"v_bitop3_b32", k_bitop3, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_xad_u32", k_xad_u32, 1},
        {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1},
        {"v_lshlrev_b64", k_lshlrev_b64, 1}, {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mul_hi+v_mul_lo", k_mul_lo_hi, 2},
        {"v_mul_f32", k_mul_f32, 1}, {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_mul_f32", k_pk_mul_f32, 1},
        {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_rcp_f64", k_rcp_f64, 1}, {"v_sqrt_f64", k_sqrt_f64, 1},
        {"v_cvt_f64_u32", k_cvt_f64_u32, 1}, {"v_cvt_f64_f32", k_cvt_f64_f32, 1}, {"v_cvt_f32_u32", k_cvt_f32_u32, 1}, {"v_cvt_u32_f32", k_cvt_u32_f32, 1},
        {"v_cvt_f32_f64", k_cvt_f32_f64, 1}, {"v_cvt_i32_f64", k_cvt_i32_f64, 1},
        {"v_rcp_f32", k_rcp_f32, 1}, {"v_sqrt_f32", k_sqrt_f32, 1}, {"v_rsq_f32", k_rsq_f32, 1}, {"v_log_f32", k_log_f32, 1}, {"v_exp_f32", k_exp_f32, 1},
        {"v_sin_f32", k_sin_f32, 1}, {"v_floor_f32", k_floor_f32, 1}, {"v_floor_f64", k_floor_f64, 1},
        {"v_mov_b32", k_mov_b32, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1}, {"v_readlane+v_mov", k_readlane, 2},
    };
    for (const Ent& e : tab) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(e.f, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
        const double per_simd = (double)iters * 32 * wpb / 4.0;      // instructions (pairs for the readlane entry) one SIMD issued
        printf("%-20s %6.2f counter ticks per instruction and SIMD\n", e.name, m / per_simd);
    }
    // the cycle counter runs at a fixed 100 MHz: convert with a known 4-cycle instruction (v_add_u32) as the unit
    return 0;
}
