// Microbenchmark: (1) issue cost of v_mfma_f64_4x4x4_4b_f64 next to v_mfma_f64_16x16x4_f64; (2) whether FP64 MFMAs and VALU work of
// OTHER waves of the same SIMD overlap on gfx950.  4 waves per SIMD; waves with (wave >> 2) & 1 == role run one loop, the others
// the other loop (so every SIMD hosts 2 + 2), or all waves run the same loop.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double loop_mfma16(int iters, double a, double b)
{
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
    }
    return acc0[0] + acc1[1];
}
__device__ __forceinline__ double loop_mfma4(int iters, double a, double b)
{
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, acc3, 0, 0, 0);
    }
    return acc0 + acc1 + acc2 + acc3;
}
template <int KIND>      // 0: f32 fma, 1: u32 mul/xor (Philox-like), 2: f64 fma
__device__ __forceinline__ double loop_valu(int iters, unsigned seed)
{
    if (KIND == 0) {
        float r[8]; for (int i = 0; i < 8; ++i) r[i] = 1.0f + seed * 1e-6f + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0000001f));
        }
        float s = 0; for (int i = 0; i < 8; ++i) s += r[i]; return s;
    } else if (KIND == 1) {
        unsigned r[8]; for (int i = 0; i < 8; ++i) r[i] = seed * 40503u + i * 7919u + 1u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r[i]) : "v"(0xD2511F53u));
        }
        unsigned s = 0; for (int i = 0; i < 8; ++i) s ^= r[i]; return (double)s;
    } else {
        double r[8]; for (int i = 0; i < 8; ++i) r[i] = 1.0 + seed * 1e-9 + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[i]) : "v"(1.0000000001));
        }
        double s = 0; for (int i = 0; i < 8; ++i) s += r[i]; return s;
    }
}

// mode: 0 all waves MFMA16; 1 all waves MFMA4; 2.. all waves VALU kind (mode-2); 10+kind: half MFMA16 / half VALU kind
template <int KIND>
__global__ void k(double* out, long long* cyc, int mode, int it_m, int it_v)
{
    const int wave = threadIdx.x >> 6;
    const bool mf = ((wave >> 2) & 1) == 0;          // waves 0-3 (one per SIMD) and 8-11: MFMA role; 4-7, 12-15: VALU role
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3, r = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mode == 0) r = loop_mfma16(it_m, a, b);
    else if (mode == 1) r = loop_mfma4(it_m, a, b);
    else if (mode == 2) r = loop_valu<KIND>(it_v, threadIdx.x);
    else { if (mf) r = loop_mfma16(it_m, a, b); else r = loop_valu<KIND>(it_v, threadIdx.x); }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int mode, int it_m, int it_v, double* out, long long* cyc)
{
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, out, cyc, mode, it_m, it_v); hipDeviceSynchronize(); }
    static long long h[256 * 16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mm = 0, mv = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) { if (((w >> 2) & 1) == 0) mm += h[b * 16 + w]; else mv += h[b * 16 + w]; }
    mm /= 256 * 8; mv /= 256 * 8;
    printf("%-44s MFMA-role waves %9.0f ticks, VALU-role waves %9.0f ticks\n", name, mm, mv);
}

int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 16 * 8);
    const int IM = 4000, IV = 4000;        // MFMA loop: 2 x IM 16x16x4 per wave; VALU loop: 32 x IV instructions per wave
    run<0>("all 16 waves: MFMA 16x16x4 f64 (2 per iter)", 0, IM, IV, out, cyc);
    run<0>("all 16 waves: MFMA 4x4x4 f64 (4 per iter)", 1, IM, IV, out, cyc);
    run<0>("all 16 waves: VALU f32 fma", 2, IM, IV, out, cyc);
    run<1>("all 16 waves: VALU u32 mul", 2, IM, IV, out, cyc);
    run<2>("all 16 waves: VALU f64 fma", 2, IM, IV, out, cyc);
    run<0>("8 waves MFMA16 + 8 waves VALU f32 fma", 10, IM, IV, out, cyc);
    run<1>("8 waves MFMA16 + 8 waves VALU u32 mul", 10, IM, IV, out, cyc);
    run<2>("8 waves MFMA16 + 8 waves VALU f64 fma", 10, IM, IV, out, cyc);
    run<0>("8 waves MFMA16 alone (others idle: it_v = 0)", 10, IM, 0, out, cyc);
    run<0>("8 waves VALU f32 alone (it_m = 0)", 10, 0, IV, out, cyc);
    return 0;
}
