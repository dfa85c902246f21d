// Probe 3: f64 MFMA rate as a function of waves per SIMD and independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k_rate(double* out, int iters)
{
    double4_t c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = double4_t{0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[j], 0, 0, 0);
    double s = 0; for (int i = 0; i < NACC; ++i) s += c[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(double* out, int blocks, int threads, const char* tag)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k_rate<NACC>, dim3(blocks), dim3(threads), 0, 0, out, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)blocks * (threads / 64) * iters * NACC;
        if (rep) printf("%-28s NACC=%d: %.1f TFLOP/s, %.1f cycles/MFMA/SIMD at 2.4 GHz\n", tag, NACC, n_mfma * 2048 / ms / 1e9,
                        ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0));
    }
}
int main()
{
    double* out; (void)hipMalloc(&out, 4096 * 256 * 8);
    run<1>(out, 256, 256, "1 wave/SIMD"); run<2>(out, 256, 256, "1 wave/SIMD"); run<4>(out, 256, 256, "1 wave/SIMD"); run<7>(out, 256, 256, "1 wave/SIMD");
    run<7>(out, 256, 512, "2 waves/SIMD"); run<7>(out, 512, 512, "4 waves/SIMD"); run<1>(out, 1024, 512, "8 waves/SIMD");
    run<7>(out, 128, 256, "1 wave/SIMD on 128 CUs");
    return 0;
}
