// Microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 (cycles per instruction and SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate mfma_f64_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double* out, long long* cyc, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main()
{
    double* out; long long* cyc;
    hipMalloc(&out, 1024 * 1024 * 8); hipMalloc(&cyc, 1024 * 8);
    const int iters = 20000;
    for (int wpb : {4, 8, 16}) {
        for (int nacc : {1, 2, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters);
                if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters);
                if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            float ms = 0;
            {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters);
                if (nacc == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters);
                if (nacc == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(64 * wpb), 0, 0, out, cyc, iters);
                hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
            const double per_simd = (double)iters * nacc * wpb / 4.0;       // MFMAs one SIMD executed (waves round-robin over 4 SIMDs)
            printf("waves/block %2d  independent accumulators %d : %.1f cycles per MFMA and SIMD (%.0f cycles, %.0f MFMAs per SIMD); wall %.3f ms = %.1f TFLOP/s, counter %.2f GHz\n", wpb, nacc, m / per_simd, m, per_simd, ms, 256.0 * wpb * iters * nacc * 2048.0 / (ms * 1e-3) / 1e12, m / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
