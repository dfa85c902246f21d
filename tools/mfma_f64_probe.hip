// Probe: is v_mfma_f64_16x16x4_f64 on gfx950 bit-identical to an ascending-k fma chain, and how fast is it?
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_f64_probe.hip -o /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

// A is 16x4 (row i, col k), B is 4x16 (row k, col j). Lane l supplies a = A[l%16][l/16], b = B[l/16][l%16].
// Output D (16x16): lane l holds D[4*(l/16) + v][l%16], v = 0..3.
__global__ void k_probe(const double* A, const double* B, const double* Cin, double* D)
{
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + (l / 16)];
    const double b = B[(l / 16) * 16 + (l % 16)];
    double4_t c;
    for (int v = 0; v < 4; ++v) c[v] = Cin[(4 * (l / 16) + v) * 16 + (l % 16)];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + (l % 16)] = c[v];
}

__global__ void k_rate_mfma(double* out, int iters)
{
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void k_rate_fma(double* out, int iters)
{
    double c[8]; for (int i = 0; i < 8; ++i) c[i] = i;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = fma(a, c[j], b);
    double s = 0; for (int i = 0; i < 8; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_rate_philox(unsigned* out, int iters)
{
    unsigned c0 = threadIdx.x, c1 = blockIdx.x, c2 = 7, c3 = 9, acc = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned k0 = 0x1234, k1 = 0x5678, x0 = c0 + i, x1 = c1, x2 = c2, x3 = c3;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const unsigned long long p0 = (unsigned long long)0xD2511F53u * x0, p1 = (unsigned long long)0xCD9E8D57u * x2;
            const unsigned n0 = (unsigned)(p1 >> 32) ^ x1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ x3 ^ k1;
            x1 = (unsigned)p1; x3 = (unsigned)p0; x0 = n0; x2 = n2; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
        acc ^= x0 ^ x1 ^ x2 ^ x3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    std::vector<double> A(64), B(64), C(256), D(256);
    srand(3);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * pow(2.0, (rand() % 40) - 20); };
    int bad_chain = 0, bad_rev = 0, bad_pair = 0;
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dC, 256 * 8); hipMalloc(&dD, 256 * 8);
    for (int trial = 0; trial < 200; ++trial) {
        for (auto& x : A) x = rnd(); for (auto& x : B) x = rnd(); for (auto& x : C) x = rnd();
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double c = C[i * 16 + j], r = c, q = c;
            for (int k = 0; k < 4; ++k) c = fma(A[i * 4 + k], B[k * 16 + j], c);
            for (int k = 3; k >= 0; --k) r = fma(A[i * 4 + k], B[k * 16 + j], r);
            q = (A[i*4+0]*B[0*16+j] + A[i*4+1]*B[1*16+j]) + (A[i*4+2]*B[2*16+j] + A[i*4+3]*B[3*16+j]) + q;
            if (c != D[i * 16 + j]) bad_chain++;
            if (r != D[i * 16 + j]) bad_rev++;
            if (q != D[i * 16 + j]) bad_pair++;
        }
    }
    printf("mfma_f64_16x16x4 vs ascending fma chain: %d mismatches; vs descending: %d; vs pairwise: %d (of %d)\n", bad_chain, bad_rev, bad_pair, 200 * 256);
    // throughput
    double* out; hipMalloc(&out, 1024 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 1024;   // 4 waves per block -> 4096 waves = 4 per SIMD
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k_rate_mfma, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 4 * (16.0 * 16 * 4 * 2);
        if (rep) printf("mfma f64: %.1f TFLOP/s (%.3f ms)\n", flops / ms / 1e9, ms);
        hipEventRecord(e0); hipLaunchKernelGGL(k_rate_fma, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        flops = (double)blocks * 256 * iters * 8 * 2;
        if (rep) printf("valu fma f64: %.1f TFLOP/s (%.3f ms)\n", flops / ms / 1e9, ms);
        hipEventRecord(e0); hipLaunchKernelGGL(k_rate_philox, dim3(blocks), dim3(256), 0, 0, (unsigned*)out, 2000); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("philox4x32-10: %.1f G calls/s (%.3f ms) = %.1f lane-cycles/call at 2.4GHz x 256CU x 128 lanes\n", (double)blocks * 256 * 2000 / ms / 1e6, ms,
                        2.4e9 * 256 * 128 / ((double)blocks * 256 * 2000 / (ms * 1e-3)));
    }
    return 0;
}
