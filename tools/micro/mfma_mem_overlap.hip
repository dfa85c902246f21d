// Microbenchmark: do FP64 MFMAs of one set of waves overlap with LDS reads / LDS writes / global loads of OTHER waves of the same
// SIMD on gfx950?  16 waves per block (4 per SIMD); waves with ((wave >> 2) & 1) == 0 run the MFMA loop, the others a memory loop.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_mem_overlap mfma_mem_overlap.hip
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
// kind 4: global_load_lds_dwordx4 x 4;  kind 0: ds_read_b64 x 8 per iteration; 1: ds_write_b64 x 8; 2: global_load_dwordx4 x 4 (L2-resident 1 MB window); 3: ds_read_b128 x 8
__device__ __forceinline__ double loop_mem(int kind, int iters, double* lds, const double* g, int tid)
{
    double s = 0;
    if (kind == 0) {
        for (int it = 0; it < iters; ++it) {
            double v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[i]) : "v"((unsigned)(tid * 8)), "n"(0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(v[i]));
        }
    } else if (kind == 1) {
        double v = tid;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)(tid * 8)), "v"(v) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (kind == 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* gp = reinterpret_cast<const d2*>(g) + tid;
        for (int it = 0; it < iters; ++it) {
            d2 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_nontemporal_load(gp + ((it * 4 + i) & 63) * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[i].x;
        }
    } else if (kind == 4) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2* gp = reinterpret_cast<const d2*>(g) + tid;
        // direct-to-LDS: the wave's 1 KB lands at (wave-uniform base) + lane * 16
        __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(lds + ((tid >> 6) & 7) * 128);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 4 + i) & 63) * 1024), dst, 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else {
        typedef double d2 __attribute__((ext_vector_type(2)));
        for (int it = 0; it < iters; ++it) {
            d2 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"((unsigned)(tid * 16)));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(v[i]));
        }
    }
    return s;
}

// mode 0: MFMA-role waves only (others idle); 1: memory-role waves only; 2: both
__global__ void k(double* out, long long* cyc, const double* g, int mode, int kind, int it_m, int it_v)
{
    __shared__ __attribute__((aligned(16))) double lds[2048];
    const int wave = threadIdx.x >> 6;
    const bool mf = ((wave >> 2) & 1) == 0;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 1024] = 1.0;
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3, r = 0;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (mf) { if (mode != 1) r = loop_mfma16(it_m, a, b); }
    else { if (mode != 0) r = loop_mem(kind, it_v, lds, g, threadIdx.x & 511); }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

static void run(const char* name, int mode, int kind, int it_m, int it_v, double* out, long long* cyc, const double* g)
{
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, cyc, g, mode, kind, it_m, it_v); hipDeviceSynchronize(); }
    static long long h[256 * 16]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mm = 0, mv = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 16; ++w) { if (((w >> 2) & 1) == 0) mm += h[b * 16 + w]; else mv += h[b * 16 + w]; }
    mm /= 256 * 8; mv /= 256 * 8;
    printf("%-52s MFMA-role waves %9.0f ticks, memory-role waves %9.0f ticks\n", name, mm, mv);
}

int main()
{
    double* out; long long* cyc; double* g;
    hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 16 * 8); hipMalloc(&g, 64 * 1024 * 16 + 1024 * 16); hipMemset(g, 0, 64 * 1024 * 16 + 1024 * 16);
    const int IM = 4000;
    const char* kn[5] = {"ds_read_b64 x8", "ds_write_b64 x8", "global_load_dwordx4 x4", "ds_read_b128 x8", "global_load_lds_dwordx4 x4"};
    const int iv[5] = {8000, 4000, 1500, 4000, 1500};
    run("8 waves MFMA 16x16x4 f64 alone", 0, 0, IM, 0, out, cyc, g);
    for (int kind = 0; kind < 5; ++kind) {
        char nm[128];
        snprintf(nm, sizeof nm, "8 waves %s alone", kn[kind]); run(nm, 1, kind, IM, iv[kind], out, cyc, g);
        snprintf(nm, sizeof nm, "8 waves MFMA + 8 waves %s", kn[kind]); run(nm, 2, kind, IM, iv[kind], out, cyc, g);
    }
    return 0;
}
