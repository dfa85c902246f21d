// Which SIMD does wave w of a 16-wave block run on?  (HW_REG_HW_ID: simd_id = bits [5:4]).  hipcc --offload-arch=gfx950 -O2 wave_simd_map.hip -o wave_simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out, size_t lds_bytes_unused)
{
    extern __shared__ double smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);      // HW_ID, offset 4, 2 bits: SIMD id
    const unsigned full = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = full; }
    if (threadIdx.x == 2048) smem[0] = 1.0;
}
int main()
{
    unsigned* d; const int nb = 512;
    hipMalloc(&d, nb * 16 * 2 * 4);
    for (size_t lds : {(size_t)0, (size_t)140 * 1024}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(1024), lds, 0, d, lds);
        static unsigned h[nb * 32];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        int hist[16][4] = {};
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 16; ++w) hist[w][h[(b * 16 + w) * 2] & 3]++;
        printf("dynamic LDS %zu KB, %d blocks of 16 waves: wave -> SIMD histogram\n", lds / 1024, nb);
        for (int w = 0; w < 16; ++w) printf("  wave %2d: SIMD0 %4d  SIMD1 %4d  SIMD2 %4d  SIMD3 %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
        printf("  block 0: "); for (int w = 0; w < 16; ++w) printf("%u ", h[w * 2] & 3); printf("\n  block 7: "); for (int w = 0; w < 16; ++w) printf("%u ", h[(7 * 16 + w) * 2] & 3); printf("\n");
    }
    return 0;
}
