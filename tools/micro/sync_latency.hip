// Host-side cost of "launch + hipStreamSynchronize" around a kernel of known length, per device schedule flag
// (usage: sync_latency <0 auto | 1 spin | 2 yield | 4 blocking>).  hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin(long long cycles, long long* out) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < cycles) {} if (out) *out = t0; }
int main(int argc, char** argv)
{
    const unsigned flag = argc > 1 ? (unsigned)atoi(argv[1]) : 0u;
    if (hipSetDeviceFlags(flag) != hipSuccess) { printf("hipSetDeviceFlags(%u) failed\n", flag); return 1; }
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (long long us : {0LL, 300LL}) {
        const long long cyc = us * 100;                    // wall_clock64 ticks at 100 MHz
        for (int i = 0; i < 50; ++i) { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, cyc, nullptr); hipStreamSynchronize(st); }
        const int n = 500;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) { hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, cyc, nullptr); hipLaunchKernelGGL(spin, dim3(256), dim3(64), 0, st, cyc, nullptr); hipStreamSynchronize(st); }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("flag %u: two %lld-us kernels + sync: %.1f us per round (%.1f beyond the kernels)\n", flag, us, 1e6 * dt / n, 1e6 * dt / n - 2.0 * us);
    }
    return 0;
}
