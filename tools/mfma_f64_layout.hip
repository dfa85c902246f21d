// Probe 2: discover the D layout of v_mfma_f64_16x16x4_f64 with exactly-representable data, then test
// accumulation-order hypotheses under that layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k_raw(const double* a_in, const double* b_in, const double* c_in, double* d_out)
{
    const int l = threadIdx.x;
    double4_t c; for (int v = 0; v < 4; ++v) c[v] = c_in[l * 4 + v];
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a_in[l], b_in[l], c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d_out[l * 4 + v] = c[v];
}
int main()
{
    std::vector<double> a(64), b(64), c(256), d(256);
    double *da, *db, *dc, *dd; 
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dc, 2048); (void)hipMalloc(&dd, 2048);
    // layout discovery: A[i][k] = 1+i+100k encoded per lane hypothesis a[l]: i=l%16,k=l/16 ; B[k][j] = 1 if k==0 else 0 -> D[i][j] = A[i][0]
    auto run = [&]() { (void)hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice); (void)hipMemcpy(dc, c.data(), 2048, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_raw, dim3(1), dim3(64), 0, 0, da, db, dc, dd); (void)hipMemcpy(d.data(), dd, 2048, hipMemcpyDeviceToHost); };
    for (int l = 0; l < 64; ++l) { a[l] = 1000 * (l / 16) + (l % 16); b[l] = ((l / 16) == 2) ? 1.0 + (l % 16) * 0.001953125 : 0.0; }
    for (auto& x : c) x = 0;
    run();
    // expected: D[i][j] = A[i][2]*B[2][j] = (2000+i)*(1+j/512)
    printf("lane0: %.6f %.6f %.6f %.6f | lane1: %.6f %.6f | lane16: %.6f %.6f %.6f %.6f | lane32: %.6f lane48: %.6f\n", d[0], d[1], d[2], d[3], d[4], d[5], d[64], d[65], d[66], d[67], d[128], d[192]);
    // decode: value = (2000+i)*(1+j/512)
    int ok_layout1 = 1, ok_layout2 = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        int j = l % 16; int i1 = 4 * (l / 16) + v; int i2 = (l / 16) + 4 * v;
        if (d[l * 4 + v] != (2000 + i1) * (1 + j / 512.0)) ok_layout1 = 0;
        if (d[l * 4 + v] != (2000 + i2) * (1 + j / 512.0)) ok_layout2 = 0;
    }
    printf("layout i=4*(l/16)+v: %d ; layout i=(l/16)+4v: %d\n", ok_layout1, ok_layout2);
    // C layout check: set c so that c value encodes (l,v); a=b=0 -> D=C at same slot trivially. skip.
    // order hypotheses with random data under each layout
    srand(5);
    auto rnd = []() { return (rand() / (double)RAND_MAX - 0.5) * pow(2.0, (rand() % 30) - 15); };
    long bad[2][6] = {{0}};
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<double> A(64), B(64), C(256);
        for (auto& x : A) x = rnd(); for (auto& x : B) x = rnd(); for (auto& x : C) x = rnd();
        for (int lay = 0; lay < 2; ++lay) {
            for (int l = 0; l < 64; ++l) { a[l] = A[(l % 16) * 4 + l / 16]; b[l] = B[(l / 16) * 16 + l % 16];
                for (int v = 0; v < 4; ++v) { int i = lay == 0 ? 4 * (l / 16) + v : (l / 16) + 4 * v; c[l * 4 + v] = C[i * 16 + l % 16]; } }
            run();
            for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
                int j = l % 16; int i = lay == 0 ? 4 * (l / 16) + v : (l / 16) + 4 * v;
                double got = d[l * 4 + v];
                double x0 = C[i * 16 + j]; for (int k = 0; k < 4; ++k) x0 = fma(A[i * 4 + k], B[k * 16 + j], x0);
                double x1 = C[i * 16 + j]; for (int k = 3; k >= 0; --k) x1 = fma(A[i * 4 + k], B[k * 16 + j], x1);
                double x2 = C[i * 16 + j]; for (int k = 0; k < 4; ++k) x2 = x2 + A[i * 4 + k] * B[k * 16 + j];
                __float128 q = C[i * 16 + j]; for (int k = 0; k < 4; ++k) q += (__float128)A[i * 4 + k] * (__float128)B[k * 16 + j];
                double x3 = (double)q;
                double x4 = 0; for (int k = 0; k < 4; ++k) x4 = fma(A[i * 4 + k], B[k * 16 + j], x4); x4 += C[i * 16 + j];
                double x5 = fma(A[i*4+3], B[3*16+j], fma(A[i*4+2], B[2*16+j], 0.0)) + fma(A[i*4+1], B[1*16+j], fma(A[i*4+0], B[0*16+j], C[i*16+j]));
                bad[lay][0] += got != x0; bad[lay][1] += got != x1; bad[lay][2] += got != x2; bad[lay][3] += got != x3; bad[lay][4] += got != x4; bad[lay][5] += got != x5;
            }
        }
    }
    for (int lay = 0; lay < 2; ++lay) printf("layout %d: mismatches asc-fma %ld desc-fma %ld asc-mul-add %ld exact-dot %ld dot-then-c %ld split %ld (of %d)\n", lay, bad[lay][0], bad[lay][1], bad[lay][2], bad[lay][3], bad[lay][4], bad[lay][5], 200 * 256);
    return 0;
}
