// Micro-benchmark: do v_mfma_f64_16x16x4_f64 and f64 VALU FMAs overlap on gfx950?
// Decides whether the Gram accumulation of mode N belongs on the matrix cores or on the VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>   // 0 mfma only, 1 valu only, 2 both interleaved in one wave, 3 waves 0-3 mfma / 4-7 valu
__global__ __launch_bounds__(512) void k(double* out, int iters, double seed) {
    const int wave = threadIdx.x >> 6;
    d4 a0 = {0,0,0,0}, a1 = {0,0,0,0}, a2 = {0,0,0,0}, a3 = {0,0,0,0};
    double v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    const double x = seed * 1.0000001, y = seed * 0.9999999;
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && wave < 4);
    const bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], x, y);
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], y, x);
        }
        if (do_mfma) {
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    d4 t = a0 + a1 + a2 + a3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + t[0] + t[1] + t[2] + t[3];
}

template <int MODE> float run(double* out, int blocks, int threads, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    double* out; CHECK(hipMalloc(&out, 256 * 8 * 512 * sizeof(double)));
    const int iters = 20000;
    // one wave per SIMD (256 threads/block, 1 block per CU) unless noted
    float m = run<0>(out, 256, 256, iters), v = run<1>(out, 256, 256, iters), b = run<2>(out, 256, 256, iters);
    float w = run<3>(out, 256, 512, iters);
    float m2 = run<0>(out, 256, 512, iters), v2 = run<1>(out, 256, 512, iters);
    // per iteration: 4 MFMA (4 x 64 cyc if 64-cycle issue) and 32 FMA wave-instructions
    printf("1 wave/SIMD : mfma-only %.3f ms  valu-only %.3f ms  both-in-one-wave %.3f ms\n", m, v, b);
    printf("2 waves/SIMD: mfma|valu split %.3f ms   mfma-only(2 waves) %.3f ms   valu-only(2 waves) %.3f ms\n", w, m2, v2);
    printf("cycles/iter @2.4GHz: mfma %.1f valu %.1f both %.1f split %.1f\n", m * 2.4e6 / iters, v * 2.4e6 / iters, b * 2.4e6 / iters, w * 2.4e6 / iters);
    return 0;
}
