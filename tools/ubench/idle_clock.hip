// Does a one-wavefront kernel on an otherwise idle chip run at a lower clock than the same kernel beside a busy one?
// One wavefront walks a chain of N dependent v_fma_f64 (known issue cost: ~8.4 cycles each, tools/ubench/dp_issue.hip) and times
// itself with wall_clock64 (constant 100 MHz); (a) alone, launched back to back like the groups of a session-sized solve, (b) while
// a second stream keeps every CU busy with FMA loops.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/idle_clock.hip -o tools/ubench/idle_clock.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

__global__ void k_chain(double* out, long long* ticks, int n) {
    double x = out[0] + 1.0, a = 0.999999, b = 1e-9;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; i += 8) {
        x = __builtin_fma(x, a, b); x = __builtin_fma(x, a, b); x = __builtin_fma(x, a, b); x = __builtin_fma(x, a, b);
        x = __builtin_fma(x, a, b); x = __builtin_fma(x, a, b); x = __builtin_fma(x, a, b); x = __builtin_fma(x, a, b);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[1] = x; ticks[0] = t1 - t0; }
}
__global__ void k_busy(double* out, int iters) {
    double x = threadIdx.x * 1e-3, y = 1.0 + blockIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) { x = __builtin_fma(x, 0.999, y); y = __builtin_fma(y, 0.999, x); }
    if (x == 123.456) out[0] = x + y;
}
int main() {
    CK(hipSetDevice(0));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    double* d; long long* t; CK(hipMalloc((void**)&d, 64)); CK(hipMemset(d, 0, 64));
    CK(hipHostMalloc((void**)&t, 64, hipHostMallocCoherent | hipHostMallocMapped));
    const int n = 4000;
    for (int mode = 0; mode < 3; ++mode) {
        double best = 1e30, sum = 0; const int reps = 200;
        if (mode == 1) hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, s2, d + 4, 40000000);      // seconds of chip-wide FMA work
        if (mode == 1) { hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, d, t, n); CK(hipStreamSynchronize(s1)); }
        for (int r = 0; r < reps; ++r) {
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s1, d, t, n);
            CK(hipStreamSynchronize(s1));
            const double us = t[0] / 100.0;
            sum += us; if (us < best) best = us;
        }
        std::printf("%-46s chain of %d dependent v_fma_f64: avg %.2f us, best %.2f us  ->  %.0f MHz at 8.4 cycles each (avg)\n",
                    mode == 0 ? "alone, one launch at a time:" : mode == 1 ? "beside a kernel that keeps every CU busy:" : "alone again:", n, sum / reps, best, n * 8.4 / (sum / reps));
        if (mode == 1) { CK(hipStreamSynchronize(s2)); }
    }
    return 0;
}
