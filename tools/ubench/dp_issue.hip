// Micro-benchmark (developer tool): the FP64 issue / latency cost model of one gfx950 SIMD, the numbers the mode-N Gram
// kernels are designed against.  Per variant: shader cycles (s_memtime) per instruction for one wavefront
// per SIMD and for two, and the clock the chip holds meanwhile (s_memtime / s_memrealtime).
//   fma_indep   64 independent v_fma_f64 chains                      -> issue cost of a DP FMA
//   fma_dep     one chain                                            -> dependent latency
//   fma_depN    N interleaved chains (N = 2, 4, 8)                   -> how much ILP hides it
//   rcp / rsq   v_rcp_f64 / v_rsq_f64 independent
//   swap32      v_permlane32_swap_b32 (the u-row / v-row exchange of k_gram2)
//   dpp         v_mov_b32_dpp quad_perm
//   accvgpr     v_accvgpr_write + read pairs
//   ldsadd      ds_add_f64 (no return), lane-private, conflict-free
//   fma+swap    8 FMAs per swap, fma+ldsadd: 8 FMAs per ds_add_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NV = 32;
template <int MODE>
__global__ __launch_bounds__(512) void k(double* out, long long* stamps, int iters, double seed) {
    __shared__ double lds[512 * 4];
    double a[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
    double x = seed * 1.0000001, y = seed * 1e-9;
    lds[threadIdx.x] = 0.0;
    int u0 = threadIdx.x, u1 = threadIdx.x * 3;
    const unsigned ldsa = (unsigned)(threadIdx.x * 8);
    __syncthreads();
    const long long r0 = wall_clock64();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 2 || MODE == 3 || MODE == 4) {
            constexpr int N = MODE == 2 ? 2 : (MODE == 3 ? 4 : 8);
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i % N]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
        } else if constexpr (MODE == 6) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
        } else if constexpr (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u0), "+v"(u1));
        } else if constexpr (MODE == 8) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u0) : "v"(u1));
        } else if constexpr (MODE == 9) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_read_b32 %0, a1" : "+v"(u0) : : "a0", "a1");
        } else if constexpr (MODE == 10) {
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("ds_add_f64 %0, %1" : : "v"(ldsa), "v"(a[i % NV]) : "memory");
        } else if constexpr (MODE == 11) {       // 8 FMAs per swap
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i % NV]) : "v"(x), "v"(y));
                if (i % 8 == 7) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u0), "+v"(u1));
            }
        } else if constexpr (MODE == 12) {       // 8 FMAs per ds_add_f64
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) {
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i % NV]) : "v"(x), "v"(y));
                if (i % 8 == 7) asm volatile("ds_add_f64 %0, %1" : : "v"(ldsa), "v"(x) : "memory");
            }
        } else if constexpr (MODE == 13) {       // v_mul_f64 independent
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i % NV]) : "v"(x));
        } else if constexpr (MODE == 14) {       // v_fmac_f64 e32 (two-address form, what the compiler emits for accumulators)
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i % NV]) : "v"(x), "v"(y));
        } else if constexpr (MODE == 15) {       // FMA with three distinct register sources drawn from the accumulators' neighbours
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i % NV]) : "v"(a[(i + 7) % NV]), "v"(a[(i + 13) % NV]));
        } else if constexpr (MODE == 16) {       // v_cvt_f64_f32
            float f = (float)seed;
#pragma unroll
            for (int i = 0; i < 2 * NV; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i % NV]) : "v"(f));
        }
    }
    const long long t1 = clock64();
    const long long r1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + u0 + u1 + lds[threadIdx.x];
    if ((threadIdx.x & 63) == 0) {
        const int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE> int run(const char* name, double* out, long long* st, int insts_per_iter) {
    const int iters = 4000;
    for (int threads : {256, 512}) {
        const int waves = 256 * threads / 64;
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, st, iters, 1.0);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, st, iters, 1.0);
        hipEventRecord(e1); CHECK(hipEventSynchronize(e1));
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * waves);
        CHECK(hipMemcpy(h.data(), st, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost));
        std::vector<double> cyc(waves), ghz(waves);
        for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double per = cyc[waves / 2] / ((double)iters * insts_per_iter);
        printf("%-10s %d waves/SIMD: %7.2f cycles/inst per wave (per SIMD: %6.2f)  clock %.2f GHz  kernel %.3f ms\n", name, threads / 256,
               per, per / (threads / 256), ghz[waves / 2], ms);
    }
    return 0;
}
int main() {
    double* out; long long* st;
    CHECK(hipMalloc(&out, 256 * 512 * sizeof(double)));
    CHECK(hipMalloc(&st, 2 * 256 * 8 * sizeof(long long)));
    run<0>("fma_indep", out, st, 64);
    run<14>("fmac_e32", out, st, 64);
    run<15>("fma_3src", out, st, 64);
    run<13>("mul_indep", out, st, 64);
    run<1>("fma_dep", out, st, 64);
    run<2>("fma_dep2", out, st, 64);
    run<3>("fma_dep4", out, st, 64);
    run<4>("fma_dep8", out, st, 64);
    run<5>("rcp", out, st, 64);
    run<6>("rsq", out, st, 64);
    run<16>("cvt_f64_f32", out, st, 64);
    run<7>("swap32", out, st, 64);
    run<8>("dpp", out, st, 64);
    run<9>("accvgpr", out, st, 64);
    run<10>("ldsadd", out, st, 64);
    run<11>("fma+swap", out, st, 72);
    run<12>("fma+ldsadd", out, st, 72);
    return 0;
}
