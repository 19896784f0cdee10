// Micro-benchmark (developer tool): the streaming ceilings of one MI355X as a trivial kernel sees them - what mode E
// (k_eval: 52 MB of corner rows in, 276-467 MB of residual + Jacobian rows out per launch) is priced against besides the
// 8 TB/s data-sheet figure.  Every variant moves the same bytes with no arithmetic at all:
//   fill      write-only, one 16-KiB tile per workgroup pass (k_eval's shape: 256 lanes x 8 B x 8 stores, full 64-B lines per
//             quarter-wavefront), plain / non-temporal stores, 8 or 16 bytes per lane and store
//   memset    hipMemsetAsync over the same buffer (the runtime's own fill kernel)
//   read      read-only (sum kept in a register, one store per workgroup)
//   copy      read N bytes, write N bytes
//   evalmix   read 1 byte per 6.3 written (mode E's ratio at D = 12)
// Sizes: 276 MB (EUCM J block of 10 000 x 144 corners), 467 MB (18-column block), and two 346-MB buffers alternating (the
// two-camera rig's 691 MB of outputs - beyond the 256-MiB Infinity Cache).
//   hipcc -O3 --offload-arch=gfx950 -o hbm_stream.bin hbm_stream.hip && ./hbm_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

// grid-stride over 16-KiB tiles (2048 doubles); TILES_PER_WG = 0: one tile per workgroup (k_eval's launch shape)
template <bool NT, int W>
__global__ __launch_bounds__(256) void k_fill(double* __restrict__ out, size_t n_tiles, double v) {
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        double* o = out + t * 2048;
        if constexpr (W == 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (NT) __builtin_nontemporal_store(v, o + i * 256 + threadIdx.x);
                else o[i * 256 + threadIdx.x] = v;
            }
        } else {
            d2 vv = { v, v };
            d2* o2 = (d2*)o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (NT) __builtin_nontemporal_store(vv, o2 + i * 256 + threadIdx.x);
                else o2[i * 256 + threadIdx.x] = vv;
            }
        }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ in, size_t n_tiles, double* sink) {
    double s = 0;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const double* p = in + t * 2048;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += NT ? __builtin_nontemporal_load(p + i * 256 + threadIdx.x) : p[i * 256 + threadIdx.x];
    }
    if (s == 123.456) sink[0] = s;
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const double* __restrict__ in, double* __restrict__ out, size_t n_tiles) {
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const double* p = in + t * 2048; double* o = out + t * 2048;
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = p[i * 256 + threadIdx.x];
#pragma unroll
        for (int i = 0; i < 8; ++i) { if (NT) __builtin_nontemporal_store(x[i], o + i * 256 + threadIdx.x); else o[i * 256 + threadIdx.x] = x[i]; }
    }
}
// mode E's ratio: per 13 doubles written (12 J columns + r, two rows of a corner = 26 doubles) a lane reads 9 floats = 36 B;
// here: per 16-KiB tile written, 2.6 KiB (1/6.3) read as floats
__global__ __launch_bounds__(256) void k_evalmix(const float* __restrict__ in, double* __restrict__ out, size_t n_tiles) {
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const float* p = in + t * 650; double* o = out + t * 2048;
        float a = p[threadIdx.x], b = p[256 + threadIdx.x], c = threadIdx.x < 138 ? p[512 + threadIdx.x] : 0.f;
        const double v = (double)a + (double)b + (double)c;
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(v + i, o + i * 256 + threadIdx.x);
    }
}

struct Timer {
    hipEvent_t e0, e1; hipStream_t s;
    template <class F> double run(F&& f, int reps = 20, int warm = 5) {      // seconds per launch
        for (int i = 0; i < warm; ++i) f(i);
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) f(i);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e-3 / reps;
    }
};

int main() {
    hipStream_t s; CHECK(hipStreamCreate(&s));
    Timer T; T.s = s; CHECK(hipEventCreate(&T.e0)); CHECK(hipEventCreate(&T.e1));
    const size_t cap = 480ull << 20;
    double *A, *B, *sink; float* F;
    CHECK(hipMalloc(&A, cap)); CHECK(hipMalloc(&B, cap)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&F, 128ull << 20));
    CHECK(hipMemsetAsync(A, 0, cap, s)); CHECK(hipMemsetAsync(B, 0, cap, s)); CHECK(hipMemsetAsync(F, 0, 128ull << 20, s));
    // clock ramp
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k_fill<true, 8>), dim3(16875), dim3(256), 0, s, A, (size_t)16875, 1.0);
    CHECK(hipStreamSynchronize(s));
    const size_t sizes[] = { 276480000ull, 466560000ull };
    for (size_t bytes : sizes) {
        const size_t nt = bytes / 16384;
        const double gb = nt * 16384 * 1e-9;
        printf("== %zu MB (%zu tiles of 16 KiB)\n", bytes / 1000000, nt);
        for (int grid_mode = 0; grid_mode < 3; ++grid_mode) {
            const unsigned g = grid_mode == 0 ? (unsigned)nt : (grid_mode == 1 ? 2048u : 8192u);
            const char* gn = grid_mode == 0 ? "one tile/wg" : (grid_mode == 1 ? "grid 2048" : "grid 8192");
            double t;
            t = T.run([&](int) { hipLaunchKernelGGL((k_fill<false, 8>), dim3(g), dim3(256), 0, s, A, nt, 1.0); });
            printf("  fill plain  8B  %-12s %7.1f us  %6.0f GB/s\n", gn, t * 1e6, gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL((k_fill<true, 8>), dim3(g), dim3(256), 0, s, A, nt, 1.0); });
            printf("  fill nt     8B  %-12s %7.1f us  %6.0f GB/s\n", gn, t * 1e6, gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL((k_fill<false, 16>), dim3(g), dim3(256), 0, s, A, nt, 1.0); });
            printf("  fill plain 16B  %-12s %7.1f us  %6.0f GB/s\n", gn, t * 1e6, gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL((k_fill<true, 16>), dim3(g), dim3(256), 0, s, A, nt, 1.0); });
            printf("  fill nt    16B  %-12s %7.1f us  %6.0f GB/s\n", gn, t * 1e6, gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL((k_read<false>), dim3(g), dim3(256), 0, s, A, nt, sink); });
            printf("  read plain      %-12s %7.1f us  %6.0f GB/s\n", gn, t * 1e6, gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL((k_read<true>), dim3(g), dim3(256), 0, s, A, nt, sink); });
            printf("  read nt         %-12s %7.1f us  %6.0f GB/s\n", gn, t * 1e6, gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL((k_copy<true>), dim3(g), dim3(256), 0, s, A, B, nt); });
            printf("  copy nt-store   %-12s %7.1f us  %6.0f GB/s (read + written)\n", gn, t * 1e6, 2 * gb / t);
            t = T.run([&](int) { hipLaunchKernelGGL(k_evalmix, dim3(g), dim3(256), 0, s, F, A, nt); });
            printf("  evalmix nt      %-12s %7.1f us  %6.0f GB/s (written + 1/6.3 read)\n", gn, t * 1e6, gb * (1 + 2600.0 / 16384) / t);
        }
        double t = T.run([&](int) { hipMemsetAsync(A, 0, nt * 16384, s); });
        printf("  hipMemsetAsync                %7.1f us  %6.0f GB/s\n", t * 1e6, gb / t);
    }
    {   // the rig: two 346-MB buffers alternating (691 MB of outputs between revisits)
        const size_t nt = 345600000ull / 16384; const double gb = nt * 16384 * 1e-9;
        double t = T.run([&](int i) { hipLaunchKernelGGL((k_fill<true, 8>), dim3((unsigned)nt), dim3(256), 0, s, (i & 1) ? A : B, nt, 1.0); });
        printf("== 2 x 346 MB alternating, fill nt 8B one tile/wg  %7.1f us  %6.0f GB/s\n", t * 1e6, gb / t);
        t = T.run([&](int i) { hipLaunchKernelGGL((k_fill<true, 8>), dim3((unsigned)nt), dim3(256), 0, s, A, nt, 1.0); });
        printf("== 1 x 346 MB rewritten,   fill nt 8B one tile/wg  %7.1f us  %6.0f GB/s\n", t * 1e6, gb / t);
    }
    CHECK(hipStreamSynchronize(s));
    printf("HBM-STREAM-DONE\n");
    return 0;
}
