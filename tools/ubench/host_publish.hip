// Two questions behind the session-size work of round 4:
//  1. What does it cost a kernel to write a 30 KB result into coherent pinned host memory - one wavefront with 8-byte stores (what
//     k_head did), 16-byte stores, four wavefronts, several workgroups?  (in-kernel wall_clock64 stamps, 100 MHz)
//  2. What does the "last arriver sums its cluster's rows" pattern cost inside a 625-workgroup kernel - an agent-scope release
//     (L2 write-back on a chip of 8 XCDs), an atomic, and for one wavefront in 16 a 16-row sum - against the same kernel without it?
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/host_publish.hip -o tools/ubench/host_publish.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

typedef double dv2 __attribute__((ext_vector_type(2)));

template <int VEC>
__global__ void k_pub(const double* src, double* dst, int n, long long* stamps) {
    const long long t0 = wall_clock64();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (VEC == 1) for (int e = tid; e < n; e += nth) dst[e] = src[e];
    else for (int e = tid; e < n / 2; e += nth) reinterpret_cast<dv2*>(dst)[e] = reinterpret_cast<const dv2*>(src)[e];
    __threadfence_system();
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = t1; }
}

constexpr int ROW = 96, REC = 192;
template <int MODE>          // 0: rows only; 1: + cluster sums by the last arriver; 2: the same with write-through (sc1) row stores and no L2 write-back
__global__ __launch_bounds__(64) void k_cluster(double* rows, double* recs, double* crow, unsigned* cnt, int n_wg, int csize, int seq) {
    const int lane = threadIdx.x, b = blockIdx.x;
    for (int e = lane; e < REC; e += 64) recs[(size_t)b * REC + e] = b * 0.25 + e + seq;
    for (int e = lane; e < ROW; e += 64) {
        const double v = (double)((b * 7 + e * 3 + seq) % 101) - 50.0;
        if (MODE == 2) __hip_atomic_store(&rows[(size_t)b * ROW + e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else rows[(size_t)b * ROW + e] = v;
    }
    if (MODE == 0) return;
    const int c = b / csize, first = c * csize, members = min(csize, n_wg - first);
    __shared__ unsigned old;
    if (MODE == 1) {
        if (lane == 0) old = __hip_atomic_fetch_add(&cnt[c], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);     // release: this wavefront's rows; acquire: the others'
    } else {
        // the rows went out as agent-scope stores (written through to where every XCD sees them): complete when vmcnt says so -
        // no write-back of the L2's other dirty lines (the records: nobody reads them in this kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) old = __hip_atomic_fetch_add(&cnt[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (old != (unsigned)(members - 1)) return;
    if (MODE == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);        // (the workgroup's other lanes: order their loads behind lane 0's atomic)
    for (int e = lane; e < ROW; e += 64) {
        double t = 0.0;
        for (int r = 0; r < members; ++r) t += __hip_atomic_load(&rows[(size_t)(first + r) * ROW + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        crow[(size_t)c * ROW + e] = t;
    }
    if (lane == 0) cnt[c] = 0;
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n = 3750 + 18;
    double *d_src, *h_coh, *h_non; long long* stamps;
    CK(hipMalloc((void**)&d_src, n * 8)); CK(hipMemset(d_src, 0, n * 8));
    CK(hipHostMalloc((void**)&h_coh, n * 8, hipHostMallocCoherent | hipHostMallocMapped));
    CK(hipHostMalloc((void**)&h_non, n * 8, hipHostMallocNonCoherent | hipHostMallocMapped));
    CK(hipHostMalloc((void**)&stamps, 2 * 64 * sizeof(long long), hipHostMallocCoherent | hipHostMallocMapped));
    struct V { const char* name; int vec, wgs, threads; bool coherent; };
    const V vs[] = { {"1 wavefront, 8-byte stores, coherent", 1, 1, 64, true}, {"1 wavefront, 16-byte stores, coherent", 2, 1, 64, true},
                     {"4 wavefronts, 8-byte stores, coherent", 1, 1, 256, true},
                     {"4 wavefronts, 16-byte stores, coherent", 2, 1, 256, true}, {"16 wavefronts (1 workgroup of 1024), 16-byte", 2, 1, 1024, true},
                     {"16 workgroups x 256, 16-byte stores, coherent", 2, 16, 256, true},
                     {"1 wavefront, 8-byte stores, non-coherent", 1, 1, 64, false}, {"4 wavefronts, 16-byte stores, non-coherent", 2, 1, 256, false} };
    for (const V& v : vs) {
        double best = 1e30, sum = 0.0; const int reps = 50;
        for (int r = 0; r < reps + 5; ++r) {
            double* dst = v.coherent ? h_coh : h_non;
            if (v.vec == 1) hipLaunchKernelGGL(k_pub<1>, dim3(v.wgs), dim3(v.threads), 0, st, d_src, dst, n, stamps);
            else hipLaunchKernelGGL(k_pub<2>, dim3(v.wgs), dim3(v.threads), 0, st, d_src, dst, n, stamps);
            CK(hipStreamSynchronize(st));
            long long t0 = stamps[0], t1 = stamps[1];
            for (int b = 1; b < v.wgs; ++b) { if (stamps[2 * b] < t0) t0 = stamps[2 * b]; if (stamps[2 * b + 1] > t1) t1 = stamps[2 * b + 1]; }
            const double us = (t1 - t0) / 100.0;
            if (r >= 5) { sum += us; if (us < best) best = us; }
        }
        std::printf("publish 30 KB: %-52s  avg %6.2f us  best %6.2f us\n", v.name, sum / reps, best);
    }
    // ---- 2. cluster sums
    const int n_wg = 625, csize = 16, n_c = (n_wg + csize - 1) / csize;
    double *rows, *recs, *crow; unsigned* cnt;
    CK(hipMalloc((void**)&rows, (size_t)n_wg * ROW * 8)); CK(hipMalloc((void**)&recs, (size_t)n_wg * REC * 8));
    CK(hipMalloc((void**)&crow, (size_t)n_c * ROW * 8)); CK(hipMalloc((void**)&cnt, n_c * 4)); CK(hipMemset(cnt, 0, n_c * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        const int reps = 500; float best = 1e30f;
        for (int trial = 0; trial < 5; ++trial) {
            for (int r = 0; r < 20; ++r) { if (mode == 2) hipLaunchKernelGGL(k_cluster<2>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, r); else if (mode) hipLaunchKernelGGL(k_cluster<1>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, r); else hipLaunchKernelGGL(k_cluster<0>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, r); }
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < reps; ++r) { if (mode == 2) hipLaunchKernelGGL(k_cluster<2>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, r); else if (mode) hipLaunchKernelGGL(k_cluster<1>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, r); else hipLaunchKernelGGL(k_cluster<0>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, r); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        std::printf("cluster kernel, 625 workgroups, mode %d (%s): %.2f us per back-to-back launch\n", mode, mode == 2 ? "write-through rows + last-arriver cluster sums, no L2 write-back" : mode ? "rows + last-arriver cluster sums" : "rows only", best * 1e3 / reps);
    }
    // correctness of the cluster sums across XCDs: 200 launches, every cluster row checked on the host
    std::vector<double> h((size_t)n_c * ROW); int bad = 0;
    for (int seq = 0; seq < 2000; ++seq) {
        hipLaunchKernelGGL(k_cluster<2>, dim3(n_wg), dim3(64), 0, st, rows, recs, crow, cnt, n_wg, csize, seq);
        CK(hipMemcpyAsync(h.data(), crow, h.size() * 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        for (int c = 0; c < n_c; ++c) for (int e = 0; e < ROW; ++e) {
            double t = 0.0; for (int b = c * csize; b < std::min(n_wg, (c + 1) * csize); ++b) t += (double)((b * 7 + e * 3 + seq) % 101) - 50.0;
            if (t != h[(size_t)c * ROW + e]) ++bad;
        }
    }
    std::printf("cluster sums (write-through form) checked over 2000 launches: %d wrong entries\n", bad);
    return bad ? 1 : 0;
}
