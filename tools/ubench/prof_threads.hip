// Stand-alone check for the rocprofv3 crashes seen in round 3 "inside a kernel launch issued from a worker thread"
// (profiles/run_profile.sh): N host threads, each with a stream of its own, launch small kernels with dynamic LDS above
// 48 KiB (hipFuncSetAttribute on first use, as the library's launchers do) and poll a pinned word - the shape of
// ccal_solve_batch's per-context workers, without the library.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/prof_threads.hip -o tools/ubench/prof_threads.bin -lpthread
//   ./prof_threads.bin [threads] [launches]                        -> "OK <threads> <launches>"
//   rocprofv3 --kernel-trace --pmc SQ_WAVES -d out -- ./prof_threads.bin 4 2000
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void k_small(double* out, volatile unsigned long long* word, int seq) {
    extern __shared__ double sm[];
    sm[threadIdx.x] = threadIdx.x * 0.5 + seq;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < 64; ++i) t += sm[(threadIdx.x + i) & 63];
    out[threadIdx.x] = t;
    if (threadIdx.x == 0) { __threadfence_system(); *word = (unsigned long long)seq; }
}

int main(int argc, char** argv) {
    const int nt = argc > 1 ? std::atoi(argv[1]) : 4, nl = argc > 2 ? std::atoi(argv[2]) : 2000;
    std::atomic<int> bad{0};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
        hipStream_t st; double* d = nullptr; unsigned long long* w = nullptr;
        if (hipSetDevice(0) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
            hipMalloc((void**)&d, 64 * sizeof(double)) != hipSuccess ||
            hipHostMalloc((void**)&w, 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { bad++; return; }
        *w = 0;
        const size_t lds = (size_t)(52 + 8 * (t & 3)) * 1024;          // different sizes per thread: concurrent attribute updates
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_small), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { bad++; return; }
        for (int i = 1; i <= nl; ++i) {
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), lds, st, d, w, i);
            if (hipGetLastError() != hipSuccess) { bad++; break; }
            if ((i & 7) == 0) { long spins = 0; while (*(volatile unsigned long long*)w < (unsigned long long)i && ++spins < 2000000000L) { } }
        }
        if (hipStreamSynchronize(st) != hipSuccess) bad++;
        hipFree(d); hipHostFree(w); hipStreamDestroy(st);
    });
    for (auto& x : th) x.join();
    if (bad.load()) { std::printf("FAILED %d\n", bad.load()); return 1; }
    std::printf("OK %d %d\n", nt, nl);
    return 0;
}
