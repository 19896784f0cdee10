// What ONE all-reduce of the optimizer step costs on this node, both ways the library can do it:
//   (a) RCCL: ncclAllReduce(sum, f64, COUNT doubles, in place) on n communicators of ONE process (ncclCommInitAll), one stream per
//       device, issued in a group - what ccal_rccl.hip issues per step;
//   (b) the in-kernel peer sum of the in-process transport: a single-wavefront kernel on every device adds all ranks' buffers in
//       rank order through system-scope loads (peer access over xGMI) - what k_head / k_solve do in front of their decision -
//       behind one event record + n - 1 stream waits per rank.
// Reported: microseconds per collective, averaged over REPS back-to-back collectives (the optimizer's steps are dependent: a
// collective's latency is what a step pays).  n = every visible GPU, or argv[1]; one GPU: n = 1 (RCCL) and two buffers on the one
// device (peer sum).  RCCL is resolved with dlopen like the library does (CCAL_RCCL_LIB, else librccl.so.1).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/allreduce_latency.hip -o tools/ubench/allreduce_latency.bin -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

constexpr int COUNT = 100;           // doubles of the step's buffer (one EUCM camera)
constexpr int MAXR = 16;
struct Peers { const double* src[MAXR]; int n; };

__global__ __launch_bounds__(64) void k_peer_sum(Peers pv, double* out) {
    for (int e = threadIdx.x; e < COUNT; e += 64) {
        double v = __hip_atomic_load(pv.src[0] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int q = 1; q < pv.n; ++q) v += __hip_atomic_load(pv.src[q] + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[e] = v;
    }
}
__global__ __launch_bounds__(64) void k_fill(double* buf, double v) { for (int e = threadIdx.x; e < COUNT; e += 64) buf[e] = v + e; }

typedef int (*init_all_fn)(void**, int, const int*);
typedef int (*allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*group_fn)();
typedef int (*destroy_fn)(void*);

int main(int argc, char** argv) {
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    int n = argc > 1 ? std::atoi(argv[1]) : ndev;
    if (n < 1 || n > ndev || n > MAXR) { std::printf("asked for %d GPU(s), %d visible: refusing\n", n, ndev); return 4; }
    const int reps = 2000;
    std::vector<hipStream_t> st(n);
    std::vector<double*> buf(n), out(n);
    for (int r = 0; r < n; ++r) {
        CK(hipSetDevice(r)); CK(hipStreamCreateWithFlags(&st[r], hipStreamNonBlocking));
        CK(hipMalloc(&buf[r], 2 * COUNT * sizeof(double))); CK(hipMalloc(&out[r], COUNT * sizeof(double)));
        hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, st[r], buf[r], (double)r);
        CK(hipStreamSynchronize(st[r]));
    }
    // ---- (a) RCCL ---------------------------------------------------------------------------------------------------------
    const char* libname = std::getenv("CCAL_RCCL_LIB");
    void* h = dlopen(libname ? libname : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) std::printf("rccl: not resolvable (%s)\n", dlerror());
    else {
        auto init_all = (init_all_fn)dlsym(h, "ncclCommInitAll");
        auto allreduce = (allreduce_fn)dlsym(h, "ncclAllReduce");
        auto gstart = (group_fn)dlsym(h, "ncclGroupStart"); auto gend = (group_fn)dlsym(h, "ncclGroupEnd");
        auto destroy = (destroy_fn)dlsym(h, "ncclCommDestroy");
        std::vector<void*> comm(n, nullptr);
        std::vector<int> devs(n);
        for (int r = 0; r < n; ++r) devs[r] = r;
        if (!init_all || !allreduce || !gstart || !gend || init_all(comm.data(), n, devs.data()) != 0) std::printf("rccl: ncclCommInitAll(%d) failed\n", n);
        else {
            auto round = [&](int k) {
                for (int i = 0; i < k; ++i) {
                    gstart();
                    for (int r = 0; r < n; ++r) allreduce(buf[r], buf[r], COUNT, 8 /* ncclFloat64 */, 0 /* ncclSum */, comm[r], st[r]);
                    gend();
                }
                for (int r = 0; r < n; ++r) { CK(hipSetDevice(r)); CK(hipStreamSynchronize(st[r])); }
            };
            round(200);
            const auto t0 = std::chrono::steady_clock::now();
            round(reps);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            std::printf("rccl      ncclAllReduce of %d doubles over %d rank(s), one process: %.2f us per collective (back to back, host enqueue included)\n", COUNT, n, us);
            for (int r = 0; r < n; ++r) if (destroy) destroy(comm[r]);
        }
    }
    // ---- (b) in-kernel peer sum --------------------------------------------------------------------------------------------
    const int nr = n > 1 ? n : 2;                     // one GPU: two shards of the one device
    bool peer_ok = true;
    for (int i = 0; i < n && peer_ok; ++i)
        for (int k = 0; k < n; ++k) {
            if (i == k) continue;
            int can = 0;
            CK(hipDeviceCanAccessPeer(&can, i, k));
            if (!can) { std::printf("peer sum: no peer access %d -> %d\n", i, k); peer_ok = false; break; }
            CK(hipSetDevice(i));
            const hipError_t e = hipDeviceEnablePeerAccess(k, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { std::printf("peer sum: hipDeviceEnablePeerAccess: %s\n", hipGetErrorString(e)); peer_ok = false; break; }
            (void)hipGetLastError();
        }
    if (peer_ok) {
        std::vector<hipEvent_t> ev(nr);
        std::vector<int> dev_of(nr);
        std::vector<hipStream_t> str(nr);
        std::vector<double*> src(nr), dst(nr);
        for (int r = 0; r < nr; ++r) {
            dev_of[r] = n > 1 ? r : 0;
            CK(hipSetDevice(dev_of[r]));
            CK(hipEventCreateWithFlags(&ev[r], hipEventDisableTiming));
            if (n > 1) { str[r] = st[r]; src[r] = buf[r]; dst[r] = out[r]; }
            else { CK(hipStreamCreateWithFlags(&str[r], hipStreamNonBlocking)); src[r] = buf[0] + r * COUNT; CK(hipMalloc(&dst[r], COUNT * sizeof(double))); }
        }
        Peers pv; pv.n = nr;
        for (int r = 0; r < nr; ++r) pv.src[r] = src[r];
        for (int r = nr; r < MAXR; ++r) pv.src[r] = nullptr;
        auto round = [&](int k) {
            for (int i = 0; i < k; ++i) {
                // every rank: (its reduce kernel - here a fill - has left the sums) record, wait for the peers, sum in the deciding kernel
                for (int r = 0; r < nr; ++r) { CK(hipSetDevice(dev_of[r])); hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, str[r], src[r], (double)i); CK(hipEventRecord(ev[r], str[r])); }
                for (int r = 0; r < nr; ++r) {
                    CK(hipSetDevice(dev_of[r]));
                    for (int q = 0; q < nr; ++q) if (q != r) CK(hipStreamWaitEvent(str[r], ev[q], 0));
                    hipLaunchKernelGGL(k_peer_sum, dim3(1), dim3(64), 0, str[r], pv, dst[r]);
                }
            }
            for (int r = 0; r < nr; ++r) { CK(hipSetDevice(dev_of[r])); CK(hipStreamSynchronize(str[r])); }
        };
        auto base = [&](int k) {                        // the same two launches per rank without the exchange: what is not the collective
            for (int i = 0; i < k; ++i)
                for (int r = 0; r < nr; ++r) {
                    CK(hipSetDevice(dev_of[r]));
                    hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, str[r], src[r], (double)i);
                    Peers one; one.n = 1; one.src[0] = src[r];
                    hipLaunchKernelGGL(k_peer_sum, dim3(1), dim3(64), 0, str[r], one, dst[r]);
                }
            for (int r = 0; r < nr; ++r) { CK(hipSetDevice(dev_of[r])); CK(hipStreamSynchronize(str[r])); }
        };
        round(200); base(200);
        auto t0 = std::chrono::steady_clock::now();
        round(reps);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        t0 = std::chrono::steady_clock::now();
        base(reps);
        const double us0 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("peer sum  %d rank(s) on %d device(s): %.2f us per step with the exchange (event record + %d waits + in-kernel sum of %d buffers), "
                    "%.2f us for the same two launches per rank without it -> %.2f us per collective\n", nr, n, us, nr - 1, nr, us0, us - us0);
    }
    return 0;
}
