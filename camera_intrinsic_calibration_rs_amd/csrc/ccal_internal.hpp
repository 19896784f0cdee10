// Host-side structures shared by the API translation unit and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/ccal.h"
#include "ccal_models.hpp"

namespace ccal {

constexpr int WAVES_PER_BLOCK = 4;     // 256 threads: one wavefront per observation frame
constexpr int PF_STRIDE = 64;          // doubles kept per frame slot for back-substitution

// Arguments of the per-camera kernels (everything device-resident).
struct KArgs {
    const float* x; const float* y; const float* z; const float* u; const float* v;
    const int64_t* obs_off;        // [n_obs+1]
    const int32_t* obs_slot;       // [n_obs]
    const int64_t* joff;           // [n_obs] offset (doubles) of the frame's block Jacobians in J_out
    const int32_t* list;           // observation frames of this camera
    int32_t n_list;
    int32_t cam;
    const double* intr;            // [n_cams][CCAL_PMAX] full params
    const double* poses;           // [n_slots][6]
    const double* extr;            // [n_cams][6]
    double huber_delta;
    ModelRt rt;                    // the context's run-time conventions (KB4 threshold, OPENCV5 coefficient order)
    int32_t apply_loss;
    double* r_out;                 // mode E
    double* J_out;
    double* err_out;               // reprojection errors
};

struct NormalWs;

struct CamLayout {
    int model = 0, P = 0, Peff = 0, D = 0;
    int col_theta = 0, col_extr = -1;
    double width = 0, height = 0;
    std::vector<int32_t> obs;          // host copy of the camera's observation frames
    int32_t* d_obs = nullptr;
};

}  // namespace ccal

struct ccal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    ccal_model_conventions conv;   // run-time model conventions (defaults: ccal_models.hpp)
    // a problem holds its context: ccal_ctx_destroy with problems still alive only marks it, the last
    // ccal_problem_destroy frees it (bindings with a garbage collector destroy in any order)
    int n_problems = 0;
    bool destroy_requested = false;
    // ccal_solve_batch: the host thread that drives this context's stream while the caller's thread drives another one
    // (created on first use, joined when the context is freed; ccal_solver.hip)
    struct ccal_ctx_worker* worker = nullptr;
    // ccal_solve_batch, session sizes: the argument blocks of a batch's problems, one launch per step for all of them
    // (k_gram1v_batch reads the table on the device; h_: its pinned staging) - grown on demand, freed with the context
    char* d_batch_tab = nullptr; char* h_batch_tab = nullptr; size_t batch_tab_bytes = 0;
    // ccal_pin_buffer: the caller's ranges this context registered (unregistered by ccal_unpin_buffer or with the context)
    std::vector<std::pair<void*, size_t>> pinned;
    // Blocks of device and pinned host memory (and side streams) that destroyed problems of this context gave back: a calibration
    // session creates a problem per camera and per retry (src/bin/camera_calibration.rs:205-265) - hipMalloc / hipHostMalloc /
    // hipStreamCreate and their frees were a third of a 600-frame session's create + first solve.  A few blocks, bounded sizes;
    // freed with the context.  `live`: what is handed out (pointer -> bytes, second = 1 for pinned host memory).
    struct Block { void* p; size_t bytes; };
    std::vector<Block> cache_dev, cache_host;
    std::vector<hipStream_t> cache_streams;
    std::vector<std::pair<void*, std::pair<size_t, int>>> live;
};
namespace ccal {
constexpr size_t kCacheDevMaxBytes = (size_t)1 << 30, kCacheHostMaxBytes = (size_t)64 << 20;
constexpr size_t kCacheMaxBlocks = 12;
// a block of at least `bytes` (256-byte granularity): from the context's cache when one fits without wasting more than a quarter
inline hipError_t ctx_alloc(ccal_ctx* ctx, void** out, size_t bytes, bool host, unsigned host_flags = 0) {
    bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    std::vector<ccal_ctx::Block>& cache = host ? ctx->cache_host : ctx->cache_dev;
    size_t best = cache.size();
    for (size_t i = 0; i < cache.size(); ++i)
        if (cache[i].bytes >= bytes && cache[i].bytes <= bytes + bytes / 4 + 4096 && (best == cache.size() || cache[i].bytes < cache[best].bytes)) best = i;
    void* p = nullptr;
    size_t got = bytes;
    if (best != cache.size()) { p = cache[best].p; got = cache[best].bytes; cache.erase(cache.begin() + (ptrdiff_t)best); }
    else {
        const hipError_t e = host ? hipHostMalloc(&p, bytes, host_flags) : hipMalloc(&p, bytes);
        if (e != hipSuccess) return e;
    }
    try { ctx->live.emplace_back(p, std::make_pair(got, host ? 1 : 0)); }
    catch (...) { if (host) (void)hipHostFree(p); else (void)hipFree(p); throw; }
    *out = p;
    return hipSuccess;
}
inline hipError_t ctx_dev_alloc(ccal_ctx* ctx, void** out, size_t bytes) { return ctx_alloc(ctx, out, bytes, false); }
// pinned, mapped, coherent host memory (every pinned block of the library is allocated that way: one kind in the cache)
inline hipError_t ctx_host_alloc(ccal_ctx* ctx, void** out, size_t bytes) { return ctx_alloc(ctx, out, bytes, true, hipHostMallocCoherent | hipHostMallocMapped); }
// give a block back (anything not handed out by ctx_alloc - or with no context left - is simply freed).  The caller has made sure
// nothing in flight still uses it.
inline void ctx_release(ccal_ctx* ctx, void* p, bool host) {
    if (!p) return;
    if (ctx) {
        for (size_t i = 0; i < ctx->live.size(); ++i) {
            if (ctx->live[i].first != p) continue;
            const size_t bytes = ctx->live[i].second.first;
            const bool is_host = ctx->live[i].second.second != 0;
            ctx->live.erase(ctx->live.begin() + (ptrdiff_t)i);
            std::vector<ccal_ctx::Block>& cache = is_host ? ctx->cache_host : ctx->cache_dev;
            size_t total = bytes;
            for (const auto& b : cache) total += b.bytes;
            if (!ctx->destroy_requested && cache.size() < kCacheMaxBlocks && total <= (is_host ? kCacheHostMaxBytes : kCacheDevMaxBytes)) {
                try { cache.push_back({ p, bytes }); return; } catch (...) { }
            }
            if (is_host) (void)hipHostFree(p); else (void)hipFree(p);
            return;
        }
    }
    if (host) (void)hipHostFree(p); else (void)hipFree(p);
}
inline hipError_t ctx_stream_get(ccal_ctx* ctx, hipStream_t* out) {
    if (!ctx->cache_streams.empty()) { *out = ctx->cache_streams.back(); ctx->cache_streams.pop_back(); return hipSuccess; }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
inline void ctx_stream_put(ccal_ctx* ctx, hipStream_t s) {
    if (!s) return;
    if (ctx && !ctx->destroy_requested && ctx->cache_streams.size() < 4) { try { ctx->cache_streams.push_back(s); return; } catch (...) { } }
    (void)hipStreamDestroy(s);
}
inline void ctx_cache_clear(ccal_ctx* ctx) {
    for (auto& b : ctx->cache_dev) (void)hipFree(b.p);
    for (auto& b : ctx->cache_host) (void)hipHostFree(b.p);
    for (hipStream_t s : ctx->cache_streams) (void)hipStreamDestroy(s);
    ctx->cache_dev.clear(); ctx->cache_host.clear(); ctx->cache_streams.clear();
}
}  // namespace ccal
namespace ccal {
// The device-side address of caller memory that is pinned (registered with ccal_pin_buffer / hipHostRegister, or allocated with
// hipHostMalloc) for all of [host, host + bytes); nullptr: ordinary pageable memory - the library stages it.
inline void* pinned_device_ptr(const void* host, size_t bytes) {
    if (!host || !bytes) return nullptr;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) return nullptr;
    hipPointerAttribute_t at2;                     // the range's last byte lies in the same registration
    if (hipPointerGetAttributes(&at2, static_cast<const char*>(host) + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (at2.type != hipMemoryTypeHost || !at2.devicePointer ||
        static_cast<char*>(at2.devicePointer) - static_cast<char*>(at.devicePointer) != (ptrdiff_t)(bytes - 1)) return nullptr;
    return at.devicePointer;
}
}  // namespace ccal
namespace ccal {
void ctx_worker_destroy(ccal_ctx* ctx);
// the context's conventions as the kernels take them
inline ModelRt model_rt(const ccal_ctx* ctx) {
    ModelRt rt = {};
    rt.kb4_eps = ctx->conv.kb4_small_radius; rt.unproject_eps = ctx->conv.unproject_small_radius;
    rt.ocv5_perm = 0;
    for (int i = 0; i < 5; ++i) rt.ocv5_perm |= (uint32_t)(ctx->conv.ocv5_order[i] & 7) << (3 * i);
    return rt;
}
inline bool ocv5_identity(const ccal_ctx* ctx) { return model_rt(ctx).ocv5_perm == kOcv5IdentityPerm; }
}  // namespace ccal

namespace ccal {
// Ragged single-camera problems: the bins of a Gram launch (gram2_bin_plan, ccal_kernels_gram2.hip).  Bin b: bin_count[b] frames - positions
// bin_first[b] .. of the problem's table of frames sorted by corner count - with bin_lpf[b] lanes per frame, worked on by the
// workgroups bin_wg0[b] .. bin_wg0[b + 1] - 1 of ONE launch.
constexpr int kGramMaxBins = 5;
struct GramBins {
    int32_t n_bins = 0;               // 0: no bins (every frame the same lanes per frame, frames in table order)
    int32_t lpf[kGramMaxBins] = {}, first[kGramMaxBins] = {}, count[kGramMaxBins] = {}, wg0[kGramMaxBins + 1] = {};
    int32_t fold = 0;                 // > 0: ONE bin, the table folded at this position (frames behind it smallest first)
};
// The plan for corner counts n[0 .. n_obs) (host): which frames go together and with how many lanes each; order = the sorted table
// (frame indices, bins in launch order).  n_bins == 0: binning does not pay (uniform frames, too few of them).
GramBins gram2_bin_plan(const int64_t* obs_off, int n_obs, bool two_per_simd, std::vector<int32_t>* order, bool rig_list = false);

}  // namespace ccal

struct ccal_problem {
    ccal_ctx* ctx = nullptr;
    int n_cams = 0, n_slots = 0, n_obs = 0, K = 0;
    bool one_focal = false;
    double huber_delta = 1.0;
    int64_t n_corners = 0, j_len = 0;
    std::vector<ccal::CamLayout> cams;
    std::vector<int64_t> h_obs_off, h_joff;
    std::vector<int32_t> h_obs_cam, h_obs_slot;
    bool slot_ident = false;       // h_obs_slot[o] == o for every observation frame
    // single camera, ragged frames: the bins of the Gram launch and the sorted table int4 { frame, first corner, corners, slot } per
    // position (a slice of d_block); gram_bins.n_bins == 0: none
    ccal::GramBins gram_bins;
    int32_t* d_bin_tab = nullptr;
    // device-resident inputs
    char* d_scratch = nullptr; size_t scratch_bytes = 0;      // validation()'s temporaries (grown on demand, kept between calls)
    char* d_block = nullptr;       // ONE device allocation: the corner arrays, the frame tables and the six parameter arrays are slices of it
    float *d_x = nullptr, *d_y = nullptr, *d_z = nullptr, *d_u = nullptr, *d_v = nullptr;
    int64_t *d_obs_off = nullptr, *d_joff = nullptr;
    int32_t *d_obs_cam = nullptr, *d_obs_slot = nullptr;
    // device-resident parameters (current point and candidate)
    double *d_intr = nullptr, *d_poses = nullptr, *d_extr = nullptr;
    double *d_intr_c = nullptr, *d_poses_c = nullptr, *d_extr_c = nullptr;
    // constraints in eff index space [n_cams][CCAL_PMAX]
    std::vector<double> lo, hi;
    std::vector<uint8_t> has_bound, fixed;
    // mode N / solver workspaces (allocated lazily)
    ccal::NormalWs* nws = nullptr;
    // lazily sized scratch for host<->device staging of ccal_eval
    double *d_r = nullptr, *d_J = nullptr, *d_err = nullptr;
    ccal_allreduce_fn allreduce = nullptr;     // callback form of the step's collective (tests over gloo)
    void* allreduce_user = nullptr;
    void* rccl_comm = nullptr;                 // ncclComm_t: the library issues ncclAllReduce itself (ccal_set_rccl_comm)
    void* peer = nullptr;                      // the library's in-process transport (ccal_multi.hip, an InprocRank): the deciding kernel
                                               // adds the ranks' buffers itself; stream-ordered, groups are enqueued ahead like with RCCL
    bool counted = false;                      // registered in ctx->n_problems (creation succeeded)
    bool sharded() const { return allreduce != nullptr || rccl_comm != nullptr || peer != nullptr; }
};

// No exception crosses the C ABI (include/ccal.h:9): every extern "C" body that can allocate host memory (operator new,
// std::vector, std::string) runs between CCAL_API_TRY and CCAL_API_CATCH(ctx).  ctx may be NULL.
#define CCAL_API_TRY try {
#define CCAL_API_CATCH(ctx_expr)                                                                   \
    }                                                                                              \
    catch (const std::bad_alloc&) { ccal::note_error((ctx_expr), "out of host memory"); return CCAL_ERR_NO_MEMORY; }  \
    catch (const std::exception& e_) { ccal::note_error((ctx_expr), e_.what()); return CCAL_ERR_HIP; }                \
    catch (...) { ccal::note_error((ctx_expr), "unknown C++ exception"); return CCAL_ERR_HIP; }

namespace ccal {
inline void note_error(ccal_ctx* ctx, const char* msg) noexcept {
    if (!ctx) return;
    try { ctx->err = msg; } catch (...) { }
}
inline void note_error(const ccal_ctx*, const char*) noexcept {}
// kernel launchers (ccal_kernels.hip)
hipError_t launch_eval(const ccal_problem* p, int cam, const KArgs& a, hipStream_t s);
hipError_t launch_reproj_err(const ccal_problem* p, int cam, const KArgs& a, hipStream_t s);
// ccal_rccl.hip: in-place sum over the ranks of an ncclComm_t, ordered on the stream; returns a ccal_status
int rccl_allreduce_sum(ccal_ctx* ctx, void* comm, double* buf, size_t count, hipStream_t st);
// ccal_solver.hip: wait for the early-exit groups a finished solve left in the stream (no-op if there are none)
int drain_pending_groups(ccal_problem* p);
// ccal_kernels_stats.hip
// first size of ccal_problem::d_scratch: what ccal_init_poses and validation() of every camera need (gathered values + the selection's work area)
inline size_t problem_scratch_hint(const ccal_problem* p) {
    return (size_t)std::max<int64_t>(p->n_corners, 1) * 32 + (size_t)(std::max(p->n_obs, 1) + 1) * 64 + (size_t)384 * 1024;
}
hipError_t validation_stats_device(ccal_problem* p, int cam, const double* d_err, double* avg_99, double* median, hipStream_t s);
hipError_t camera_errors_device(ccal_problem* p, int cam, const double* d_err, double** d_out, int64_t* n_out, hipStream_t s);     // *d_out: a slice of p->d_scratch
size_t order_stats_block_bytes(int64_t n, hipStream_t s);      // block size order_stats_block needs for n values (0: sizing failed)
hipError_t order_stats_block(char* block /* values in front */, size_t block_bytes, int64_t n, double* avg_99, double* median, hipStream_t s);
// ccal_api.hip: reprojection errors of every corner at the given parameters into p->d_err (device); ccal_multi.hip uses it per shard
int reprojection_errors_dev(ccal_problem* p, const double* intr, const double* poses, const double* extr);
// ccal_kernels_init.hip
hipError_t launch_pose_init(const ccal_problem* p, int cam, const double* d_intr, double* d_poses_obs, int32_t* d_valid,
                            int min_points, hipStream_t s);
// Dynamic LDS above 48 KiB has to be enabled per kernel AND per device: remember the largest size enabled on each
// device of this process (contexts on several GPUs may share the process).  Launchers run on any host thread
// (ccal_solve_batch / ccal_solve_sharded drive one thread per context): the fast path is one atomic load, the
// check-and-set runs under one process-wide mutex, so the attribute only ever grows and `enabled` never claims more
// than what was last set.
struct DynLdsGuard { std::atomic<size_t> enabled[16] = {}; };
inline std::mutex& dyn_lds_mutex() { static std::mutex m; return m; }
inline hipError_t ensure_dyn_lds(const void* fn, size_t lds, DynLdsGuard& g) {
    if (lds <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<size_t>& have = g.enabled[dev & 15];
    if (lds <= have.load(std::memory_order_acquire)) return hipSuccess;
    std::lock_guard<std::mutex> lk(dyn_lds_mutex());
    if (lds <= have.load(std::memory_order_relaxed)) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) have.store(lds, std::memory_order_release);
    else (void)hipGetLastError();      // reported here: the runtime's sticky copy must not surface in an unrelated launcher's hipGetLastError()
    return e;
}

// ccal_multi.hip: the in-process transport of single-process sharded solves (shards on one GPU, or on GPUs with peer
// access when RCCL is not there) - events order the ranks' streams, the deciding kernel (k_head / k_solve) adds the ranks'
// buffers in rank order itself: PeerView = every rank's buffer of the step's sums, in its argument block.
constexpr int kMaxPeers = CCAL_MULTI_MAX_DEVICES;
struct PeerView { const double* src[kMaxPeers]; int32_t n; };
struct InprocComm;
InprocComm* inproc_create(int n, const int* devices, std::string* err);
void inproc_destroy(InprocComm* c);
void inproc_abort(InprocComm* c);                       // a rank failed: release the ranks waiting in the host rendezvous
bool inproc_aborted(const InprocComm* c);
int inproc_recover(InprocComm* c);                      // after an abort, no rank inside: drain the devices, reset the rendezvous
void inproc_set_timeout(InprocComm* c, double seconds); // host rendezvous: how long a rank waits for its peers (<= 0: 600 s)
void* inproc_rank_handle(InprocComm* c, int rank);      // ccal_problem::peer of that rank
int inproc_parity(const void* rank_handle);             // which of its two sum buffers the rank's next collective uses
int inproc_post(void* rank_handle, const double* device_buf, size_t count, void* hip_stream, PeerView* out);   // 0 = ok
// a context's persistent helper thread (ccal_solver.hip; created on first use): run fn on it / wait until it has finished
void ctx_worker_submit(ccal_ctx* ctx, std::function<void()> fn);
void ctx_worker_wait(ccal_ctx* ctx);
// why the last ccal_ctx_create / ccal_multi_create* on this thread failed (ccal_create_last_error)
void note_create_error(const std::string& msg) noexcept;
// ccal_rccl.hip: communicators of one process, one per device (ncclCommInitAll); abort = ncclCommAbort
int rccl_comm_init_all(const int* devices, int n, void** comms_out, std::string* err);
void rccl_comm_abort(void* comm);
// ccal_solver.hip: n shards of ONE problem (distinct contexts, a transport set on every shard), each driven by a host
// thread of its own; `inproc` (may be NULL) is aborted when a rank fails
int solve_sharded_run(ccal_problem** shards, int n, const ccal_solver_opts* o, double* intr_io, double* const* poses_io,
                      double* extr_io, ccal_report* rep, InprocComm* inproc);
}  // namespace ccal
