// C ABI of the engine (include/ccal.h): context, problem residency, mode E entry points,
// constraints, validation.  The solver loop lives in ccal_solver.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>

#include "ccal_internal.hpp"
#include "ccal_normal.hpp"

using namespace ccal;

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return CCAL_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

// the context stream has just been synchronised: whatever early-exit groups a finished solve left in it are gone
static void stream_synced(ccal_problem* p) {
    if (!p->nws) return;
    p->nws->tail_pending = false;
    if (p->nws->fws) p->nws->fws->tail_pending = false;
}

static int fail(ccal_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// One element of slack behind every array: the Gram / eval kernels request a frame's first corner row before they look at
// its corner count (software prefetch), which for an EMPTY last frame is index n - one past the data, inside the allocation.
template <class T>
static int upload(ccal_ctx* ctx, T** dst, const T* src, size_t n) {
    HIP_TRY(ctx, hipMalloc((void**)dst, (n + 1) * sizeof(T)));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(*dst + n, 0, sizeof(T), ctx->stream));
    return CCAL_OK;
}

static void default_conventions(ccal_model_conventions* cv) {
    cv->kb4_small_radius = kDefaultKb4SmallRadius;
    cv->unproject_small_radius = kDefaultUnprojectSmallRadius;
    for (int i = 0; i < 5; ++i) cv->ocv5_order[i] = i;
    cv->reserved_ = 0;
    for (int m = 0; m < kNumModels; ++m)
        for (int i = 0; i < kMaxDist; ++i) { cv->dist_lo[m][i] = kDefaultDistLo[m][i]; cv->dist_hi[m][i] = kDefaultDistHi[m][i]; }
}
static_assert(sizeof(((ccal_model_conventions*)nullptr)->dist_lo) == sizeof(kDefaultDistLo), "conventions table shape");

extern "C" {

int ccal_get_model_conventions(const ccal_ctx* ctx, ccal_model_conventions* out) {
    if (!ctx || !out) return CCAL_ERR_INVALID_ARG;
    *out = ctx->conv;
    return CCAL_OK;
}
int ccal_set_model_conventions(ccal_ctx* ctx, const ccal_model_conventions* in) {
    if (!ctx) return CCAL_ERR_INVALID_ARG;
    if (!in) { default_conventions(&ctx->conv); return CCAL_OK; }
    if (!(in->kb4_small_radius >= 0.0) || !(in->unproject_small_radius >= 0.0)) return CCAL_ERR_INVALID_ARG;
    {   // ocv5_order: a permutation of 0..4
        int seen = 0;
        for (int i = 0; i < 5; ++i) { if (in->ocv5_order[i] < 0 || in->ocv5_order[i] > 4) return CCAL_ERR_INVALID_ARG; seen |= 1 << in->ocv5_order[i]; }
        if (seen != 31) return CCAL_ERR_INVALID_ARG;
    }
    for (int m = 0; m < kNumModels; ++m)
        for (int i = 0; i < model_np(m) - 4; ++i) if (!(in->dist_lo[m][i] <= in->dist_hi[m][i])) return CCAL_ERR_INVALID_ARG;
    ctx->conv = *in;
    return CCAL_OK;
}

const char* ccal_version(void) { return "ccal-mi355x 0.2.0 (gfx950)"; }
int ccal_model_num_params(int model) { return (model >= 0 && model < kNumModels) ? model_np(model) : model == CCAL_MODEL_EUCMT ? 8 : -1; }

int ccal_ctx_create(int device_id, void* hip_stream, ccal_ctx** out) {
    if (!out) return CCAL_ERR_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); note_create_error("ccal_ctx_create: no HIP device"); return CCAL_ERR_HIP; }
    if (device_id < 0 || device_id >= n) { note_create_error("ccal_ctx_create: device " + std::to_string(device_id) + " of " + std::to_string(n)); return CCAL_ERR_HIP; }
    ccal_ctx* c = new (std::nothrow) ccal_ctx();
    if (!c) return CCAL_ERR_NO_MEMORY;
    default_conventions(&c->conv);
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) { delete c; note_create_error("ccal_ctx_create: hipSetDevice failed"); return CCAL_ERR_HIP; }
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; note_create_error("ccal_ctx_create: hipStreamCreate failed"); return CCAL_ERR_HIP; }
        c->own_stream = true;
    }
    *out = c;
    return CCAL_OK;
}
static void ctx_free(ccal_ctx* ctx) {
    ctx_worker_destroy(ctx);
    (void)hipSetDevice(ctx->device);
    ctx_cache_clear(ctx);
    if (!ctx->pinned.empty()) {
        (void)hipSetDevice(ctx->device);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        for (auto& pr : ctx->pinned) (void)hipHostUnregister(pr.first);
        (void)hipGetLastError();
        ctx->pinned.clear();
    }
    if (ctx->d_batch_tab || ctx->h_batch_tab) {
        (void)hipSetDevice(ctx->device);
        if (ctx->d_batch_tab) (void)hipFree(ctx->d_batch_tab);
        if (ctx->h_batch_tab) (void)hipHostFree(ctx->h_batch_tab);
    }
    if (ctx->own_stream && ctx->stream) { (void)hipSetDevice(ctx->device); (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
}
void ccal_ctx_destroy(ccal_ctx* ctx) {
    if (!ctx) return;
    if (ctx->n_problems > 0) { ctx->destroy_requested = true; return; }     // freed by its last problem
    ctx_free(ctx);
}
const char* ccal_last_error(const ccal_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int ccal_sync(ccal_ctx* ctx) {
    if (!ctx) return CCAL_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CCAL_OK;
}

int ccal_pin_buffer(ccal_ctx* ctx, void* host_ptr, size_t bytes) {
    if (!ctx || !host_ptr || !bytes) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (auto& pr : ctx->pinned) if (pr.first == host_ptr) return fail(ctx, CCAL_ERR_INVALID_ARG, "ccal_pin_buffer: this address is pinned already");
    if (pinned_device_ptr(host_ptr, bytes)) return CCAL_OK;          // pinned by the caller itself (hipHostMalloc / hipHostRegister): nothing to do
    ctx->pinned.reserve(ctx->pinned.size() + 1);
    const hipError_t e = hipHostRegister(host_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(ctx, CCAL_ERR_HIP, (std::string("ccal_pin_buffer: hipHostRegister: ") + hipGetErrorString(e)).c_str()); }
    ctx->pinned.emplace_back(host_ptr, bytes);
    return CCAL_OK;
    CCAL_API_CATCH(ctx)
}
int ccal_unpin_buffer(ccal_ctx* ctx, void* host_ptr) {
    if (!ctx || !host_ptr) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    for (size_t i = 0; i < ctx->pinned.size(); ++i) {
        if (ctx->pinned[i].first != host_ptr) continue;
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));             // nothing may still read or write the range
        ctx->pinned.erase(ctx->pinned.begin() + (ptrdiff_t)i);
        HIP_TRY(ctx, hipHostUnregister(host_ptr));
        return CCAL_OK;
    }
    return CCAL_OK;                                                   // not registered through this context (the caller's own pinning): nothing to undo
    CCAL_API_CATCH(ctx)
}

int ccal_set_defaults(ccal_solver_opts* o) {
    if (!o) return CCAL_ERR_INVALID_ARG;
    o->method = CCAL_METHOD_GN; o->max_iterations = 100;
    o->min_abs_error_decrease = 1e-5; o->min_rel_error_decrease = 1e-5; o->min_error = 1e-10;
    o->lm_initial_radius = 1e4; o->lm_min_diagonal = 1e-6; o->lm_max_diagonal = 1e32;
    o->verbose = 0; o->timeout_s = 0;
    o->error_metric = CCAL_ERROR_SQUARED_NORM; o->reserved_ = 0;
    return CCAL_OK;
}

int ccal_problem_create(ccal_ctx* ctx, const ccal_problem_desc* d, ccal_problem** out) {
    if (!ctx || !d || !out) return CCAL_ERR_INVALID_ARG;
    *out = nullptr;
    CCAL_API_TRY
    if (d->n_cams < 1 || d->n_cams > CCAL_MAX_CAMS || d->n_slots < 0 || d->n_obs < 0 || !d->model ||
        (d->n_obs > 0 && (!d->obs_cam || !d->obs_slot || !d->obs_offsets)))
        return fail(ctx, CCAL_ERR_INVALID_ARG, "bad problem description");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    struct Destroy { void operator()(ccal_problem* q) const { ccal_problem_destroy(q); } };
    std::unique_ptr<ccal_problem, Destroy> hold(new ccal_problem());      // freed on every early return and on a throw
    ccal_problem* p = hold.get();
    p->ctx = ctx; p->n_cams = d->n_cams; p->n_slots = d->n_slots; p->n_obs = d->n_obs;
    p->one_focal = d->xy_same_focal != 0; p->huber_delta = d->huber_delta;
    p->cams.resize(d->n_cams);
    int K = 0;
    for (int c = 0; c < d->n_cams; ++c) {
        CamLayout& cl = p->cams[c];
        cl.model = d->model[c];
        if (cl.model == CCAL_MODEL_EUCMT) return fail(ctx, CCAL_ERR_UNSUPPORTED, "EUCMT is a parameter container in this build (its projection is only in the absent camera-intrinsic-model crate)");
        if (cl.model < 0 || cl.model >= kNumModels) return fail(ctx, CCAL_ERR_INVALID_ARG, "unknown camera model");
        cl.P = model_np(cl.model); cl.Peff = cl.P - (p->one_focal ? 1 : 0);
        cl.D = cl.Peff + (c == 0 ? 6 : 12);
        cl.col_theta = K; K += cl.Peff;
        if (c > 0) { cl.col_extr = K; K += 6; }
        cl.width = d->width ? d->width[c] : 0.0; cl.height = d->height ? d->height[c] : 0.0;
    }
    if (K >= CCAL_KMAX) return fail(ctx, CCAL_ERR_INVALID_ARG, "reduced system too large");      // k_solve: thread i owns row i of K + 1 <= 128 (row K = right-hand side)
    p->K = K;
    if (d->n_obs > 0) {
        p->h_obs_off.assign(d->obs_offsets, d->obs_offsets + d->n_obs + 1);
        p->h_obs_cam.assign(d->obs_cam, d->obs_cam + d->n_obs);
        p->h_obs_slot.assign(d->obs_slot, d->obs_slot + d->n_obs);
    } else {
        p->h_obs_off.assign(1, 0);
    }
    p->h_joff.resize(d->n_obs + 1);
    // one observation frame per (camera, slot): the reference has one FrameFeature per camera and frame index
    // (src/util.rs:595-601), and the single-camera kernels index their per-frame records by slot
    std::vector<uint8_t> seen((size_t)d->n_cams * (size_t)std::max(d->n_slots, 1), 0);
    int64_t j = 0;
    for (int o = 0; o < d->n_obs; ++o) {
        const int cam = p->h_obs_cam[o], slot = p->h_obs_slot[o];
        const int64_t n = p->h_obs_off[o + 1] - p->h_obs_off[o];
        if (cam < 0 || cam >= d->n_cams || slot < 0 || slot >= d->n_slots || n < 0 || p->h_obs_off[0] != 0)
            return fail(ctx, CCAL_ERR_INVALID_ARG, "bad observation frame table");
        uint8_t& sn = seen[(size_t)cam * d->n_slots + slot];
        if (sn) return fail(ctx, CCAL_ERR_INVALID_ARG, "two observation frames for the same (camera, slot)");
        sn = 1;
        p->h_joff[o] = j; j += n * 2 * p->cams[cam].D;
        p->cams[cam].obs.push_back(o);
    }
    p->h_joff[d->n_obs] = j; p->j_len = j;
    p->slot_ident = d->n_slots == d->n_obs;
    for (int o = 0; o < d->n_obs && p->slot_ident; ++o) p->slot_ident = p->h_obs_slot[o] == o;
    p->n_corners = p->h_obs_off[d->n_obs];
    // the Gram kernels address a problem's corner rows by 32-bit byte offsets (ccal_kernels_gram2.hip); 2^30 corners are 21 GB of
    // inputs and 245 GB of mode-E outputs - beyond that, shard the frames (ccal_multi_*)
    if (p->n_corners >= ((int64_t)1 << 30)) return fail(ctx, CCAL_ERR_INVALID_ARG, "more than 2^30 - 1 corners in one problem: shard the frames (ccal_multi_problem_create)");
    const size_t nc = (size_t)p->n_corners;
    if (nc && (!d->p3d_x || !d->p3d_y || !d->p3d_z || !d->p2d_u || !d->p2d_v)) return fail(ctx, CCAL_ERR_INVALID_ARG, "null corner arrays");
    // ONE device allocation, cleared once, sliced (a calibration session creates its problem once: sixteen hipMalloc + memset pairs
    // were a third of ccal_problem_create's 0.25 ms at 600 frames).  One element of slack behind every uploaded array, as upload().
    const size_t ni = (size_t)d->n_cams * CCAL_PMAX, np6 = (size_t)std::max(d->n_slots, 1) * 6, ne = (size_t)d->n_cams * 6;
    auto up256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t total = 5 * up256((nc + 1) * sizeof(float)) + 2 * up256((p->h_obs_off.size() + 1) * sizeof(int64_t)) +
                   2 * up256((p->h_obs_cam.size() + 1) * sizeof(int32_t)) + 2 * (up256(ni * 8) + up256(np6 * 8) + up256(ne * 8));
    for (int c = 0; c < d->n_cams; ++c) total += up256((p->cams[c].obs.size() + 1) * sizeof(int32_t));
    // single camera, ragged frames (real sessions: 24 .. 144 corners per frame, src/data_loader.rs:15): the Gram launch works on bins of
    // frames sorted by corner count, each bin with the lanes per frame its frames need (ccal_kernels_gram2.hip); the sorted table -
    // { frame, first corner, corners, slot } per position - is made once, here
    std::vector<int32_t> bin_tab;
    if (d->n_cams == 1 && d->n_obs > 0) {
        std::vector<int32_t> order;
        const int m0 = p->cams[0].model;
        p->gram_bins = gram2_bin_plan(p->h_obs_off.data(), d->n_obs, m0 == kUCM || m0 == kEUCM, &order);
        if (p->gram_bins.n_bins > 0) {
            bin_tab.assign((size_t)d->n_obs * 4 + 4, 0);               // (+ one spare entry: a 16-byte load never ends outside the slice)
            for (int i = 0; i < d->n_obs; ++i) {
                const int o = order[(size_t)i];
                bin_tab[4 * (size_t)i] = o; bin_tab[4 * (size_t)i + 1] = (int32_t)p->h_obs_off[o];
                bin_tab[4 * (size_t)i + 2] = (int32_t)(p->h_obs_off[o + 1] - p->h_obs_off[o]); bin_tab[4 * (size_t)i + 3] = p->h_obs_slot[o];
            }
            total += up256((bin_tab.size() + 1) * sizeof(int32_t));
        }
    }
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&p->d_block, total));
    // Session-sized problems (up to 8 MB of inputs) are packed into ONE pinned staging block laid out like the device block and
    // uploaded with ONE copy: ten hipMemcpyAsync calls from pageable memory - each staged by the runtime on its own - were
    // ~0.1 ms of ccal_problem_create's 0.2 ms at 600 frames.  Larger problems keep the per-array copies (packing 29 MB on the
    // host would cost more than it saves).
    char* h_pack = nullptr;
    struct PackGuard { ccal_ctx* c; char*& p; ~PackGuard() { if (p) { (void)hipStreamSynchronize(c->stream); ccal::ctx_release(c, p, true); } } } pack_guard{ ctx, h_pack };      // (a copy may still read it on an error exit)
    if (total <= ((size_t)8 << 20)) HIP_TRY(ctx, ctx_host_alloc(ctx, (void**)&h_pack, total));
    if (!h_pack) HIP_TRY(ctx, hipMemsetAsync(p->d_block, 0, total, ctx->stream));
    {
        char* q = p->d_block;
        auto put = [&](auto** dst, const auto* src, size_t n) -> hipError_t {
            using T = std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>;
            *dst = reinterpret_cast<T*>(q);
            const size_t room = up256((n + 1) * sizeof(T));
            if (h_pack) {                 // (the slack behind the data must be zeros: the kernels load - and mask - one element past an empty last frame)
                char* h = h_pack + (q - p->d_block);
                const size_t nb = (n && src) ? n * sizeof(T) : 0;
                if (nb) std::memcpy(h, src, nb);
                std::memset(h + nb, 0, room - nb);
                q += room;
                return hipSuccess;
            }
            q += room;
            return (n && src) ? hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream) : hipSuccess;
        };
        HIP_TRY(ctx, put(&p->d_x, d->p3d_x, nc)); HIP_TRY(ctx, put(&p->d_y, d->p3d_y, nc)); HIP_TRY(ctx, put(&p->d_z, d->p3d_z, nc));
        HIP_TRY(ctx, put(&p->d_u, d->p2d_u, nc)); HIP_TRY(ctx, put(&p->d_v, d->p2d_v, nc));
        HIP_TRY(ctx, put(&p->d_obs_off, (const int64_t*)p->h_obs_off.data(), p->h_obs_off.size()));
        HIP_TRY(ctx, put(&p->d_joff, (const int64_t*)p->h_joff.data(), p->h_joff.size()));
        HIP_TRY(ctx, put(&p->d_obs_cam, (const int32_t*)p->h_obs_cam.data(), p->h_obs_cam.size()));
        HIP_TRY(ctx, put(&p->d_obs_slot, (const int32_t*)p->h_obs_slot.data(), p->h_obs_slot.size()));
        for (int c = 0; c < d->n_cams; ++c) HIP_TRY(ctx, put(&p->cams[c].d_obs, (const int32_t*)p->cams[c].obs.data(), p->cams[c].obs.size()));
        if (!bin_tab.empty()) HIP_TRY(ctx, put(&p->d_bin_tab, (const int32_t*)bin_tab.data(), bin_tab.size()));
        double** bufs[6] = { &p->d_intr, &p->d_poses, &p->d_extr, &p->d_intr_c, &p->d_poses_c, &p->d_extr_c };
        const size_t sz[6] = { ni, np6, ne, ni, np6, ne };
        const size_t packed = (size_t)(q - p->d_block);                     // the uploaded arrays; the parameter sets behind them start as zeros
        for (int i = 0; i < 6; ++i) { *bufs[i] = reinterpret_cast<double*>(q); q += up256(sz[i] * 8); }
        if ((size_t)(q - p->d_block) > total) return fail(ctx, CCAL_ERR_HIP, "problem block layout");
        if (h_pack) {
            HIP_TRY(ctx, hipMemcpyAsync(p->d_block, h_pack, packed, hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(p->d_block + packed, 0, total - packed, ctx->stream));
        }
    }
    p->lo.assign(ni, 0.0); p->hi.assign(ni, 0.0); p->has_bound.assign(ni, 0); p->fixed.assign(ni, 0);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(ctx, CCAL_ERR_HIP, "upload failed");
    p->counted = true;
    ctx->n_problems += 1;
    *out = hold.release();
    return CCAL_OK;
    CCAL_API_CATCH(ctx)
}

void ccal_problem_destroy(ccal_problem* p) {
    if (!p) return;
    // early-exit groups of the last solve may still be queued and hold these pointers: drain them BEFORE anything is freed
    // (normal_ws_destroy would, but only after the parameter and input buffers are gone)
    if (p->nws && p->ctx && (p->nws->tail_pending || (p->nws->fws && p->nws->fws->tail_pending))) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipDeviceSynchronize();
        p->nws->tail_pending = false;
        if (p->nws->fws) p->nws->fws->tail_pending = false;
    }
    ccal_ctx* ctx = p->ctx;
    // nothing of this problem may still run when its blocks go back to the context (the next problem gets them at once): the
    // stream is drained - it is idle after every blocking entry point; a device-resident solve may have left its last launches
    // (a caller-provided stream may be gone already when a binding's garbage collector gets here: the device is drained then)
    if (ctx && (p->d_block || p->nws)) {
        (void)hipSetDevice(ctx->device);
        if (ctx->own_stream && ctx->stream) (void)hipStreamSynchronize(ctx->stream); else (void)hipDeviceSynchronize();
        (void)hipGetLastError();
    }
    void* ptrs[] = { p->d_block, p->d_r, p->d_J, p->d_err, p->d_scratch };          // (corner arrays, frame tables, parameter arrays: slices of d_block)
    for (void* q : ptrs) if (q) ctx_release(ctx, q, false);
    normal_ws_destroy(p);
    const bool counted = p->counted;
    delete p;
    if (ctx && counted && --ctx->n_problems == 0 && ctx->destroy_requested) ctx_free(ctx);
}

int64_t ccal_num_corners(const ccal_problem* p) { return p ? p->n_corners : -1; }
int ccal_reduced_dim(const ccal_problem* p) { return p ? p->K : -1; }
int ccal_block_dim(const ccal_problem* p, int cam) { return (p && cam >= 0 && cam < p->n_cams) ? p->cams[cam].D : -1; }
int ccal_eff_num_params(const ccal_problem* p, int cam) { return (p && cam >= 0 && cam < p->n_cams) ? p->cams[cam].Peff : -1; }
int64_t ccal_jacobian_len(const ccal_problem* p) { return p ? p->j_len : -1; }

static bool idx_ok(const ccal_problem* p, int cam, int idx) {
    return p && cam >= 0 && cam < p->n_cams && idx >= 0 && idx < p->cams[cam].Peff;
}
int ccal_set_bounds(ccal_problem* p, int cam, int idx, double lo, double hi) {
    if (!idx_ok(p, cam, idx) || !(lo <= hi)) return CCAL_ERR_INVALID_ARG;
    p->lo[cam * CCAL_PMAX + idx] = lo; p->hi[cam * CCAL_PMAX + idx] = hi; p->has_bound[cam * CCAL_PMAX + idx] = 1;
    return CCAL_OK;
}
int ccal_clear_bounds(ccal_problem* p, int cam, int idx) {
    if (!idx_ok(p, cam, idx)) return CCAL_ERR_INVALID_ARG;
    p->has_bound[cam * CCAL_PMAX + idx] = 0; return CCAL_OK;
}
int ccal_fix_param(ccal_problem* p, int cam, int idx) {
    if (!idx_ok(p, cam, idx)) return CCAL_ERR_INVALID_ARG;
    p->fixed[cam * CCAL_PMAX + idx] = 1; return CCAL_OK;
}
int ccal_unfix_param(ccal_problem* p, int cam, int idx) {
    if (!idx_ok(p, cam, idx)) return CCAL_ERR_INVALID_ARG;
    p->fixed[cam * CCAL_PMAX + idx] = 0; return CCAL_OK;
}

// set_problem_parameter_bound (src/util.rs:29-49).  The distortion bounds come from camera-intrinsic-model's
// distortion_params_bound(), whose source is absent: the values are the context's conventions table (defaults in
// ccal_models.hpp, ccal_set_model_conventions to change them); single entries can be overridden with ccal_set_bounds.
int ccal_apply_reference_bounds(ccal_problem* p) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    const int shift = p->one_focal ? 1 : 0;
    const ccal_model_conventions& cv = p->ctx->conv;
    for (int c = 0; c < p->n_cams; ++c) {
        const CamLayout& cl = p->cams[c];
        ccal_set_bounds(p, c, 0, 0.0, 10000.0);
        ccal_set_bounds(p, c, 1 - shift, 0.0, 10000.0);
        ccal_set_bounds(p, c, 2 - shift, 0.0, cl.width);
        ccal_set_bounds(p, c, 3 - shift, 0.0, cl.height);
        // the table is indexed by coefficient (OPENCV5: k1, k2, p1, p2, k3), the bound goes to where the caller's vector keeps it
        for (int i = 4; i < cl.P; ++i) {
            const int at = cl.model == kOCV5 ? 4 + cv.ocv5_order[i - 4] : i;
            ccal_set_bounds(p, c, at - shift, cv.dist_lo[cl.model][i - 4], cv.dist_hi[cl.model][i - 4]);
        }
    }
    return CCAL_OK;
}
// set_problem_parameter_disabled (src/util.rs:50-71): fix the last k distortion parameters at 0.
int ccal_disable_distortions(ccal_problem* p, int n_disabled, double* intr_io) {
    if (!p || n_disabled < 0) return CCAL_ERR_INVALID_ARG;
    const int shift = p->one_focal ? 1 : 0;
    for (int c = 0; c < p->n_cams; ++c) {
        for (int i = 0; i < n_disabled; ++i) {
            const int eff = p->cams[c].P - 1 - shift - i;
            if (eff < 0) return CCAL_ERR_INVALID_ARG;
            ccal_fix_param(p, c, eff);
            if (intr_io) intr_io[c * CCAL_PMAX + eff + shift] = 0.0;   // eff -> full index
        }
    }
    return CCAL_OK;
}
int ccal_set_allreduce(ccal_problem* p, ccal_allreduce_fn fn, void* user) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    if (p->allreduce != fn || p->allreduce_user != user) { const int rc = drain_pending_groups(p); if (rc != CCAL_OK) return rc; }
    p->allreduce = fn; p->allreduce_user = user;
    return CCAL_OK;
}

int ccal_upload_params(ccal_problem* p, const double* intr, const double* poses, const double* extr) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (intr) HIP_TRY(ctx, hipMemcpyAsync(p->d_intr, intr, sizeof(double) * p->n_cams * CCAL_PMAX, hipMemcpyHostToDevice, ctx->stream));
    if (poses && p->n_slots) HIP_TRY(ctx, hipMemcpyAsync(p->d_poses, poses, sizeof(double) * p->n_slots * 6, hipMemcpyHostToDevice, ctx->stream));
    if (extr) HIP_TRY(ctx, hipMemcpyAsync(p->d_extr, extr, sizeof(double) * p->n_cams * 6, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // host buffers may be reused by the caller
    stream_synced(p);
    return CCAL_OK;
}
int ccal_download_params(ccal_problem* p, double* intr, double* poses, double* extr) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (intr) HIP_TRY(ctx, hipMemcpyAsync(intr, p->d_intr, sizeof(double) * p->n_cams * CCAL_PMAX, hipMemcpyDeviceToHost, ctx->stream));
    if (poses && p->n_slots) HIP_TRY(ctx, hipMemcpyAsync(poses, p->d_poses, sizeof(double) * p->n_slots * 6, hipMemcpyDeviceToHost, ctx->stream));
    if (extr) HIP_TRY(ctx, hipMemcpyAsync(extr, p->d_extr, sizeof(double) * p->n_cams * 6, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    stream_synced(p);
    if (intr && p->one_focal) for (int c = 0; c < p->n_cams; ++c) intr[c * CCAL_PMAX + 1] = intr[c * CCAL_PMAX];   // fy = f (src/util.rs:467-470)
    return CCAL_OK;
}

static KArgs make_args(const ccal_problem* p, int cam) {
    KArgs a = {};
    a.x = p->d_x; a.y = p->d_y; a.z = p->d_z; a.u = p->d_u; a.v = p->d_v;
    a.obs_off = p->d_obs_off; a.obs_slot = p->d_obs_slot; a.joff = p->d_joff;
    a.list = p->cams[cam].d_obs; a.n_list = (int32_t)p->cams[cam].obs.size(); a.cam = cam;
    a.intr = p->d_intr; a.poses = p->d_poses; a.extr = p->d_extr;
    a.huber_delta = p->huber_delta; a.rt = model_rt(p->ctx);
    return a;
}

int ccal_eval_dev(ccal_problem* p, int apply_loss, double* r_dev, double* J_dev) {
    if (!p || !r_dev || !J_dev) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));      // the caller's thread may have another device current
    for (int c = 0; c < p->n_cams; ++c) {
        KArgs a = make_args(p, c);
        a.apply_loss = apply_loss; a.r_out = r_dev; a.J_out = J_dev;
        HIP_TRY(ctx, launch_eval(p, c, a, ctx->stream));
    }
    return CCAL_OK;
}

int ccal_eval(ccal_problem* p, const double* intr, const double* poses, const double* extr,
              int apply_loss, double* r_out, double* J_out) {
    if (!p || !intr || (!poses && p->n_slots) || (!extr && p->n_cams > 1) || !r_out || !J_out) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    int rc = ccal_upload_params(p, intr, poses, extr);
    if (rc != CCAL_OK) return rc;
    if (!p->d_r) HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&p->d_r, sizeof(double) * std::max<int64_t>(2 * p->n_corners, 2)));
    if (!p->d_J) HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&p->d_J, sizeof(double) * std::max<int64_t>(p->j_len, 2)));
    rc = ccal_eval_dev(p, apply_loss, p->d_r, p->d_J);
    if (rc != CCAL_OK) return rc;
    if (p->n_corners) {
        HIP_TRY(ctx, hipMemcpyAsync(r_out, p->d_r, sizeof(double) * 2 * p->n_corners, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(J_out, p->d_J, sizeof(double) * p->j_len, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CCAL_OK;
}

}  // extern "C"
namespace ccal {
int reprojection_errors_dev(ccal_problem* p, const double* intr, const double* poses, const double* extr) {
    ccal_ctx* ctx = p->ctx;
    int rc = ccal_upload_params(p, intr, poses, extr);
    if (rc != CCAL_OK) return rc;
    if (!p->d_err) HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&p->d_err, sizeof(double) * std::max<int64_t>(p->n_corners, 1)));
    for (int c = 0; c < p->n_cams; ++c) {
        KArgs a = make_args(p, c);
        a.err_out = p->d_err;
        HIP_TRY(ctx, launch_reproj_err(p, c, a, ctx->stream));
    }
    return CCAL_OK;
}
}  // namespace ccal
extern "C" {

int ccal_reprojection_errors(ccal_problem* p, const double* intr, const double* poses, const double* extr, double* err_out) {
    if (!p || !intr || (!poses && p->n_slots) || (!extr && p->n_cams > 1) || !err_out) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    int rc = reprojection_errors_dev(p, intr, poses, extr);
    if (rc != CCAL_OK) return rc;
    if (p->n_corners) HIP_TRY(ctx, hipMemcpyAsync(err_out, p->d_err, sizeof(double) * p->n_corners, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CCAL_OK;
}

// Pose initialisation of every observation frame (src/util.rs:418-436), see ccal_kernels_init.hip.
int ccal_init_poses(ccal_problem* p, const double* intr, int min_points, double* poses_obs, int32_t* n_used) {
    if (!p || !intr || !poses_obs || !n_used) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    int rc = ccal_upload_params(p, intr, nullptr, nullptr);
    if (rc != CCAL_OK) return rc;
    const size_t no = (size_t)std::max(p->n_obs, 1);
    // the two temporaries are slices of the problem's scratch block (kept for the next call: no hipMalloc / hipFree pair per call)
    const size_t b_po = (no * 6 * sizeof(double) + 255) & ~(size_t)255, b_va = (no * sizeof(int32_t) + 255) & ~(size_t)255;
    if (p->scratch_bytes < b_po + b_va) {
        if (p->d_scratch) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx_release(ctx, p->d_scratch, false); p->d_scratch = nullptr; p->scratch_bytes = 0; }
        const size_t want = std::max(b_po + b_va, problem_scratch_hint(p));       // (room for validation()'s temporaries too: growing the block later costs a free)
        HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&p->d_scratch, want));
        p->scratch_bytes = want;
    }
    double* d_po = reinterpret_cast<double*>(p->d_scratch);
    int32_t* d_va = reinterpret_cast<int32_t*>(p->d_scratch + b_po);
    hipError_t e = hipMemsetAsync(d_va, 0, no * sizeof(int32_t), ctx->stream);
    for (int c = 0; c < p->n_cams && e == hipSuccess; ++c) e = launch_pose_init(p, c, p->d_intr, d_po, d_va, min_points, ctx->stream);
    if (e == hipSuccess && p->n_obs) {
        e = hipMemcpyAsync(poses_obs, d_po, (size_t)p->n_obs * 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(n_used, d_va, (size_t)p->n_obs * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->err = std::string("ccal_init_poses: ") + hipGetErrorString(e); return CCAL_ERR_HIP; }
    return CCAL_OK;
}

// validation() statistics of one camera (src/util.rs:778-795): errors of that camera's corners,
// sorted; median = e[len/2]; avg_99 = sum_{i < len*99/100} e_i / (len*99/100).
int ccal_validation(ccal_problem* p, int cam, const double* intr, const double* poses, const double* extr,
                    double* avg_99, double* median) {
    if (!p || cam < 0 || cam >= p->n_cams || !avg_99 || !median || !intr || (!poses && p->n_slots) || (!extr && p->n_cams > 1)) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    if (p->cams[cam].obs.empty()) return fail(ctx, CCAL_ERR_INVALID_ARG, "camera has no observations");
    int rc = ccal_upload_params(p, intr, poses, extr);
    if (rc != CCAL_OK) return rc;
    if (!p->d_err) HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&p->d_err, sizeof(double) * std::max<int64_t>(p->n_corners, 1)));
    KArgs a = make_args(p, cam);
    a.err_out = p->d_err;
    HIP_TRY(ctx, launch_reproj_err(p, cam, a, ctx->stream));
    HIP_TRY(ctx, validation_stats_device(p, cam, p->d_err, avg_99, median, ctx->stream));     // gather + radix sort + sums on the device
    return CCAL_OK;
}

}  // extern "C"
