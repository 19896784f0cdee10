// Mode N host side: workspaces, ccal_build_normal and the optimizer loop (ccal_solve).
//
// The loop restates tiny_solver::GaussNewtonOptimizer::optimize as the reference calls it
// (src/util.rs:443-463, 668-670; defaults in ccal_set_defaults) -- per iteration: Jacobian at x,
// solve the normal equations, x <- clamp(x + dx), error(x), stop on min_error / |d error| thresholds --
// and adds a Ceres-style Levenberg-Marquardt mode on the same kernels.  Both loops (single camera: solve_fused,
// everything else: the general loop in ccal_solve) are device-resident: the decisions run in a kernel, the host
// enqueues groups of launches one ahead and polls a status word in pinned memory.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include <cstdlib>

#include "ccal_fused.hpp"

using namespace ccal;

#define HIP_TRYN(ctx, expr)                                                                        \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return -CCAL_ERR_HIP;                                                                  \
        }                                                                                          \
    } while (0)

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                        \
            return CCAL_ERR_HIP;                                                                   \
        }                                                                                          \
    } while (0)

namespace ccal {

void normal_ws_destroy(ccal_problem* p) {
    NormalWs* w = p->nws;
    if (!w) return;
    // early-exit groups of the last solve may still be queued: they publish into the pinned status words freed below
    if (w->tail_pending || (w->fws && w->fws->tail_pending)) (void)hipStreamSynchronize(p->ctx->stream);
    void* ptrs[] = { w->G[0], w->G[1], w->cost_o[0], w->cost_o[1], w->d_goff, w->d_slot_off, w->d_slot_obs, w->d_obs_cam,
                     w->d_caminfo, w->partial, w->red, w->pf, w->dc, w->mc_slot, w->scal, w->flags, w->cols, w->d_slot_desc };
    for (void* q : ptrs) if (q) (void)hipFree(q);
    if (w->h_pinned) (void)hipHostFree(w->h_pinned);
    if (w->d_gstate) (void)hipFree(w->d_gstate);
    if (w->side) (void)hipStreamDestroy(w->side);
    if (w->h_gstatus) (void)hipHostFree(w->h_gstatus);
    if (w->h_gstate) (void)hipHostFree(w->h_gstate);
    if (FusedWs* f = w->fws) {
        void* fp[] = { f->pf[0], f->pf[1], f->praw[0], f->praw[1], f->partial, f->red, f->d_state, f->fcbuf, f->mc_f, f->cost_f };
        for (void* q : fp) if (q) (void)hipFree(q);
        if (f->h_status) (void)hipHostFree(f->h_status);
        if (f->h_stage) (void)hipHostFree(f->h_stage);
        if (f->d_stage) (void)hipFree(f->d_stage);
        if (f->side) (void)hipStreamDestroy(f->side);
        delete f;
    }
    delete w;
    p->nws = nullptr;
}

static int fused_ws_ensure(ccal_problem* p) {
    NormalWs* w = p->nws;
    if (w->fws) return CCAL_OK;
    ccal_ctx* ctx = p->ctx;
    FusedWs* f = new FusedWs();
    w->fws = f;
    const int K1 = p->K + 1;
    f->PRAW = (21 + 6 * K1 + K1 * K1 + 1) & ~1;
    f->RB1 = 2 * K1 * K1 + 2;
    const char* env_pw = std::getenv("CCAL_FUSED_WAVES");
    int n_pw = std::min(std::max(p->n_obs, 1), env_pw ? std::atoi(env_pw) : 16384);   // 4 workgroups of 4 waves per CU
    f->n_pw = (n_pw + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
    const size_t ns = (size_t)std::max(p->n_slots, 1), no = (size_t)std::max(p->n_obs, 1);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipMalloc((void**)&f->pf[i], ns * w->PF * sizeof(double)));
        HIP_TRY(ctx, hipMemset(f->pf[i], 0, ns * w->PF * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void**)&f->praw[i], no * f->PRAW * sizeof(double)));
        HIP_TRY(ctx, hipMemset(f->praw[i], 0, no * f->PRAW * sizeof(double)));
    }
    HIP_TRY(ctx, hipMalloc((void**)&f->fcbuf, no * 40 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&f->mc_f, no * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&f->cost_f, no * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&f->partial, (size_t)f->RB1 * f->n_pw * sizeof(double)));
    HIP_TRY(ctx, hipMemset(f->partial, 0, (size_t)f->RB1 * f->n_pw * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&f->red, (size_t)(f->RB1 + 7) * sizeof(double)));
    HIP_TRY(ctx, hipMemset(f->red, 0, (size_t)(f->RB1 + 7) * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&f->d_state, sizeof(DevState)));
    HIP_TRY(ctx, hipHostMalloc((void**)&f->h_status, sizeof(HostStatus), hipHostMallocCoherent | hipHostMallocMapped));
    const size_t stage_bytes = ns * 6 * sizeof(double) + CCAL_PMAX * sizeof(double) + sizeof(DevState) + CCAL_KMAX * sizeof(ColInfo) + 64;
    HIP_TRY(ctx, hipHostMalloc((void**)&f->h_stage, stage_bytes, hipHostMallocDefault));
    HIP_TRY(ctx, hipMalloc((void**)&f->d_stage, stage_bytes));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking));
    std::memset((void*)f->h_status, 0, sizeof(HostStatus));
    return CCAL_OK;
}

template <class T>
static int dev_upload(ccal_ctx* ctx, T** dst, const std::vector<T>& src) {
    HIP_TRY(ctx, hipMalloc((void**)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return CCAL_OK;
}

int normal_ws_ensure(ccal_problem* p) {
    if (p->nws) return CCAL_OK;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    NormalWs* w = new NormalWs();
    p->nws = w;
    w->K = p->K; w->RB = red_size(p->K); w->PF = pf_size(p->K);
    // persistent Schur waves: at most 2 workgroups per CU worth, never more than slots
    const char* env_sw = std::getenv("CCAL_SCHUR_WAVES");
    // persistent wavefronts of k_schur: 4 per SIMD (measured at 10 000 slots x 2 cameras: 2048 -> 165.7, 4096 -> 158.5, 8192 -> 173 us per build)
    int n_pw = std::min(std::max(p->n_slots, 1), env_sw ? std::max(4, std::atoi(env_sw)) : 4096);
    n_pw = (n_pw + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
    w->n_pw = n_pw;
    std::vector<int64_t> goff(p->n_obs);
    std::vector<int32_t> caminfo(p->n_cams * 4);
    int64_t gl = 0;
    for (int c = 0; c < p->n_cams; ++c) {
        const int ncp = (p->cams[c].D + 1) <= 16 ? 16 : 32;
        caminfo[c * 4 + 0] = p->cams[c].Peff; caminfo[c * 4 + 1] = p->cams[c].col_theta;
        caminfo[c * 4 + 2] = p->cams[c].col_extr; caminfo[c * 4 + 3] = ncp;
    }
    for (int o = 0; o < p->n_obs; ++o) { goff[o] = gl; const int ncp = caminfo[p->h_obs_cam[o] * 4 + 3]; gl += (int64_t)ncp * ncp; }
    w->g_len = gl;
    std::vector<int32_t> slot_off(p->n_slots + 1, 0), slot_obs(p->n_obs);
    for (int o = 0; o < p->n_obs; ++o) slot_off[p->h_obs_slot[o] + 1]++;
    for (int s = 0; s < p->n_slots; ++s) slot_off[s + 1] += slot_off[s];
    { std::vector<int32_t> cur(slot_off.begin(), slot_off.end() - 1);
      for (int o = 0; o < p->n_obs; ++o) slot_obs[cur[p->h_obs_slot[o]]++] = o; }
    std::vector<int64_t> slot_desc(p->n_obs);
    for (int i = 0; i < p->n_obs; ++i) slot_desc[i] = goff[slot_obs[i]] * 8 + p->h_obs_cam[slot_obs[i]];
    int rc;
    if ((rc = dev_upload(ctx, &w->d_slot_desc, slot_desc))) return rc;
    if ((rc = dev_upload(ctx, &w->d_goff, goff)) || (rc = dev_upload(ctx, &w->d_slot_off, slot_off)) ||
        (rc = dev_upload(ctx, &w->d_slot_obs, slot_obs)) || (rc = dev_upload(ctx, &w->d_obs_cam, p->h_obs_cam)) ||
        (rc = dev_upload(ctx, &w->d_caminfo, caminfo)))
        return rc;
    const size_t gbytes = std::max<int64_t>(gl, 1) * sizeof(double);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, hipMalloc((void**)&w->G[i], gbytes));
        HIP_TRY(ctx, hipMemset(w->G[i], 0, gbytes));     // tile (1,0) of two-tile blocks is never written
        HIP_TRY(ctx, hipMalloc((void**)&w->cost_o[i], std::max(p->n_obs, 1) * sizeof(double)));
        HIP_TRY(ctx, hipMemset(w->cost_o[i], 0, std::max(p->n_obs, 1) * sizeof(double)));
    }
    HIP_TRY(ctx, hipMalloc((void**)&w->partial, (size_t)w->RB * n_pw * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&w->red, (size_t)(w->RB + 8) * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&w->pf, (size_t)std::max(p->n_slots, 1) * w->PF * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&w->dc, CCAL_KMAX * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&w->mc_slot, (size_t)std::max(p->n_slots, 1) * sizeof(double)));
    HIP_TRY(ctx, hipMemset(w->mc_slot, 0, (size_t)std::max(p->n_slots, 1) * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&w->scal, 8 * sizeof(double)));
    HIP_TRY(ctx, hipMemset(w->scal, 0, 8 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&w->flags, 4 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&w->d_gstate, sizeof(DevState)));
    HIP_TRY(ctx, hipStreamCreateWithFlags(&w->side, hipStreamNonBlocking));
    HIP_TRY(ctx, hipHostMalloc((void**)&w->h_gstatus, sizeof(HostStatus), hipHostMallocCoherent | hipHostMallocMapped));
    HIP_TRY(ctx, hipHostMalloc((void**)&w->h_gstate, sizeof(DevState), hipHostMallocDefault));
    std::memset((void*)w->h_gstatus, 0, sizeof(HostStatus));
    HIP_TRY(ctx, hipMemset(w->flags, 0, 4 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&w->cols, CCAL_KMAX * sizeof(ColInfo)));
    HIP_TRY(ctx, hipHostMalloc((void**)&w->h_pinned, (size_t)(w->RB + 16) * sizeof(double), hipHostMallocDefault));
    return normal_upload_cols(p);
}

static void build_cols(const ccal_problem* p, ColInfo* cols) {
    std::memset(cols, 0, CCAL_KMAX * sizeof(ColInfo));
    for (int c = 0; c < p->n_cams; ++c) {
        const CamLayout& cl = p->cams[c];
        for (int i = 0; i < cl.Peff; ++i) {
            ColInfo& ci = cols[cl.col_theta + i];
            const int q = c * CCAL_PMAX + i;
            ci.lo = p->lo[q]; ci.hi = p->hi[q]; ci.has_bound = p->has_bound[q]; ci.fixed = p->fixed[q];
            ci.is_extr = 0;
            const int full = p->one_focal ? (i == 0 ? 0 : i + 1) : i;     // eff -> full index (fy re-inserted)
            ci.dst = c * CCAL_PMAX + full;
            ci.dst2 = (p->one_focal && i == 0) ? c * CCAL_PMAX + 1 : -1;
        }
        if (c > 0) for (int i = 0; i < 6; ++i) {
            ColInfo& ci = cols[cl.col_extr + i];
            ci.is_extr = 1; ci.dst = c * 6 + i; ci.dst2 = -1;
        }
    }
}

int normal_upload_cols(ccal_problem* p) {
    NormalWs* w = p->nws;
    ccal_ctx* ctx = p->ctx;
    std::vector<ColInfo> cols(CCAL_KMAX);
    build_cols(p, cols.data());
    HIP_TRY(ctx, hipMemcpyAsync(w->cols, cols.data(), CCAL_KMAX * sizeof(ColInfo), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CCAL_OK;
}

}  // namespace ccal

// gram (all cameras) at the current or candidate parameters into G[gbuf]
static int enqueue_gram(ccal_problem* p, bool cand, int gbuf) {
    ccal_ctx* ctx = p->ctx;
    for (int c = 0; c < p->n_cams; ++c) HIP_TRY(ctx, launch_gram(p, c, cand, gbuf, ctx->stream));
    return CCAL_OK;
}
// schur + reduce (+ all-reduce) of G[gbuf] -> red
static int enqueue_reduce_system(ccal_problem* p, int gbuf, double lambda, double min_diag, double max_diag) {
    ccal_ctx* ctx = p->ctx;
    NormalWs* w = p->nws;
    HIP_TRY(ctx, launch_schur(p, gbuf, lambda, min_diag, max_diag, ctx->stream));
    HIP_TRY(ctx, launch_reduce(p, ctx->stream));
    if (p->allreduce) {
        if (p->allreduce(p->allreduce_user, w->red, (size_t)w->RB, (void*)ctx->stream) != 0) { ctx->err = "all-reduce callback failed"; return CCAL_ERR_HIP; }
    }
    return CCAL_OK;
}

// Argument block of the single-camera kernels: set 0 = (p->d_intr, p->d_poses), set 1 = the *_c buffers.
static FusedArgs make_fused_args(const ccal_problem* p, double min_diag, double max_diag) {
    const NormalWs* w = p->nws;
    const FusedWs* f = w->fws;
    FusedArgs fa = {};
    fa.x = p->d_x; fa.y = p->d_y; fa.z = p->d_z; fa.u = p->d_u; fa.v = p->d_v;
    fa.obs_off = p->d_obs_off; fa.obs_slot = p->d_obs_slot;
    fa.n_obs = p->n_obs; fa.K = p->K; fa.PF = w->PF; fa.PRAW = f->PRAW; fa.n_pw = f->n_pw;
    fa.fcbuf = f->fcbuf; fa.mc_f = f->mc_f; fa.cost_f = f->cost_f;
    fa.huber_delta = p->huber_delta; fa.min_diag = min_diag; fa.max_diag = max_diag;
    fa.intr[0] = p->d_intr; fa.intr[1] = p->d_intr_c; fa.poses[0] = p->d_poses; fa.poses[1] = p->d_poses_c;
    fa.pf[0] = f->pf[0]; fa.pf[1] = f->pf[1]; fa.praw[0] = f->praw[0]; fa.praw[1] = f->praw[1];
    fa.dc = w->dc; fa.st = f->d_state; fa.st_flags = w->flags; fa.partial = f->partial; fa.red = f->red;
    fa.ticket = w->flags + 2;
    return fa;
}
// Per-frame elimination with four frames per wavefront (k_schur1m) when one pass covers the problem; sets fa.n_pw.
static bool fused_use_schur1m(const ccal_problem* p, FusedArgs& fa) {
    static const bool off = [] { const char* e = std::getenv("CCAL_SCHUR1M"); return e && e[0] == '0'; }();
    const int n_pw = (p->n_obs + 15) / 16 * 4;
    if (off || n_pw > p->nws->fws->n_pw) return false;
    fa.n_pw = n_pw;
    return true;
}
// register-resident Gram when the triangle of [J|r]^T[J|r] fits the VGPR/AGPR file next to the row math
// (measured at 10 000 frames, GN solve: KB4 0.58 vs 0.60 ms, OPENCV5 0.59 vs 0.55 ms); matrix-core Gram
// otherwise.  CCAL_GRAM=mfma|valu overrides.
static bool fused_use_valu_gram(const ccal_problem* p) {
    const int ncols = p->cams[0].D + 1;
    bool v = ncols * (ncols + 1) / 2 <= 120;      // UCM, EUCM, KB4; OPENCV5 (136 entries) is faster on the matrix cores
    if (const char* g = std::getenv("CCAL_GRAM")) v = (g[0] == 'v') && ncols * (ncols + 1) / 2 <= 136;
    return v;
}

// Host side of the device-resident loops: spin on the status word a decision kernel publishes to pinned memory.
static int wait_status(ccal_ctx* ctx, hipStream_t st, HostStatus* hst, const DevState* d_state, int target) {
    const auto tw = std::chrono::steady_clock::now();
    long spins = 0;
    while (hst->seq < target) {
        if ((++spins & 0xFFF) == 0) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
            if (el > 0.002 && hipStreamQuery(st) == hipSuccess && hst->seq < target) {
                // stream drained but the word did not arrive: fall back to an explicit copy
                DevState ds;
                HIP_TRY(ctx, hipMemcpy(&ds, d_state, sizeof ds, hipMemcpyDeviceToHost));
                hst->done = ds.done; hst->done_seq = ds.done_seq; hst->iter = ds.iter; hst->cur = ds.cur; hst->lm_accepted = ds.lm_accepted;
                hst->lm_rejected = ds.lm_rejected; hst->cur_cost = ds.cur_cost; hst->initial_cost = ds.initial_cost;
                hst->seq = target;
                break;
            }
            if (el > 30.0) { ctx->err = "device-resident solve timed out"; return CCAL_ERR_HIP; }
        }
    }
    return CCAL_OK;
}

// ---------------------------------------------------------------------------------------------
// Single-camera fused path: the host only enqueues (k_normal1, k_reduce1, k_head) groups, two
// iterations ahead, and watches a status word in pinned memory; accept/reject, damping, convergence
// tests and the camera solve run on the device (ccal_kernels_fused.hip).  Same decisions and the same
// arithmetic as the general loop in ccal_solve below.
// ---------------------------------------------------------------------------------------------
static int solve_fused(ccal_problem* p, const ccal_solver_opts* o, double* intr_io, double* poses_io, ccal_report* rep) {
    ccal_ctx* ctx = p->ctx;
    NormalWs* w = p->nws;
    int rc = fused_ws_ensure(p);
    if (rc != CCAL_OK) return rc;
    FusedWs* f = w->fws;
    hipStream_t st = ctx->stream;
    const bool lm = o->method == CCAL_METHOD_LM;
    const int K = p->K, K1 = K + 1;
    const auto t0 = std::chrono::steady_clock::now();

    DevState hs0;
    std::memset(&hs0, 0, sizeof hs0);
    hs0.radius = o->lm_initial_radius; hs0.dec = 2.0;
    hs0.lambda = lm ? 1.0 / o->lm_initial_radius : 0.0;
    hs0.min_error = o->min_error; hs0.min_abs = o->min_abs_error_decrease; hs0.min_rel = o->min_rel_error_decrease;
    hs0.cur = 0; hs0.first = 1; hs0.done = 0; hs0.iter = 0; hs0.max_iter = o->max_iterations; hs0.method = o->method;
    // one pinned staging block [intr | state | cols | poses] -> ONE async copy -> k_unpack1 (both parameter sets start
    // from the caller's values; slots without observations never change)
    static_assert(sizeof(ColInfo) % 8 == 0, "ColInfo is staged as doubles");
    if (f->tail_pending) { HIP_TRY(ctx, hipStreamSynchronize(st)); f->tail_pending = false; }   // stale k_head must not publish into this solve
    const size_t np6 = (size_t)p->n_slots * 6;
    const size_t small_doubles = CCAL_PMAX + sizeof(DevState) / sizeof(double) + CCAL_KMAX * sizeof(ColInfo) / sizeof(double);
    double* h_intr = f->h_stage;
    DevState* h_state = reinterpret_cast<DevState*>(h_intr + CCAL_PMAX);
    ColInfo* h_cols = reinterpret_cast<ColInfo*>(h_state + 1);
    double* h_poses = f->h_stage + small_doubles;
    std::memcpy(h_poses, poses_io, np6 * sizeof(double));
    std::memcpy(h_intr, intr_io, CCAL_PMAX * sizeof(double));
    *h_state = hs0;
    build_cols(p, h_cols);
    HIP_TRY(ctx, hipMemcpyAsync(f->d_stage, f->h_stage, (small_doubles + np6) * sizeof(double), hipMemcpyHostToDevice, st));
    {
        UnpackArgs ua = { f->d_stage, (int64_t)small_doubles, (int64_t)np6, p->d_intr, p->d_intr_c, p->d_poses, p->d_poses_c,
                          f->d_state, w->cols, w->flags };
        HIP_TRY(ctx, launch_unpack1(ua, st));
    }
    HostStatus* hst = f->h_status;
    hst->seq = 0; hst->done = 0; hst->done_seq = 0;

    FusedArgs fa = make_fused_args(p, o->lm_min_diagonal, o->lm_max_diagonal);
    // CCAL_FUSE_TAIL=1: the last Schur workgroup reduces and decides itself (two launches per GN iteration
    // instead of four).  Off by default: measured 10 us per iteration SLOWER than the split launches on
    // MI355X (10k frames 0.51 vs 0.46 ms, 1k frames 0.23 vs 0.20 ms for 3 GN iterations) - the ticket round
    // trip, the L2-bypassing reduction and the serial decision on one CU cost more than two launches do.
    const char* env_ft = std::getenv("CCAL_FUSE_TAIL");
    const char* env_tw = std::getenv("CCAL_FUSED_TAIL_WAVES");
    const bool fuse_tail = !p->allreduce && env_ft && env_ft[0] == '1';
    if (fuse_tail) fa.n_pw = std::min(f->n_pw, std::max(4, (env_tw ? std::atoi(env_tw) : 2048) / 4 * 4));
    const bool schur_m = !fuse_tail && fused_use_schur1m(p, fa);
    HeadArgs ha = {};
    ha.st = f->d_state; ha.hs = hst; ha.red = f->red; ha.cols = w->cols; ha.flags = w->flags;
    ha.intr[0] = p->d_intr; ha.intr[1] = p->d_intr_c; ha.dc = w->dc; ha.K = K;
    ha.min_diag = o->lm_min_diagonal; ha.max_diag = o->lm_max_diagonal;
    const int model = p->cams[0].model;
    const bool use_valu_gram = fused_use_valu_gram(p);
    int seq = 0;
    auto enqueue = [&]() -> int {         // one evaluation + decision + solve; returns the seq that marks its end
        if (use_valu_gram) HIP_TRYN(ctx, launch_gram1v(model, p->one_focal, fa, st));
        else HIP_TRYN(ctx, launch_gram1(model, p->one_focal, fa, st));
        if (!lm) {
            if (fuse_tail) {
                ha.phase = 3; ha.seq = ++seq;
                HIP_TRYN(ctx, launch_schur1(fa, 0, &ha, st));
            } else {
                if (schur_m) HIP_TRYN(ctx, launch_schur1m(fa, 0, st));
                else HIP_TRYN(ctx, launch_schur1(fa, 0, nullptr, st));
                HIP_TRYN(ctx, launch_reduce1(fa, 0, 2 * K1 * K1, st));
                if (p->allreduce && p->allreduce(p->allreduce_user, f->red, (size_t)f->RB1, (void*)st) != 0) { ctx->err = "all-reduce callback failed"; return -CCAL_ERR_HIP; }
                ha.phase = 3; ha.seq = ++seq;
                HIP_TRYN(ctx, launch_head(ha, st));
            }
        } else {
            // sharded LM: two small all-reduces per group - [cost, model decrease of the pose blocks] before the
            // decision, [A_dir | Y^T Y] before the camera solve; the decisions are then identical on every rank
            HIP_TRYN(ctx, launch_cost1(fa, st));
            if (p->allreduce && p->allreduce(p->allreduce_user, f->red + 2 * K1 * K1, 2, (void*)st) != 0) { ctx->err = "all-reduce callback failed"; return -CCAL_ERR_HIP; }
            ha.phase = 1 | 4; ha.seq = ++seq;       // decide; the group's status is published by its last step
            HIP_TRYN(ctx, launch_head(ha, st));
            ha.phase = 2; ha.seq = ++seq;
            if (fuse_tail) {
                HIP_TRYN(ctx, launch_schur1(fa, 1, &ha, st));
            } else {
                if (schur_m) HIP_TRYN(ctx, launch_schur1m(fa, 1, st));
                else HIP_TRYN(ctx, launch_schur1(fa, 1, nullptr, st));
                HIP_TRYN(ctx, launch_reduce1(fa, 0, 2 * K1 * K1, st));
                if (p->allreduce && p->allreduce(p->allreduce_user, f->red, (size_t)(2 * K1 * K1), (void*)st) != 0) { ctx->err = "all-reduce callback failed"; return -CCAL_ERR_HIP; }
                HIP_TRYN(ctx, launch_head(ha, st));
            }
        }
        return seq;
    };
    auto wait_seq = [&](int target) -> int { return wait_status(ctx, st, hst, f->d_state, target); };

    // keep two groups in flight: the GPU never waits for the host, the host wastes at most two
    // early-exit groups after convergence
    std::vector<int> pending;
    int enq = 0;
    const int max_groups = o->max_iterations + 1;
    int status = CCAL_OK;
    bool finished = false;
    while (!finished) {
        // Sharded solves (all-reduce hook set) must issue the SAME sequence of collectives on every rank.  They do:
        // the fill rule below is a function of the group that reported `done` only (groups enqueued = that index +
        // depth, whatever the host timing), and `done` is decided from all-reduced sums, identically on every rank.
        // Default with a hook: no group ahead (CCAL_FUSED_DEPTH_HOOK=2 enqueues one; on a 1-rank RCCL group the extra
        // early-exit group with its hook calls cost as much as the overlap gained: 0.46 vs 0.42 ms LM at 1 000 frames).
        static const int env_depth = [] { const char* e = std::getenv("CCAL_FUSED_DEPTH"); return e ? std::max(1, std::atoi(e)) : 2; }();
        static const int env_depth_hook = [] { const char* e = std::getenv("CCAL_FUSED_DEPTH_HOOK"); return e ? std::max(1, std::atoi(e)) : 1; }();
        const int depth = p->allreduce ? env_depth_hook : env_depth;
        while ((int)pending.size() < depth && enq < max_groups) {
            const int s = enqueue();
            if (s < 0) return -s;
            pending.push_back(s); ++enq;
        }
        if (pending.empty()) break;
        const int waited = pending.front();
        rc = wait_seq(waited);
        if (rc != CCAL_OK) return rc;
        pending.erase(pending.begin());
        // act on `done` only when it was set by a step this thread has waited for: a later group may already have
        // published it, and how many groups get enqueued must not depend on that race (sharded ranks would issue
        // different numbers of collectives)
        if (hst->done && hst->done_seq <= waited) { status = hst->done - 1; finished = true; }
        else if (o->verbose) std::printf("[ccal fused %s] iter %d cost %.12g\n", lm ? "LM" : "GN", hst->iter, hst->cur_cost);
    }
    // hst->done was published by the last instruction of the deciding k_head (after a system-scope fence): everything
    // the result depends on is complete.  The download goes through a side stream so that it does not queue behind
    // the (at most two) early-exit groups still in the main stream; the next solve drains those before it starts.
    hipStream_t dl = st;
    if (hst->done && !pending.empty()) { dl = f->side; f->tail_pending = true; }
    else HIP_TRY(ctx, hipStreamSynchronize(st));
    struct { int done, iter, cur, acc, rej; double cur_cost, initial_cost; } ds =
        { hst->done, hst->iter, hst->cur, hst->lm_accepted, hst->lm_rejected, hst->cur_cost, hst->initial_cost };
    if (!ds.done) { status = CCAL_ERR_NO_CONVERGENCE; }
    else status = ds.done - 1;
    if (ds.cur == 1) { std::swap(p->d_intr, p->d_intr_c); std::swap(p->d_poses, p->d_poses_c); }
    ccal_report R = {};
    R.status = status; R.iterations = ds.iter; R.lm_accepted = ds.acc; R.lm_rejected = ds.rej;
    R.initial_cost = ds.initial_cost; R.final_cost = ds.cur_cost;
    if (np6) HIP_TRY(ctx, hipMemcpyAsync(h_poses, p->d_poses, np6 * sizeof(double), hipMemcpyDeviceToHost, dl));
    HIP_TRY(ctx, hipMemcpyAsync(h_intr, p->d_intr, CCAL_PMAX * sizeof(double), hipMemcpyDeviceToHost, dl));
    HIP_TRY(ctx, hipStreamSynchronize(dl));
    std::memcpy(poses_io, h_poses, np6 * sizeof(double));
    std::memcpy(intr_io, h_intr, CCAL_PMAX * sizeof(double));
    if (p->one_focal) intr_io[1] = intr_io[0];           // fy = f (src/util.rs:467-470)
    rc = CCAL_OK;
    R.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep) *rep = R;
    if (status == CCAL_ERR_NOT_PD) ctx->err = "normal equations are not positive definite";
    if (rc != CCAL_OK) return rc;
    return status;
}

extern "C" {

// developer hook (not part of include/ccal.h): copy the per-frame scratch of the fast path to the host;
// diagnostic builds (tools/) park in-kernel timestamps there
int ccal_debug_fcbuf(ccal_problem* p, double* out, int64_t n) {
    if (!p || !p->nws || !p->nws->fws || !out) return CCAL_ERR_INVALID_ARG;
    const int64_t m = std::min<int64_t>(n, (int64_t)std::max(p->n_obs, 1) * 40);
    if (hipMemcpy(out, p->nws->fws->fcbuf, m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return CCAL_ERR_HIP;
    return CCAL_OK;
}

int ccal_build_normal_dev(ccal_problem* p, double lambda) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    int rc = normal_ws_ensure(p);
    if (rc != CCAL_OK) return rc;
    NormalWs* w = p->nws;
    w->red_fused = false;
    if (p->n_cams == 1 && p->n_obs > 0 && !p->allreduce && !std::getenv("CCAL_DISABLE_FUSED")) {
        // single camera: the device loop's own kernels (register / LDS Gram + per-frame elimination), evaluated at the
        // current parameters with the state set to "first evaluation"; fws->red = [A_dir | Y^T Y | . | .]
        ccal_ctx* ctx = p->ctx;
        if ((rc = fused_ws_ensure(p)) != CCAL_OK) return rc;
        FusedWs* f = w->fws;
        hipStream_t st = ctx->stream;
        FusedArgs fa = make_fused_args(p, 1e-6, 1e32);
        const bool schur_m = fused_use_schur1m(p, fa);
        HIP_TRY(ctx, launch_state_eval(f->d_state, lambda, st));
        if (fused_use_valu_gram(p)) HIP_TRY(ctx, launch_gram1v(p->cams[0].model, p->one_focal, fa, st));
        else HIP_TRY(ctx, launch_gram1(p->cams[0].model, p->one_focal, fa, st));
        if (schur_m) HIP_TRY(ctx, launch_schur1m(fa, 0, st));
        else HIP_TRY(ctx, launch_schur1(fa, 0, nullptr, st));
        HIP_TRY(ctx, launch_reduce1(fa, 0, 2 * (p->K + 1) * (p->K + 1), st));
        w->red_fused = true;
        return CCAL_OK;
    }
    if ((rc = enqueue_gram(p, false, w->cur)) != CCAL_OK) return rc;
    return enqueue_reduce_system(p, w->cur, lambda, 1e-6, 1e32);
}

int ccal_build_normal(ccal_problem* p, const double* intr, const double* poses, const double* extr,
                      double lambda, double* S, double* b, double* cost) {
    if (!p || !intr || (!poses && p->n_slots)) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    int rc = ccal_upload_params(p, intr, poses, extr);
    if (rc != CCAL_OK) return rc;
    if ((rc = normal_ws_ensure(p)) != CCAL_OK) return rc;
    NormalWs* w = p->nws;
    HIP_TRY(ctx, hipMemsetAsync(w->flags, 0, 4 * sizeof(int32_t), ctx->stream));
    if ((rc = ccal_build_normal_dev(p, lambda)) != CCAL_OK) return rc;
    const int K = w->K, K1 = K + 1;
    int32_t flags[4];
    if (w->red_fused) {
        // [A_dir | Y^T Y]: S = A_dir - Y^T Y on the camera block, b its last column, cost = the r x r corner of A_dir
        double* h = w->fws->h_stage;
        HIP_TRY(ctx, hipMemcpyAsync(h, w->fws->red, 2 * K1 * K1 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(flags, w->flags, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double* A = h; const double* Y = h + K1 * K1;
        if (S) for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
            double v = A[i * K1 + j] - Y[i * K1 + j];
            if (i == j && lambda > 0.0) v += lambda * std::min(std::max(A[i * K1 + i], 1e-6), 1e32);
            S[i * K + j] = v;
        }
        if (b) for (int i = 0; i < K; ++i) b[i] = A[i * K1 + K] - Y[i * K1 + K];
        if (cost) *cost = A[K * K1 + K];
    } else {
        double* h = w->h_pinned;
        HIP_TRY(ctx, hipMemcpyAsync(h, w->red, w->RB * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(flags, w->flags, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double* hd = h + K1 * K1;
        if (S) for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
            double v = h[i * K1 + j];
            if (i == j && lambda > 0.0) v += lambda * std::min(std::max(hd[i], 1e-6), 1e32);
            S[i * K + j] = v;
        }
        if (b) for (int i = 0; i < K; ++i) b[i] = h[i * K1 + K];
        if (cost) *cost = h[w->RB - 1];
    }
    if (flags[0]) { ctx->err = "a frame's pose block is not positive definite"; return CCAL_ERR_NOT_PD; }
    return CCAL_OK;
}

int ccal_solve(ccal_problem* p, const ccal_solver_opts* o, double* intr_io, double* poses_io, double* extr_io, ccal_report* rep) {
    if (!p || !o || !intr_io || (!poses_io && p->n_slots)) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = normal_ws_ensure(p);
    if (rc != CCAL_OK) return rc;
    const bool lm = o->method == CCAL_METHOD_LM;
    // single camera: its own device-resident loop (GN and LM, sharded or not)
    if (p->n_cams == 1 && p->n_obs > 0 && !std::getenv("CCAL_DISABLE_FUSED"))
        return solve_fused(p, o, intr_io, poses_io, rep);
    // General loop (several cameras, sharded LM, or CCAL_DISABLE_FUSED): device-resident as well.  One group =
    //   k_schur -> k_reduce -> (all-reduce red) -> k_solve -> k_backsub -> k_gram at the candidate (per camera) -> k_sum2
    //   -> (all-reduce cost, model decrease) -> k_gdecide
    // every kernel picks the current parameter / Gram set by st->cur and its damping from st->lambda, k_gdecide applies
    // the oracle's accept / reject / stop rules and publishes a status word; the host only enqueues groups (one ahead
    // of the group it waits for) and polls.  Set 0 = (p->d_*, G[w->cur]), set 1 = (p->d_*_c, G[w->cur ^ 1]).
    NormalWs* w = p->nws;
    hipStream_t st = ctx->stream;
    if (w->tail_pending) { HIP_TRY(ctx, hipStreamSynchronize(st)); w->tail_pending = false; }   // a stale k_gdecide must not publish into this solve
    if ((rc = ccal_upload_params(p, intr_io, poses_io, extr_io)) != CCAL_OK) return rc;
    if ((rc = normal_upload_cols(p)) != CCAL_OK) return rc;
    const double min_d = o->lm_min_diagonal, max_d = o->lm_max_diagonal;
    const auto t0 = std::chrono::steady_clock::now();
    DevState* hs0 = w->h_gstate;
    std::memset(hs0, 0, sizeof *hs0);
    hs0->radius = o->lm_initial_radius; hs0->dec = 2.0;
    hs0->lambda = lm ? 1.0 / o->lm_initial_radius : 0.0;
    hs0->min_error = o->min_error; hs0->min_abs = o->min_abs_error_decrease; hs0->min_rel = o->min_rel_error_decrease;
    hs0->max_iter = o->max_iterations; hs0->method = o->method;
    HIP_TRY(ctx, hipMemcpyAsync(w->d_gstate, hs0, sizeof(DevState), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(w->flags, 0, 4 * sizeof(int32_t), st));
    HostStatus* hst = w->h_gstatus;
    hst->seq = 0; hst->done = 0; hst->done_seq = 0;
    const DevState* ds = w->d_gstate;
    int seq = 0;
    auto ar = [&](double* buf, size_t n) -> int {
        if (p->allreduce && p->allreduce(p->allreduce_user, buf, n, (void*)st) != 0) { ctx->err = "all-reduce callback failed"; return CCAL_ERR_HIP; }
        return CCAL_OK;
    };
    auto enqueue = [&](bool init) -> int {          // returns the sequence number that marks the group's end, < 0 on error
        if (!init) {
            HIP_TRYN(ctx, launch_schur(p, w->cur, 0.0, min_d, max_d, st, ds));
            HIP_TRYN(ctx, launch_reduce(p, st, ds));
            if (ar(w->red, (size_t)w->RB) != CCAL_OK) return -CCAL_ERR_HIP;
            HIP_TRYN(ctx, launch_solve(p, 0.0, min_d, max_d, st, ds));
            HIP_TRYN(ctx, launch_backsub(p, 0.0, min_d, max_d, st, ds));
        }
        for (int c = 0; c < p->n_cams; ++c) HIP_TRYN(ctx, launch_gram_dev(p, c, ds, init ? 0 : 1, st));
        if (!p->allreduce) {         // sums and decision in one launch
            HIP_TRYN(ctx, launch_sum_cost_dev(p, w->d_gstate, init ? 0 : 1, hst, init, ++seq, st));
        } else {
            HIP_TRYN(ctx, launch_sum_cost_dev(p, w->d_gstate, init ? 0 : 1, nullptr, init, 0, st));
            if (ar(w->scal, 2) != CCAL_OK) return -CCAL_ERR_HIP;
            HIP_TRYN(ctx, launch_gdecide(p, w->d_gstate, hst, init, ++seq, st));
        }
        return seq;
    };
    std::vector<int> pending;
    int enq = 0;
    const int max_groups = o->max_iterations + 1;
    // sharded solves: every rank issues the same sequence of collectives - the decisions come from all-reduced sums and
    // the number of groups enqueued depends only on the group that reported `done` (that index + depth)
    static const int env_gdepth = [] { const char* e = std::getenv("CCAL_GENERAL_DEPTH"); return e ? std::max(1, std::atoi(e)) : 2; }();
    const int depth = p->allreduce ? 1 : env_gdepth;
    bool finished = false;
    while (!finished) {
        while ((int)pending.size() < depth && enq < max_groups) {
            const int sq = enqueue(enq == 0);
            if (sq < 0) return -sq;
            pending.push_back(sq); ++enq;
        }
        if (pending.empty()) break;
        const int waited = pending.front();
        if ((rc = wait_status(ctx, st, hst, w->d_gstate, waited)) != CCAL_OK) return rc;
        pending.erase(pending.begin());
        if (o->verbose) std::printf("[ccal %s] iter %d cost %.12g radius %.3g\n", lm ? "LM" : "GN", hst->iter, hst->cur_cost, hst->radius);
        if (hst->done && hst->done_seq <= waited) finished = true;     // see solve_fused: no dependence on publication races
    }
    // the deciding k_gdecide published after a system-scope fence: the result is complete; it is downloaded through a
    // side stream so that it does not queue behind the early-exit group enqueued ahead (drained before the next solve)
    hipStream_t dl = st;
    if (hst->done && !pending.empty()) { dl = w->side; w->tail_pending = true; }
    else HIP_TRY(ctx, hipStreamSynchronize(st));
    int status = hst->done ? hst->done - 1 : CCAL_ERR_NO_CONVERGENCE;
    if (hst->cur == 1) {             // the accepted point lives in set 1: make it set 0 for whoever comes next
        std::swap(p->d_intr, p->d_intr_c); std::swap(p->d_poses, p->d_poses_c); std::swap(p->d_extr, p->d_extr_c);
        w->cur ^= 1;
    }
    ccal_report R = {};
    R.status = status; R.iterations = hst->iter; R.lm_accepted = hst->lm_accepted; R.lm_rejected = hst->lm_rejected;
    R.initial_cost = hst->initial_cost; R.final_cost = hst->cur_cost;
    HIP_TRY(ctx, hipMemcpyAsync(intr_io, p->d_intr, sizeof(double) * p->n_cams * CCAL_PMAX, hipMemcpyDeviceToHost, dl));
    if (poses_io && p->n_slots) HIP_TRY(ctx, hipMemcpyAsync(poses_io, p->d_poses, sizeof(double) * p->n_slots * 6, hipMemcpyDeviceToHost, dl));
    if (extr_io) HIP_TRY(ctx, hipMemcpyAsync(extr_io, p->d_extr, sizeof(double) * p->n_cams * 6, hipMemcpyDeviceToHost, dl));
    HIP_TRY(ctx, hipStreamSynchronize(dl));
    if (p->one_focal) for (int c = 0; c < p->n_cams; ++c) intr_io[c * CCAL_PMAX + 1] = intr_io[c * CCAL_PMAX];   // fy = f (src/util.rs:467-470)
    rc = CCAL_OK;
    R.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep) *rep = R;
    if (status == CCAL_ERR_NOT_PD) ctx->err = "normal equations are not positive definite";
    if (rc != CCAL_OK) return rc;
    return status;
}

}  // extern "C"
