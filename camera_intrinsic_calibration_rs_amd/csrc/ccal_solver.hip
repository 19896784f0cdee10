// Mode N host side: workspaces, ccal_build_normal and the optimizer loop (ccal_solve).
//
// The loop restates tiny_solver::GaussNewtonOptimizer::optimize as the reference calls it
// (src/util.rs:443-463, 668-670; defaults in ccal_set_defaults) -- per iteration: Jacobian at x,
// solve the normal equations, x <- clamp(x + dx), error(x), stop on min_error / |d error| thresholds --
// and adds a Ceres-style Levenberg-Marquardt mode on the same kernels.  Both loops (single camera: solve_fused,
// everything else: solve_general) are device-resident: the decisions run in a kernel, the host
// enqueues groups of launches one ahead and polls a status word in pinned memory.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include <cstdlib>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "ccal_fused.hpp"
#include "ccal_devopt.hpp"

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
    // early-exit groups of the last solve (or of a solve that ended in an error: every such exit sets tail_pending) may
    // still be queued: they publish into the pinned status words freed below.  The whole device is drained rather than
    // the stream: the stream may be the caller's own (ccal_ctx_create with a stream) and already destroyed when a
    // binding's garbage collector gets here.
    if (w->tail_pending || (w->fws && w->fws->tail_pending)) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipDeviceSynchronize();
    }
    void* ptrs[] = { w->G[0], w->G[1], w->cost_o[0], w->cost_o[1], w->d_goff, w->d_slot_off, w->d_slot_obs, w->d_obs_cam,
                     w->d_caminfo, w->partial, w->red, w->pf, w->dc, w->mc_slot, w->scal, w->flags, w->cols, w->d_slot_desc, w->d_slot_rec, w->d_all_obs, w->d_obs_owner };
    // (the blocks go back to the context's cache - ccal_internal.hpp - for its next problem; ccal_problem_destroy has drained the stream)
    ccal_ctx* ctx = p->ctx;
    for (void* q : ptrs) if (q) ctx_release(ctx, q, false);
    for (int32_t* q : w->d_gen_sorted) if (q) ctx_release(ctx, q, false);
    if (w->h_pinned) (void)hipHostFree(w->h_pinned);
    if (w->d_gstate) ctx_release(ctx, w->d_gstate, false);
    if (w->side) { (void)hipStreamSynchronize(w->side); ctx_stream_put(ctx, w->side); }
    if (w->h_gstatus) ctx_release(ctx, w->h_gstatus, true);
    if (w->h_gstate) (void)hipHostFree(w->h_gstate);
    if (FusedWs* f = w->fws) {
        if (f->side) { (void)hipStreamSynchronize(f->side); ctx_stream_put(ctx, f->side); }      // (the result's download ran there)
        if (f->d_block) ctx_release(ctx, f->d_block, false);          // every device buffer of the workspace is a slice of it
        if (f->h_block) ctx_release(ctx, f->h_block, true);           // h_status | h_result | h_stage
        if (f->fcbuf) (void)hipFree(f->fcbuf);
        delete f;
    }
    delete w;
    p->nws = nullptr;
}

int drain_pending_groups(ccal_problem* p) {
    NormalWs* w = p->nws;
    if (!w || !(w->tail_pending || (w->fws && w->fws->tail_pending))) return CCAL_OK;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    w->tail_pending = false;
    if (w->fws) w->fws->tail_pending = false;
    return CCAL_OK;
}

// Session-sized solves (poses up to this many bytes: 2 048 frames) move their parameters without a copy operation: k_unpack1
// reads the caller's poses from pinned host memory, the k_head that finishes the solve writes the result there.  Larger
// problems stage through device copies (a kernel reading / writing half a megabyte across the bus one wavefront wide would
// cost more than the DMA it saves).
constexpr size_t kZeroCopyBytes = 96 * 1024;
// poses a finishing single-launch group writes to the host with ALL its workgroups: up to 5 461 frames.  Stores of a shader cross the bus in
// 64-byte packets, the DMA engine's in 256+: 2 500 frames GN 0.150 -> 0.136 ms, 5 000: 0.184 -> 0.178, but 10 000: 0.238 -> 0.255 and, from a
// table sorted by corner count, scattered pose READS over the bus on top (0.219 -> 0.256): profiles/r06/ab_result_spread.txt
constexpr size_t kSpreadBytes = 256 * 1024;

// rows of the single-camera loop's partial-sum buffer (fused_ws_ensure)
static int fused_partial_rows(int n_obs) {
    const int n_pw = std::max(std::min(std::max(n_obs, 1), 16384), (std::max(n_obs, 1) + 9) / 10 + 8);
    return (n_pw + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
}
static int fused_ws_ensure(ccal_problem* p) {
    NormalWs* w = p->nws;
    if (w->fws) return CCAL_OK;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FusedWs* f = new FusedWs();
    w->fws = f;
    f->PRAW = praw_size(p->K);
#ifdef CCAL_LEGACY_KERNELS
    { const char* e = dev_env("CCAL_FUSE_ELIM"); f->fuse_elim = !(e && e[0] == '0'); }      // second library: the separate elimination launch (k_schur1m)
#endif
    f->RB1 = fused_red_size(p->K);
    // rows of the partial-sum buffer = most wavefronts a Gram launch of this problem may have: one row per wavefront (fused
    // elimination).  Up to 16 384 frames any lanes-per-frame mapping fits; beyond, the launchers keep to mappings that do - six
    // lanes per frame (ten frames per wavefront) always does
    f->n_pw = fused_partial_rows(p->n_obs);
    const size_t ns = (size_t)std::max(p->n_slots, 1), no = (size_t)std::max(p->n_obs, 1);
    // ONE device allocation and ONE pinned allocation, sliced (a calibration session creates a problem and solves it once or
    // twice: fifteen hipMalloc / hipHostMalloc calls and five memsets were 0.46 ms of the first solve's 0.58 at 600 frames)
    const size_t stage_bytes = std::max((ns * 6 + CCAL_PMAX) * sizeof(double) + 64, (size_t)(f->RB1 + 8) * sizeof(double));      // (ccal_build_normal stages the reduced sums here)
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_pf = up(ns * w->PF * sizeof(double)), b_praw = up(no * f->PRAW * sizeof(double)), b_no = up(no * sizeof(double));
    const size_t b_part = up((size_t)f->RB1 * f->n_pw * sizeof(double)), b_red = up((size_t)(f->RB1 + 7) * sizeof(double));
    const size_t b_state = up(3 * sizeof(DevState)), b_stage = up(stage_bytes);
    const size_t b_cnt = 256;
    const size_t zeroed = 2 * b_pf + 2 * b_praw + b_part + 2 * b_red + b_cnt;              // the slices that must start as zeros come first (red: two buffers,
                                                                                  // the in-process transport alternates between them)
    const size_t d_total = zeroed + 2 * b_no + b_state + b_stage;
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&f->d_block, d_total));
    HIP_TRY(ctx, hipMemsetAsync(f->d_block, 0, zeroed, ctx->stream));             // (stream-ordered in front of everything that uses the workspace)
    {
        char* q = f->d_block;
        for (int i = 0; i < 2; ++i) { f->pf[i] = reinterpret_cast<double*>(q); q += b_pf; }
        for (int i = 0; i < 2; ++i) { f->praw[i] = reinterpret_cast<double*>(q); q += b_praw; }
        f->partial = reinterpret_cast<double*>(q); q += b_part;
        f->red = reinterpret_cast<double*>(q); q += 2 * b_red; f->red_stride = b_red / sizeof(double);
        f->done_cnt = reinterpret_cast<int32_t*>(q); q += b_cnt;
        f->mc_f = reinterpret_cast<double*>(q); q += b_no;
        f->cost_f = reinterpret_cast<double*>(q); q += b_no;
        f->d_state = reinterpret_cast<DevState*>(q); q += b_state;      // [0] the loops' state; [1], [2]: single-launch groups alternate
        f->d_stage = reinterpret_cast<double*>(q); q += b_stage;
    }
    // per-frame scratch of diagnostic builds (-DCCAL_STAMPS: in-kernel timestamps, tools/stamps_*.py); the product allocates nothing
#ifdef CCAL_STAMPS
    HIP_TRY(ctx, hipMalloc((void**)&f->fcbuf, std::max<size_t>(no * 40, 32768) * sizeof(double)));
#endif
    {
        const bool zc = ns * 6 * sizeof(double) <= kSpreadBytes;         // (beyond kZeroCopyBytes: single-launch groups only, FusedJob::begin)
        const size_t b_hs = up(sizeof(HostStatus)), b_res = zc ? up((ns * 6 + CCAL_PMAX) * sizeof(double)) : 0;
        HIP_TRY(ctx, ctx_host_alloc(ctx, (void**)&f->h_block, b_hs + b_res + b_stage));
        char* q = f->h_block;
        f->h_status = reinterpret_cast<HostStatus*>(q); q += b_hs;
        f->h_result = zc ? reinterpret_cast<double*>(q) : nullptr; q += b_res;
        f->h_stage = reinterpret_cast<double*>(q);
    }
    HIP_TRY(ctx, ctx_stream_get(ctx, &f->side));
    std::memset((void*)f->h_status, 0, sizeof(HostStatus));
    return CCAL_OK;
}

template <class T>
static int dev_upload(ccal_ctx* ctx, T** dst, const std::vector<T>& src) {
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) HIP_TRY(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return CCAL_OK;
}

// The part every loop needs: sizes, the column table and the camera step.  The single-camera loop (FusedWs) needs nothing else of
// NormalWs; the general loop's buffers - per-frame record sets, slot tables, partial sums, a second status block and stream:
// ~25 allocations, seven synchronous uploads - are made when a general-loop entry first asks (normal_ws_ensure_general): they
// were 0.4 ms of a single-camera session's first solve.
int normal_ws_ensure(ccal_problem* p) {
    if (p->nws) return CCAL_OK;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    NormalWs* w = new NormalWs();
    p->nws = w;
    w->K = p->K; w->RB = red_size(p->K); w->PF = pf_size(p->K);
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->dc, CCAL_KMAX * sizeof(double)));
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->cols, CCAL_KMAX * sizeof(ColInfo)));
    return CCAL_OK;
}
int normal_ws_ensure_general(ccal_problem* p) {
    int rc0 = normal_ws_ensure(p);
    if (rc0 != CCAL_OK) return rc0;
    NormalWs* w = p->nws;
    if (w->general_ready) return CCAL_OK;
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // persistent Schur waves: at most 2 workgroups per CU worth, never more than slots
    // persistent wavefronts of k_schur: 4 per SIMD (measured at 10 000 slots x 2 cameras: 2048 -> 165.7, 4096 -> 158.5, 8192 -> 173 us per build)
    int n_pw = std::min(std::max(p->n_slots, 1), std::max(4, dev_env_int("CCAL_SCHUR_WAVES", 4096)));
    n_pw = (n_pw + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK * WAVES_PER_BLOCK;
    // 64 .. 127 columns (five and more cameras): the (K + 1)^2 accumulators of a wavefront take up to 131 KB of LDS - one
    // wavefront per workgroup and CU, 512 of them (each writes its own row of partial sums: 512 x RB doubles)
    // (also below 64 columns when four wavefronts' images do not fit the CU's 160 KB: K + 1 = 62 .. 64 with OPENCV5-sized records)
    {
        int stg = 0;
        for (int c = 0; c < p->n_cams; ++c) stg = std::max(stg, gen_e_off(p->cams[c].Peff) + 144);       // as launch_schur
        const int K1 = p->K + 1;
        const size_t ws4 = sizeof(double) * WAVES_PER_BLOCK * (size_t)((((w->RB + 12 * K1 + 36 + 1) & ~1)) + stg);
        if (p->K >= 64 || ws4 + 6 * 1024 > 160 * 1024) { w->schur_wpb = 1; n_pw = std::min(std::max(p->n_slots, 1), 512); }
    }
    w->n_pw = n_pw;
    std::vector<int64_t> goff(p->n_obs);
    std::vector<int32_t> caminfo(p->n_cams * 4);
    int64_t gl = 0;
    // Register Gram kernels + record format for every camera (k_gram1v / k_gram1w, GEN; k_schur expands the records).
    // CCAL_GENERAL_GRAM=mfma: the matrix-core kernel k_gram with its 16 x 16 / 32-stride tiles (19-column other-camera
    // blocks), kept as the independent second implementation the tests compare against.
#ifdef CCAL_LEGACY_KERNELS
    { const char* e = dev_env("CCAL_GENERAL_GRAM"); w->register_gram = !(e && e[0] == 'm') || w->schur_wpb == 1; }      // the matrix-core pair: four-wavefront elimination only
#else
    w->register_gram = true;               // (the matrix-core pair lives in libccal_hip_legacy.so only)
#endif
    // two cameras with equal blocks: k_schurq from 1 000 slots (round 4, with its eight-lanes-per-slot form up to 8 192 slots; whole
    // build against the generic k_schur<true>: 1 000 slots 27.5 against 28.2 us, 3 000: 37.6 / 42.6, 6 000: 56.1 / 63.8, 10 000: 71.7
    // with k_schur at 81); below that the generic kernel's many short wavefronts win.  CCAL_SCHURQ=1 / 0 forces it / the generic
    // k_schur<true>, which every other rig takes
    {
        int pe[CCAL_MAX_CAMS], ct[CCAL_MAX_CAMS], ce[CCAL_MAX_CAMS];
        for (int c = 0; c < p->n_cams; ++c) { pe[c] = p->cams[c].Peff; ct[c] = p->cams[c].col_theta; ce[c] = p->cams[c].col_extr; }
        const char* e = dev_env("CCAL_SCHURQ");
        w->schurq = w->register_gram && schurq_fits(p->n_cams, pe, ct, ce) && (e ? e[0] != '0' : p->n_slots >= 1000);
    }
    w->schurq_slots = schurq_slots_per_wave(p->n_slots);
    { const int v = dev_env_int("CCAL_SCHURQ_SLOTS", 0); if (v == 8 || v == 16) w->schurq_slots = v; }
    w->n_rows = w->schurq ? schurq_rows(p->n_slots, w->schurq_slots) : n_pw / w->schur_wpb;
    // cameras of one model (and the problem's one focal mode): their blocks go through ONE launch (CCAL_MERGE_GRAM=0: one per camera)
    {
        bool same = p->n_cams > 1;
        for (int c = 1; c < p->n_cams; ++c) same = same && p->cams[c].model == p->cams[0].model && p->cams[c].Peff == p->cams[0].Peff;
        const char* e = dev_env("CCAL_MERGE_GRAM");
        w->merged_gram = w->register_gram && same && !(e && e[0] == '0');
    }
    std::vector<int> ncp_of(p->n_cams);
    for (int c = 0; c < p->n_cams; ++c) {
        ncp_of[c] = (p->cams[c].D + 1) <= 16 ? 16 : 32;
        caminfo[c * 4 + 0] = p->cams[c].Peff; caminfo[c * 4 + 1] = p->cams[c].col_theta;
        caminfo[c * 4 + 2] = p->cams[c].col_extr; caminfo[c * 4 + 3] = w->register_gram ? 0 : ncp_of[c];
    }
    if (w->schurq) {
        // k_schurq: a slot's two records side by side at (2 slot + camera) x record size - the kernel computes the addresses and
        // requests the records with its first instructions instead of after a look-up; a missing record is a hole that stays
        // zero (the buffers are cleared below and nothing ever writes there)
        const int64_t rs = gen_rec_size(p->cams[0].Peff);
        for (int o = 0; o < p->n_obs; ++o) goff[o] = (2 * (int64_t)p->h_obs_slot[o] + p->h_obs_cam[o]) * rs;
        gl = 2 * (int64_t)std::max(p->n_slots, 1) * rs;
    } else {
        for (int o = 0; o < p->n_obs; ++o) {
            goff[o] = gl;
            const int c = p->h_obs_cam[o];
            gl += w->register_gram ? (int64_t)gen_rec_size(p->cams[c].Peff) : (int64_t)ncp_of[c] * ncp_of[c];
        }
    }
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
    {
        std::vector<int8_t> owner(std::max(p->n_obs, 1), 0);
        w->all_slots_observed = true;
        for (int sl = 0; sl < p->n_slots; ++sl) {
            if (slot_off[sl + 1] > slot_off[sl]) owner[slot_obs[slot_off[sl]]] = 1;
            else w->all_slots_observed = false;
        }
        if ((rc = dev_upload(ctx, &w->d_obs_owner, owner))) return rc;
    }
    if (w->merged_gram) {
        std::vector<int32_t> all;
        all.reserve(p->n_obs);
        for (int c = 0; c < p->n_cams; ++c) all.insert(all.end(), p->cams[c].obs.begin(), p->cams[c].obs.end());
        if ((rc = dev_upload(ctx, &w->d_all_obs, all))) return rc;
    }
    if (w->register_gram) {
        // ragged frames: every Gram launch's list sorted by corner count, with the bins of the launch (the single-camera loop's plan,
        // ccal_kernels_gram2.hip: gram2_bin_plan - uniform frames and short lists get none)
        auto plan_list = [&](const std::vector<int32_t>& list, int slot, int model) -> int {
            if (list.size() < 2000) return CCAL_OK;
            std::vector<int64_t> off(list.size() + 1, 0);
            for (size_t i = 0; i < list.size(); ++i) off[i + 1] = off[i] + (p->h_obs_off[list[i] + 1] - p->h_obs_off[list[i]]);
            std::vector<int32_t> order;
            const GramBins gb = gram2_bin_plan(off.data(), (int)list.size(), model == kUCM || model == kEUCM, &order, true);
            if (gb.n_bins <= 0) return CCAL_OK;
            std::vector<int32_t> sorted(list.size());
            for (size_t i = 0; i < list.size(); ++i) sorted[i] = list[(size_t)order[i]];
            if (int r = dev_upload(ctx, &w->d_gen_sorted[slot], sorted)) return r;
            w->gen_bins[slot] = gb;
            return CCAL_OK;
        };
        if (w->merged_gram) {
            std::vector<int32_t> all;
            for (int c = 0; c < p->n_cams; ++c) all.insert(all.end(), p->cams[c].obs.begin(), p->cams[c].obs.end());
            if ((rc = plan_list(all, 0, p->cams[0].model))) return rc;
        } else {
            for (int c = 0; c < p->n_cams; ++c) if ((rc = plan_list(p->cams[c].obs, 1 + c, p->cams[c].model))) return rc;
        }
    }
    if (w->schurq) {
        std::vector<int64_t> slot_rec((size_t)std::max(p->n_slots, 1) * 2, -1);
        for (int o = 0; o < p->n_obs; ++o) slot_rec[(size_t)p->h_obs_slot[o] * 2 + p->h_obs_cam[o]] = goff[o];
        if ((rc = dev_upload(ctx, &w->d_slot_rec, slot_rec))) return rc;
    }
    if ((rc = dev_upload(ctx, &w->d_goff, goff)) || (rc = dev_upload(ctx, &w->d_slot_off, slot_off)) ||
        (rc = dev_upload(ctx, &w->d_slot_obs, slot_obs)) || (rc = dev_upload(ctx, &w->d_obs_cam, p->h_obs_cam)) ||
        (rc = dev_upload(ctx, &w->d_caminfo, caminfo)))
        return rc;
    const size_t gbytes = std::max<int64_t>(gl, 1) * sizeof(double);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->G[i], gbytes));
        HIP_TRY(ctx, hipMemsetAsync(w->G[i], 0, gbytes, ctx->stream));     // tile (1,0) of two-tile blocks is never written
        HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->cost_o[i], std::max(p->n_obs, 1) * sizeof(double)));
        HIP_TRY(ctx, hipMemsetAsync(w->cost_o[i], 0, std::max(p->n_obs, 1) * sizeof(double), ctx->stream));
    }
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->partial, (size_t)w->RB * std::max(n_pw, w->n_rows) * sizeof(double)));
    // (every clear of this function is ordered on the context's stream: it does not synchronise with the null stream, and the
    // kernels rely on what is never written staying zero - holes in the record buffers, upper-triangle rows of `partial`)
    // k_schurq writes the lower triangle and the extras only: the rows of the upper triangle stay zero.  On the context's
    // stream: it does not synchronise with the null stream, a plain hipMemset could still be running when the first
    // elimination writes the buffer
    HIP_TRY(ctx, hipMemsetAsync(w->partial, 0, (size_t)w->RB * std::max(n_pw, w->n_rows) * sizeof(double), ctx->stream));
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->red, 2 * (size_t)(w->RB + 8) * sizeof(double)));      // two buffers: the in-process transport alternates
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->pf, (size_t)std::max(p->n_slots, 1) * w->PF * sizeof(double)));
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->mc_slot, (size_t)std::max(p->n_slots, 1) * sizeof(double)));
    HIP_TRY(ctx, hipMemsetAsync(w->mc_slot, 0, (size_t)std::max(p->n_slots, 1) * sizeof(double), ctx->stream));
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->scal, 8 * sizeof(double)));
    HIP_TRY(ctx, hipMemsetAsync(w->scal, 0, 8 * sizeof(double), ctx->stream));
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->flags, 4 * sizeof(int32_t)));
    HIP_TRY(ctx, ctx_dev_alloc(ctx, (void**)&w->d_gstate, sizeof(DevState)));
    HIP_TRY(ctx, ctx_stream_get(ctx, &w->side));
    HIP_TRY(ctx, ctx_host_alloc(ctx, (void**)&w->h_gstatus, sizeof(HostStatus)));
    HIP_TRY(ctx, hipHostMalloc((void**)&w->h_gstate, sizeof(DevState), hipHostMallocDefault));
    std::memset((void*)w->h_gstatus, 0, sizeof(HostStatus));
    HIP_TRY(ctx, hipMemsetAsync(w->flags, 0, 4 * sizeof(int32_t), ctx->stream));
    HIP_TRY(ctx, hipHostMalloc((void**)&w->h_pinned, (size_t)(w->RB + 16) * sizeof(double), hipHostMallocDefault));
    w->general_ready = true;
    return normal_upload_cols(p);
}

static void build_cols(const ccal_problem* p, ColInfo* cols) {
    std::memset(cols, 0, CCAL_KMAX * sizeof(ColInfo));
    for (int c = 0; c < p->n_cams; ++c) {
        const CamLayout& cl = p->cams[c];
        for (int i = 0; i < cl.Peff; ++i) {
            ColInfo& ci = cols[cl.col_theta + i];
            // column i of the camera's block is the kernels' canonical parameter `full`; an OPENCV5 coefficient lives at
            // position 4 + ocv5_order[.] of the CALLER's vector: that is where its bounds / fixed flag were set (eff index
            // space of the caller) and where the step goes
            int full = p->one_focal ? (i == 0 ? 0 : i + 1) : i;          // eff -> full index (fy re-inserted)
            if (cl.model == kOCV5 && full >= 4) full = 4 + p->ctx->conv.ocv5_order[full - 4];
            const int q = c * CCAL_PMAX + (full - (p->one_focal && full > 0 ? 1 : 0));
            ci.lo = p->lo[q]; ci.hi = p->hi[q]; ci.has_bound = p->has_bound[q]; ci.fixed = p->fixed[q];
            ci.is_extr = 0;
            ci.dst = c * CCAL_PMAX + full;
            ci.dst2 = (p->one_focal && i == 0) ? c * CCAL_PMAX + 1 : -1;
        }
        if (c > 0) for (int i = 0; i < 6; ++i) {
            ColInfo& ci = cols[cl.col_extr + i];
            ci.is_extr = 1; ci.dst = c * 6 + i; ci.dst2 = -1;
        }
    }
}

// reduced-system column (the kernels' canonical parameter order) -> the same parameter's column in the CALLER's order
// (differs only for OPENCV5 cameras under a non-default ccal_model_conventions.ocv5_order)
static void caller_columns(const ccal_problem* p, int* ext) {
    for (int k = 0; k < p->K; ++k) ext[k] = k;
    for (int c = 0; c < p->n_cams; ++c) {
        const CamLayout& cl = p->cams[c];
        if (cl.model != kOCV5) continue;
        const int shift = p->one_focal ? 1 : 0;
        for (int d = 0; d < 5; ++d) ext[cl.col_theta + 4 - shift + d] = cl.col_theta + 4 - shift + p->ctx->conv.ocv5_order[d];
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
    NormalWs* w = p->nws;
    if (w->register_gram && !cand && gbuf == w->cur) {
        // record format (what k_schur expects): the device loop's launcher with the state set to "first evaluation of set 0"
        if (!w->gstate_is_eval) {      // as in the single-camera build: set once, valid until a solve rewrites the state
            HIP_TRY(ctx, launch_state_eval(w->d_gstate, 0.0, ctx->stream));
            w->gstate_is_eval = true;
        }
        HIP_TRY(ctx, launch_gram_dev_all(p, w->d_gstate, ctx->stream));
        return CCAL_OK;
    }
    if (w->register_gram) { ctx->err = "enqueue_gram: candidate sets go through the device loop"; return CCAL_ERR_UNSUPPORTED; }
    for (int c = 0; c < p->n_cams; ++c) HIP_TRY(ctx, launch_gram(p, c, cand, gbuf, ctx->stream));
    return CCAL_OK;
}

// The step's one collective: in-place sum of `n` doubles over the ranks, ordered on the context stream.  Native RCCL
// when a communicator is set (ccal_rccl.hip), else the callback, else nothing (single GPU).
static int allreduce(ccal_problem* p, double* buf, size_t n) {
    ccal_ctx* ctx = p->ctx;
    if (p->peer) { ctx->err = "this entry point does not run on a shard of an in-process device set (its sums are added by the solver's deciding kernel)"; return CCAL_ERR_UNSUPPORTED; }
    if (p->rccl_comm) return rccl_allreduce_sum(ctx, p->rccl_comm, buf, n, ctx->stream);
    if (p->allreduce && p->allreduce(p->allreduce_user, buf, n, (void*)ctx->stream) != 0) { ctx->err = "all-reduce callback failed"; return CCAL_ERR_HIP; }
    return CCAL_OK;
}

// schur + reduce (+ all-reduce) of G[gbuf] -> red
static int enqueue_reduce_system(ccal_problem* p, int gbuf, double lambda, double min_diag, double max_diag) {
    ccal_ctx* ctx = p->ctx;
    NormalWs* w = p->nws;
    HIP_TRY(ctx, launch_schur(p, gbuf, lambda, min_diag, max_diag, ctx->stream));
    HIP_TRY(ctx, launch_reduce(p, ctx->stream));
    return allreduce(p, w->red, (size_t)w->RB);
}

// Argument block of the single-camera kernels: set 0 = (p->d_intr, p->d_poses), set 1 = the *_c buffers.
static FusedArgs make_fused_args(const ccal_problem* p, double min_diag, double max_diag) {
    const NormalWs* w = p->nws;
    const FusedWs* f = w->fws;
    FusedArgs fa = {};
    fa.x = p->d_x; fa.y = p->d_y; fa.z = p->d_z; fa.u = p->d_u; fa.v = p->d_v;
    fa.obs_off = p->d_obs_off; fa.obs_slot = p->d_obs_slot; fa.slot_ident = p->slot_ident ? 1 : 0;
    fa.n_obs = p->n_obs; fa.K = p->K; fa.PF = w->PF; fa.PRAW = f->PRAW; fa.n_pw = f->n_pw;
    fa.fcbuf = f->fcbuf; fa.mc_f = f->mc_f; fa.cost_f = f->cost_f;
    fa.huber_delta = p->huber_delta; fa.min_diag = min_diag; fa.max_diag = max_diag; fa.rt = model_rt(p->ctx);
    fa.intr[0] = p->d_intr; fa.intr[1] = p->d_intr_c; fa.poses[0] = p->d_poses; fa.poses[1] = p->d_poses_c;
    fa.pf[0] = f->pf[0]; fa.pf[1] = f->pf[1]; fa.praw[0] = f->praw[0]; fa.praw[1] = f->praw[1];
    fa.dc = w->dc; fa.st = f->d_state; fa.partial = f->partial; fa.red = f->red;
    fa.avg_corners = (int32_t)(p->n_corners / std::max(p->n_obs, 1));
    fa.part_cap = f->n_pw;
    // ragged frames: the bins ccal_problem_create planned for the Gram launch (n_bins == 0: none)
    const GramBins& gb = p->gram_bins;
    fa.n_bins = (gb.n_bins > 0 && p->d_bin_tab) ? gb.n_bins : 0; fa.bin_tab = p->d_bin_tab;
    for (int b = 0; b < kGramMaxBins; ++b) { fa.bin_lpf[b] = gb.lpf[b]; fa.bin_first[b] = gb.first[b]; fa.bin_count[b] = gb.count[b]; fa.bin_wg0[b] = gb.wg0[b]; }
    fa.bin_wg0[kGramMaxBins] = gb.wg0[kGramMaxBins];
    return fa;
}
// Second library only: the separate elimination launch (k_schur1m, four frames per wavefront) fits the partial-sum buffer; sets fa.n_pw.
static bool fused_use_schur1m(const ccal_problem* p, FusedArgs& fa) {
    const int n_pw = (p->n_obs + 15) / 16 * 4;
    if (n_pw > p->nws->fws->n_pw) return false;
    fa.n_pw = n_pw;
    return true;
}
// Register-resident (VALU) Gram for every model: with the rotation columns in the phi basis it beats the matrix-core
// kernel also for the 120 / 136-entry triangles of OPENCV5 (10 000 frames, whole build: 70.0 vs 71.8 us two-focal, 59.9 vs
// 66.1 us one-focal; KB4 64.3 vs 74.0) - f64 MFMA shares the FP64 datapath with the VALU and a 16 x 16 tile computes
// 256 products where the triangle needs <= 136.  CCAL_GRAM=mfma selects the matrix-core kernel (k_gram1).
static bool fused_use_valu_gram(const ccal_problem* p) {
    (void)p;
#ifdef CCAL_LEGACY_KERNELS
    if (const char* g = dev_env("CCAL_GRAM")) return g[0] != 'm';
#endif
    return true;
}
// Gram + elimination of one group on the single-camera path (fa.n_part = the number of partial sums per entry)
static hipError_t enqueue_fused_gram_schur(const ccal_problem* p, FusedArgs& fa, bool schur_m, hipStream_t st) {
    fa.n_part = 0;
    if (p->n_obs <= 0) return hipSuccess;      // a rank whose shard is empty still takes part in the collective, with zeros
    // CCAL_FUSE_ELIM=0 (developer switch, read when the problem's workspace is created): always the separate elimination launch
    const bool valu = fused_use_valu_gram(p);
    fa.fuse_elim = (p->nws->fws->fuse_elim && valu) ? 1 : 0; fa.elim_fused = 0;
    hipError_t e = valu ? launch_gram1v(p->cams[0].model, p->one_focal, fa, st) : launch_gram1(p->cams[0].model, p->one_focal, fa, st);
    if (e != hipSuccess) return e;
    if (fa.elim_fused) return hipSuccess;      // the Gram kernel eliminated its frames' pose blocks itself: fa.n_part rows of partial sums
    return schur_m ? launch_schur1m(fa, st) : hipErrorNotSupported;      // (second library: matrix-core Gram / CCAL_FUSE_ELIM=0)
}
// ... + k_reduce1: the reduced sums in fws->red (the all-reduce buffer of sharded solves; ccal_build_normal_dev)
static hipError_t enqueue_fused_system(const ccal_problem* p, FusedArgs& fa, bool schur_m, hipStream_t st) {
    if (hipError_t e = enqueue_fused_gram_schur(p, fa, schur_m, st); e != hipSuccess) return e;
    return launch_reduce1(fa, st);
}

// Host side of the device-resident loops: watch the status word a decision kernel publishes to pinned memory.
// One look, no blocking: *arrived = the step `target` has been published.  `since` = when the wait for this step began;
// timeout_s: seconds without the awaited step completing before the wait gives up; <= 0 = never (sharded solves: a late
// peer is waited for - an asymmetric give-up would leave the other ranks inside a collective with no partner).
// `spins` throttles the expensive checks (clock, stream query) to one in 4 096 looks.
static int check_status(ccal_ctx* ctx, hipStream_t st, HostStatus* hst, const DevState* d_state, int target, double timeout_s,
                        std::chrono::steady_clock::time_point since, long* spins, bool* arrived) {
    *arrived = status_seq(hst->word) >= target;
    if (*arrived || (++*spins & 0xFFF) != 0) return CCAL_OK;
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - since).count();
    const hipError_t q = el > 0.002 ? hipStreamQuery(st) : hipErrorNotReady;
    if (q != hipSuccess && q != hipErrorNotReady) {      // the stream itself failed (a fault, a lost device): no step will ever publish
        ctx->err = std::string("device-resident solve: ") + hipGetErrorString(q);
        return CCAL_ERR_HIP;
    }
    if (q == hipSuccess && status_seq(hst->word) < target) {
        // stream drained but the word did not arrive: fall back to an explicit copy
        DevState ds;
        HIP_TRY(ctx, hipMemcpy(&ds, d_state, sizeof ds, hipMemcpyDeviceToHost));
        hst->iter = ds.iter; hst->cur = ds.cur; hst->lm_accepted = ds.lm_accepted;
        hst->lm_rejected = ds.lm_rejected; hst->cur_cost = ds.cur_cost; hst->initial_cost = ds.initial_cost;
        hst->spec_hits = ds.spec_hits; hst->spec_misses = ds.spec_misses;
        hst->word = status_word(target, ds.done, ds.done_seq);
        *arrived = true;
        return CCAL_OK;
    }
    if (timeout_s > 0.0 && el > timeout_s) { ctx->err = "device-resident solve timed out"; return CCAL_ERR_HIP; }
    return CCAL_OK;
}

// Seconds WITHOUT A STEP COMPLETING before the host gives up.  Single GPU: 30.  Sharded: 600 - a peer that is still
// uploading or setting up its RCCL channels is waited for (a rank that gave up after 30 s left its peers inside a
// collective with no partner), but a peer that died must not hang the survivors for ever.
static double wait_timeout(const ccal_problem* p, const ccal_solver_opts* o) {
    if (o->timeout_s > 0) return (double)o->timeout_s;
    return p->sharded() ? 600.0 : 30.0;
}
static void init_state(DevState* s, const ccal_solver_opts* o) {
    std::memset(s, 0, sizeof *s);
    const bool lm = o->method == CCAL_METHOD_LM;
    s->radius = o->lm_initial_radius; s->dec = 2.0;
    s->lambda = lm ? 1.0 / o->lm_initial_radius : 0.0;
    s->lambda_spec = lm ? 1.0 / lm_radius_cap(o->lm_initial_radius) : 0.0;
    s->min_error = o->min_error; s->min_abs = o->min_abs_error_decrease; s->min_rel = o->min_rel_error_decrease;
    s->cur = 0; s->first = 1; s->max_iter = o->max_iterations; s->method = o->method;
    s->error_metric = o->error_metric ? 1 : 0;
}
// Groups the host may enqueue at most: one per decision plus one re-elimination group per LM decision, plus the first.
static int max_groups_for(const ccal_solver_opts* o) {
    const int64_t n = (int64_t)(o->method == CCAL_METHOD_LM ? 2 : 1) * std::max(o->max_iterations, 0) + 2;
    return (int)std::min<int64_t>(n, kMaxGroups);      // sequence numbers travel in 24 bits of the status word
}
// How many groups are kept in flight ahead of the one the host waits for.  With native RCCL the collective is just
// another stream operation, so sharded solves run ahead like single-GPU ones; a callback is host code - the loop waits
// for every group before the next one (no collective is ever issued for a solve that has finished).
static int groups_in_flight(const ccal_problem* p, const char* env_name) {
    if (p->allreduce && !p->rccl_comm && !p->peer) {
        static const int hook_depth = std::max(1, dev_env_int("CCAL_FUSED_DEPTH_HOOK", 1));
        return hook_depth;
    }
    return std::max(1, dev_env_int(env_name, 2));
}

// ---------------------------------------------------------------------------------------------
// A solve as a resumable job: begin() stages the starting point and enqueues the first groups, poll() takes ONE look at
// the status word - consumes a published step, keeps `depth` groups in flight, reports when the solve has finished - and
// never blocks, end() fetches the result.  ccal_solve / ccal_solve_dev spin on poll(); ccal_solve_batch drives many jobs
// (each on its own context's stream) round-robin from one host thread, so that session-sized problems, which leave the
// GPU ~96 % idle one at a time, run side by side.
// Both loops are device-resident: accept / reject, damping, convergence tests and the camera solve run in a kernel.
// ---------------------------------------------------------------------------------------------
struct SolveJob {
    ccal_problem* p; const ccal_solver_opts* o; bool host_io;
    double* intr_io; double* poses_io; double* extr_io;
    ccal_ctx* ctx; NormalWs* w; hipStream_t st;
    std::chrono::steady_clock::time_point t0, t_wait;
    std::vector<int> pending;
    int enq = 0, max_groups = 0, depth = 1, seq = 0;
    long spins = 0;
    bool finished = false, enqueued_any = false;
    double timeout_s = 30.0;
    // single-process sharded solves: raised by the first shard that fails with something other than the solver's verdicts; its
    // peers - whose collectives will never find their partner - stop waiting at their next look instead of at the timeout
    const std::atomic<int>* cancel = nullptr;
    int share = 1;                    // ccal_solve_batch: problems solved side by side on this GPU (FusedArgs::share)
    SolveJob(ccal_problem* p_, const ccal_solver_opts* o_, bool hio, double* i, double* po, double* e)
        : p(p_), o(o_), host_io(hio), intr_io(i), poses_io(po), extr_io(e), ctx(p_->ctx), w(p_->nws), st(p_->ctx->stream) {}
    virtual ~SolveJob() {}
    virtual int begin() = 0;
    virtual int enqueue() = 0;                       // one group; returns its sequence number, < 0 on error
    virtual HostStatus* status() = 0;
    virtual const DevState* dev_state() = 0;
    virtual void mark_tail_pending() = 0;            // kernels are in flight that publish into the pinned status word
    virtual int end(ccal_report* rep) = 0;
    int fail_enqueued(int code) { mark_tail_pending(); return code; }
    // Sharded solves must issue the SAME sequence of collectives on every rank.  They do: every group carries exactly
    // one, the fill rule below is a function of the group that reported `done` only (groups enqueued = that index +
    // depth - 1, whatever the host timing), and `done` is decided from all-reduced sums, identically on every rank.
    int fill() {
        while ((int)pending.size() < depth && enq < max_groups) {
            const int sq = enqueue();
            if (sq < 0) return enqueued_any ? fail_enqueued(-sq) : -sq;
            enqueued_any = true;
            if (pending.empty()) t_wait = std::chrono::steady_clock::now();
            pending.push_back(sq); ++enq;
        }
        return CCAL_OK;
    }
    int poll(bool* fin) {
        *fin = finished;
        if (finished) return CCAL_OK;
        if (cancel && cancel->load(std::memory_order_acquire)) { ctx->err = "sharded solve: a peer shard failed"; return fail_enqueued(CCAL_ERR_HIP); }
        int rc = fill();
        if (rc != CCAL_OK) return rc;
        if (pending.empty()) { finished = true; *fin = true; return CCAL_OK; }
        const int waited = pending.front();
        bool arrived = false;
        HostStatus* hst = status();
        if ((rc = check_status(ctx, st, hst, dev_state(), waited, timeout_s, t_wait, &spins, &arrived)) != CCAL_OK) return fail_enqueued(rc);
        if (!arrived) return CCAL_OK;
        pending.erase(pending.begin());
        t_wait = std::chrono::steady_clock::now(); spins = 0;
        // act on `done` only when it was set by a step this thread has waited for: a later group may already have
        // published it, and how many groups get enqueued must not depend on that race (sharded ranks would issue
        // different numbers of collectives)
        const uint64_t wd = hst->word;
        if (status_done(wd) && status_done_seq(wd) <= waited) finished = true;
        else {
            if (o->verbose) std::printf("[ccal %s] iter %d cost %.12g\n", o->method == CCAL_METHOD_LM ? "LM" : "GN", hst->iter, hst->cur_cost);
            if ((rc = fill()) != CCAL_OK) return rc;
            if (pending.empty()) finished = true;
        }
        *fin = finished;
        return CCAL_OK;
    }
};

// Single-camera path: groups of (gram + elimination, reduce, [all-reduce], head), kernels of ccal_kernels_fused.hip /
// ccal_kernels_gram2.hip.  host_io: parameters come from / go back to the caller's host arrays (ccal_solve); otherwise they
// are and stay on the device (ccal_solve_dev).
struct FusedJob : SolveJob {
    FusedWs* f = nullptr;
    FusedArgs fa; HeadArgs ha; bool schur_m = false, sharded = false;
    // single-launch groups (k_gram1v<.., ITER>, IterArgs): launch s reads state buffer s & 1 and the rows of launch s - 1,
    // writes state buffer (s + 1) & 1 and its own rows into the other half of the partial-sum buffer
    int iter_rows = 0;
    bool batch_member = false;        // ccal_solve_batch drives this job's launches together with other problems' (run_iter_group): begin() enqueues nothing
    bool fold = false;                // the first single-launch group is also the solve's k_unpack1 (IterArgs::fold)
    UnpackArgs ua0;                   // its payload
    DevState* iter_state(int s) const { return f->d_state + 1 + (s & 1); }
    double* iter_partial(int s) const { return f->partial + (size_t)(s & 1) * iter_rows * f->RB1; }
    double* h_poses = nullptr;
    size_t np6 = 0;
    bool zero_copy = false;           // session-sized ccal_solve: poses read from / result written to pinned host memory by the kernels
    bool spread = false;              // zero_copy of a larger problem: every workgroup of the finishing single-launch group writes its slice of the poses
    void* pinned_dev = nullptr;       // large ccal_solve whose caller pinned poses_io (ccal_pin_buffer): its device-side address - no staging copy either way
    using SolveJob::SolveJob;
    HostStatus* status() override { return f->h_status; }
    const DevState* dev_state() override { return iter_rows ? iter_state(seq + 1) : f->d_state; }     // (read once the stream has drained)
    void mark_tail_pending() override { if (f) f->tail_pending = true; }
    int begin() override {
        int rc = fused_ws_ensure(p);
        if (rc != CCAL_OK) return rc;
        f = w->fws;
        const int K = p->K;
        t0 = std::chrono::steady_clock::now();
        // one pinned staging block [intr | state | cols | poses] -> ONE async copy -> k_unpack1 (both parameter sets start
        // from the same values; slots without observations never change).  ccal_solve_dev stages only the ~1 KB head.
        static_assert(sizeof(ColInfo) % 8 == 0, "ColInfo is staged as doubles");
        if (f->tail_pending) { HIP_TRY(ctx, hipStreamSynchronize(st)); f->tail_pending = false; }   // stale k_head must not publish into this solve
        np6 = (size_t)p->n_slots * 6;
        h_poses = f->h_stage;
        fa = make_fused_args(p, o->lm_min_diagonal, o->lm_max_diagonal);
        sharded = p->sharded();
        fa.share = share;
        iter_rows = 0;
        if (!sharded && f->fuse_elim && fused_use_valu_gram(p)) {
            const int rows = fused_iter_rows(p->cams[0].model, p->one_focal, p->n_obs, fa.avg_corners, K, share, batch_member);
            if (rows > 0 && 2 * rows <= f->n_pw) iter_rows = rows;
        }
        // the result straight into pinned host memory: session sizes (one workgroup writes it: k_head or the single-launch group's
        // workgroup 0); larger problems only where EVERY workgroup of the finishing launch writes a slice (single-launch groups, not
        // the members of a lockstep batch).  CCAL_RESULT_SPREAD=0 (developer switch): the DMA behind the last launch.
        static const bool spread_off = dev_env_int("CCAL_RESULT_SPREAD", 1) == 0;
        const bool large = np6 * sizeof(double) > kZeroCopyBytes;
        spread = host_io && large && f->h_result != nullptr && iter_rows > 0 && !batch_member && !spread_off;
        zero_copy = host_io && f->h_result != nullptr && (!large || spread);
        {
            // state, column table and intrinsics travel in k_unpack1's argument block; the poses are read in place from
            // pinned host memory (session sizes) or staged with one copy (large problems)
            UnpackArgs ua = {};
            init_state(&ua.st0, o);
            if (K > kFusedMaxK) { ctx->err = "single-camera loop: more than 9 camera columns"; return CCAL_ERR_INVALID_ARG; }
            ColInfo cols_all[CCAL_KMAX];
            build_cols(p, cols_all);
            for (int i = 0; i < K; ++i) ua.col0[i] = cols_all[i];
            ua.n_cols = K;
            ua.np6 = (int64_t)np6; ua.poses_on_device = host_io ? 0 : 1;
            ua.intr0 = p->d_intr; ua.intr1 = p->d_intr_c; ua.poses0 = p->d_poses; ua.poses1 = p->d_poses_c;
            ua.st = iter_rows ? iter_state(1) : f->d_state; ua.cols = w->cols;
            pinned_dev = (host_io && large && np6) ? pinned_device_ptr(poses_io, np6 * sizeof(double)) : nullptr;
            if (host_io && pinned_dev) {
                // the caller's poses are pinned: k_unpack1 reads them where they are (no copy into the staging block, no DMA)
                std::memcpy(ua.intr_h, intr_io, CCAL_PMAX * sizeof(double));
                ua.poses_src = static_cast<const double*>(pinned_dev);
            } else if (host_io) {
                std::memcpy(ua.intr_h, intr_io, CCAL_PMAX * sizeof(double));
                std::memcpy(h_poses, poses_io, np6 * sizeof(double));
                if (zero_copy || np6 == 0) ua.poses_src = h_poses;
                else {
                    HIP_TRY(ctx, hipMemcpyAsync(f->d_stage, h_poses, np6 * sizeof(double), hipMemcpyHostToDevice, st));
                    ua.poses_src = f->d_stage;
                }
            }
            f->state_is_eval = false;
            // every slot observed (single camera: one frame per slot): the first single-launch group unpacks for itself - one launch
            // and the host's gap behind it less per solve.  CCAL_ITER_FOLD=0: k_unpack1.
            static const bool fold_off = dev_env_int("CCAL_ITER_FOLD", 1) == 0;
            if (f->all_slots_observed < 0) {
                std::vector<char> seen((size_t)std::max(p->n_slots, 1), 0);
                int n_seen = 0;
                for (int sl : p->h_obs_slot) if (sl >= 0 && sl < p->n_slots && !seen[(size_t)sl]) { seen[(size_t)sl] = 1; ++n_seen; }
                f->all_slots_observed = n_seen == p->n_slots ? 1 : 0;
            }
            fold = iter_rows > 0 && !fold_off && f->all_slots_observed == 1;
            ua0 = ua;                  // (fold: the first single-launch group's payload)
            if (!fold) HIP_TRY(ctx, launch_unpack1(ua, st));
        }
        enqueued_any = true;          // from here on an error exit leaves kernels in flight: the next solve / the destructor drains
        f->h_status->word = 0;
        schur_m = fused_use_schur1m(p, fa);
        ha = HeadArgs{};
        ha.st = f->d_state; ha.hs = f->h_status; ha.red = f->red; ha.cols = w->cols;
        ha.intr[0] = p->d_intr; ha.intr[1] = p->d_intr_c; ha.dc = w->dc; ha.K = K;
        ha.min_diag = o->lm_min_diagonal; ha.max_diag = o->lm_max_diagonal;
        ha.publish_all = o->verbose ? 1 : 0;
        ha.result_host = zero_copy ? f->h_result : nullptr;
        ha.poses[0] = p->d_poses; ha.poses[1] = p->d_poses_c; ha.np6 = (int64_t)np6;
        max_groups = max_groups_for(o) + (iter_rows ? 1 : 0);          // (single-launch groups: the decision on launch s is taken in launch s + 1)
        depth = groups_in_flight(p, "CCAL_FUSED_DEPTH");
        // single-launch groups publish their status word from the FRONT of the launch (the decision), ~10 us before the launch
        // ends: the host has the next launch in the stream in time without one enqueued ahead - and a finished solve leaves
        // no launches behind that exit early (each of which still sums the rows: four / eight sessions side by side 0.244 / 0.464
        // -> 0.236 / 0.444 ms per batch, one session 0.108 ms either way).  CCAL_FUSED_DEPTH overrides.
        if (iter_rows && !dev_env("CCAL_FUSED_DEPTH")) depth = 1;
        timeout_s = wait_timeout(p, o);
        return batch_member ? CCAL_OK : fill();
    }
    // what every single-launch group of this solve finds in IterArgs (the per-launch fields are set by enqueue() / derived from the
    // launch number by k_gram1v_batch)
    void iter_args_common(IterArgs& it) const {
        it.on = 1; it.publish_all = o->verbose ? 1 : 0; it.n_part_in = iter_rows;
        it.hs = f->h_status; it.cols = w->cols; it.dc_out = w->dc;
        it.result_host = zero_copy ? f->h_result : nullptr; it.np6 = (int64_t)np6;
        it.result_poses = nullptr; it.done_cnt = nullptr;          // (a lockstep batch's members: session sizes)
        it.n_cols = ua0.n_cols; it.poses_on_device = ua0.poses_on_device; it.poses_src = ua0.poses_src; it.cols_out = ua0.cols;
        it.st0 = ua0.st0;
        for (int i = 0; i < kFusedMaxK; ++i) it.col0[i] = ua0.col0[i];
        std::memcpy(it.intr_h, ua0.intr_h, sizeof it.intr_h);
    }
    // this problem's entry of a batch's table (k_gram1v_batch): bases instead of per-launch pointers
    FusedArgs batch_entry() const {
        FusedArgs e = fa;
        iter_args_common(e.it);
        e.it.fold = fold ? 1 : 0; e.it.skip_head = 0; e.it.seq = 0;
        e.it.st_in = iter_state(0); e.it.st_out = nullptr; e.it.partial_in = nullptr;
        e.partial = f->partial; e.n_part = iter_rows; e.fuse_elim = 1; e.elim_fused = 1;
        return e;
    }
    int enqueue() override {          // one group: evaluation + elimination + ONE collective + decision/solve
        // CCAL_HEAD_REDUCE_ROWS=0 (developer switch): always the separate reduce launch
        static const int head_reduce_max_rows = std::min(dev_env_int("CCAL_HEAD_REDUCE_ROWS", kHeadReduceRows), kHeadReduceRows);
        if (iter_rows) {
            // session sizes on one GPU: the whole group is ONE launch - the Gram kernel sums the previous launch's rows, decides
            // and solves the camera system in front of its own evaluation (every workgroup the same arithmetic, workgroup 0 writes)
            const int sq = ++seq;
            IterArgs& it = fa.it;
            it.on = 1; it.skip_head = sq == 1 ? 1 : 0; it.seq = sq; it.publish_all = o->verbose ? 1 : 0;
            it.st_in = iter_state(sq); it.st_out = iter_state(sq + 1);
            it.partial_in = iter_partial(sq - 1); it.n_part_in = iter_rows; fa.partial = iter_partial(sq);
            it.hs = f->h_status; it.cols = w->cols; it.dc_out = w->dc;
            it.result_host = zero_copy ? f->h_result : nullptr; it.np6 = (int64_t)np6;
            it.result_poses = spread ? (pinned_dev ? static_cast<double*>(pinned_dev) : f->h_result + CCAL_PMAX) : nullptr;
            it.done_cnt = spread ? f->done_cnt : nullptr;
            it.fold = (fold && sq == 1) ? 1 : 0;
            if (it.fold) {
                it.n_cols = ua0.n_cols; it.poses_on_device = ua0.poses_on_device; it.poses_src = ua0.poses_src; it.cols_out = ua0.cols;
                it.st0 = ua0.st0;
                for (int i = 0; i < kFusedMaxK; ++i) it.col0[i] = ua0.col0[i];
                std::memcpy(it.intr_h, ua0.intr_h, sizeof it.intr_h);
            }
            HIP_TRYN(ctx, launch_gram_iter(p->cams[0].model, p->one_focal, fa, st));
            if (fa.n_part != iter_rows) { ctx->err = "single-launch group: row count changed"; return -CCAL_ERR_INVALID_ARG; }
            return sq;
        }
        if (sharded) {
            // Gram -> elimination -> reduce -> all-reduce of the packed sums -> head
            if (p->peer) {
                // in-process transport: the reduce leaves this rank's sums in one of two buffers, the head adds all ranks' (rank order)
                fa.red = f->red + (size_t)inproc_parity(p->peer) * f->red_stride;
                HIP_TRYN(ctx, enqueue_fused_system(p, fa, schur_m, st));
                if (inproc_post(p->peer, fa.red, (size_t)f->RB1, (void*)st, &ha.peers) != 0) { ctx->err = "in-process transport: a peer shard failed or did not arrive"; return -CCAL_ERR_HIP; }
            } else {
                HIP_TRYN(ctx, enqueue_fused_system(p, fa, schur_m, st));
                if (int e = allreduce(p, f->red, (size_t)f->RB1); e != CCAL_OK) return -e;
            }
            ha.partial = nullptr; ha.n_part = 0;
        } else {
            // single GPU: for session-sized problems (<= 40 rows of partial sums = 1 280 frames) the head adds them up itself,
            // three launches per group (600 / 1 000 frames: 23.6 / 24.4 -> 23.2 / 23.5 us per group); beyond ~3 000 frames one
            // workgroup reading all rows costs more than the reduce launch it replaces (10 000 frames: 58.6 vs 52.6 us).
            // Same sums bit for bit either way (reduce_partial_rows).
            HIP_TRYN(ctx, enqueue_fused_gram_schur(p, fa, schur_m, st));
            if (fa.n_part <= head_reduce_max_rows) { ha.partial = f->partial; ha.n_part = fa.n_part; }
            else { HIP_TRYN(ctx, launch_reduce1(fa, st)); ha.partial = nullptr; ha.n_part = 0; }
        }
        ha.seq = ++seq;
        HIP_TRYN(ctx, launch_head(ha, st));
        return seq;
    }
    int end(ccal_report* rep) override {
        HostStatus* hst = f->h_status;
        // hst->done was published by the last instruction of the deciding k_head (after a system-scope fence): everything
        // the result depends on is complete.  A download goes through a side stream so that it does not queue behind
        // the early-exit groups still in the main stream; the next solve drains those before it starts.
        hipStream_t dl = st;
        // (a finished solve whose result is already on the host - or stays on the device - does not wait for the stream either:
        // the launch that said `done` has nothing left to write; whatever runs next on this stream is ordered behind it)
        if (status_done(hst->word) && (!pending.empty() || zero_copy || !host_io)) { dl = f->side; f->tail_pending = true; }
        else HIP_TRY(ctx, hipStreamSynchronize(st));
        struct { int done, iter, cur, acc, rej, hits, misses; double cur_cost, initial_cost; } ds =
            { status_done(hst->word), hst->iter, hst->cur, hst->lm_accepted, hst->lm_rejected, hst->spec_hits, hst->spec_misses, hst->cur_cost, hst->initial_cost };
        const int status = ds.done ? ds.done - 1 : CCAL_ERR_NO_CONVERGENCE;
        if (ds.cur == 1) { std::swap(p->d_intr, p->d_intr_c); std::swap(p->d_poses, p->d_poses_c); }
        ccal_report R = {};
        R.status = status; R.iterations = ds.iter; R.lm_accepted = ds.acc; R.lm_rejected = ds.rej;
        R.lm_spec_hits = ds.hits; R.lm_spec_misses = ds.misses;
        R.initial_cost = ds.initial_cost; R.final_cost = ds.cur_cost;
        if (host_io) {
            if (zero_copy && ds.done) {
                // the k_head that set `done` wrote the result into pinned memory before it published the word this thread
                // has just read: nothing to copy, nothing to wait for
                std::memcpy(intr_io, const_cast<const double*>(f->h_result), CCAL_PMAX * sizeof(double));
                if (!(spread && pinned_dev)) std::memcpy(poses_io, const_cast<const double*>(f->h_result) + CCAL_PMAX, np6 * sizeof(double));     // (pinned: written in place)
            } else if (pinned_dev) {
                // the caller's array is pinned: ONE DMA writes the result where the caller reads it
                double* h_intr = h_poses;
                HIP_TRY(ctx, hipMemcpyAsync(poses_io, p->d_poses, np6 * sizeof(double), hipMemcpyDeviceToHost, dl));
                HIP_TRY(ctx, hipMemcpyAsync(h_intr, p->d_intr, CCAL_PMAX * sizeof(double), hipMemcpyDeviceToHost, dl));
                HIP_TRY(ctx, hipStreamSynchronize(dl));
                std::memcpy(intr_io, h_intr, CCAL_PMAX * sizeof(double));
            } else {
                double* h_intr = h_poses + np6;
                if (np6) HIP_TRY(ctx, hipMemcpyAsync(h_poses, p->d_poses, np6 * sizeof(double), hipMemcpyDeviceToHost, dl));
                HIP_TRY(ctx, hipMemcpyAsync(h_intr, p->d_intr, CCAL_PMAX * sizeof(double), hipMemcpyDeviceToHost, dl));
                HIP_TRY(ctx, hipStreamSynchronize(dl));
                std::memcpy(poses_io, h_poses, np6 * sizeof(double));
                std::memcpy(intr_io, h_intr, CCAL_PMAX * sizeof(double));
            }
            if (p->one_focal) intr_io[1] = intr_io[0];           // fy = f (src/util.rs:467-470)
        }
        R.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep) *rep = R;
        if (status == CCAL_ERR_NOT_PD) ctx->err = "normal equations are not positive definite";
        return status;
    }
};

// General loop (several cameras, or CCAL_DISABLE_FUSED): the same group shape:
//   k_gram per camera at the evaluated set -> k_schur -> k_reduce -> (ONE all-reduce) -> k_solve (decision + camera
//   solve + candidate intrinsics / extrinsics) -> k_backsub (candidate poses, model decrease per slot)
// every kernel picks its parameter / Gram set and damping from the device state, k_solve applies the shared decision
// function and publishes a status word.  Set 0 = (p->d_*, G[w->cur]), set 1 = (p->d_*_c, G[w->cur ^ 1]).
struct GeneralJob : SolveJob {
    using SolveJob::SolveJob;
    HostStatus* status() override { return w->h_gstatus; }
    const DevState* dev_state() override { return w->d_gstate; }
    void mark_tail_pending() override { w->tail_pending = true; }
    int begin() override {
        int rc;
        if ((rc = normal_ws_ensure_general(p)) != CCAL_OK) return rc;
        if (w->tail_pending) { HIP_TRY(ctx, hipStreamSynchronize(st)); w->tail_pending = false; }   // a stale k_solve must not publish into this solve
        if (host_io && (rc = ccal_upload_params(p, intr_io, poses_io, extr_io)) != CCAL_OK) return rc;
        if ((rc = normal_upload_cols(p)) != CCAL_OK) return rc;
        w->peers.n = 0; w->red_out = nullptr;
        // the candidate poses are formed in the Gram kernels' prologue (one launch less per group: k_backsub).  Slots that no
        // frame observes are then never written: both parameter sets start from the same poses
        static const bool gbs_off = dev_env_int("CCAL_GEN_BACKSUB", 1) == 0;
        // (session-sized rigs: 600 slots x 2 / 3 cameras GN 0.229 / 0.277 -> 0.221 / 0.268 ms; at 2 x 10 000 frames the prologue's
        // extra work in 4 000 wavefronts costs more than the launch it saves - 0.459 against 0.440 ms - so: up to 4 000 frames)
        w->gen_backsub = w->register_gram && !gbs_off && p->n_obs <= 4000;
        w->lm_min_diag = o->lm_min_diagonal; w->lm_max_diag = o->lm_max_diagonal;
        if (w->gen_backsub && p->n_slots && !w->all_slots_observed)
            HIP_TRY(ctx, hipMemcpyAsync(p->d_poses_c, p->d_poses, sizeof(double) * p->n_slots * 6, hipMemcpyDeviceToDevice, st));
        t0 = std::chrono::steady_clock::now();
        w->gstate_is_eval = false;
        init_state(w->h_gstate, o);
        HIP_TRY(ctx, hipMemcpyAsync(w->d_gstate, w->h_gstate, sizeof(DevState), hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemsetAsync(w->flags, 0, 4 * sizeof(int32_t), st));
        if (p->n_slots) HIP_TRY(ctx, hipMemsetAsync(w->mc_slot, 0, (size_t)p->n_slots * sizeof(double), st));
        w->h_gstatus->word = 0;
        enqueued_any = true;
        max_groups = max_groups_for(o);
        // sharded solves: every rank issues the same sequence of collectives - one per group, the decisions come from
        // all-reduced sums and the number of groups enqueued depends only on the group that reported `done`
        depth = groups_in_flight(p, "CCAL_GENERAL_DEPTH");
        timeout_s = wait_timeout(p, o);
        return fill();
    }
    int enqueue() override {          // returns the sequence number that marks the group's end, < 0 on error
        const double min_d = o->lm_min_diagonal, max_d = o->lm_max_diagonal;
        DevState* ds = w->d_gstate;
        HIP_TRYN(ctx, launch_gram_dev_all(p, ds, st));
        HIP_TRYN(ctx, launch_schur(p, w->cur, 0.0, min_d, max_d, st, ds));
        if (p->peer) {
            w->red_out = w->red + (size_t)inproc_parity(p->peer) * (size_t)(w->RB + 8);
            HIP_TRYN(ctx, launch_reduce(p, st, ds));
            const int bad = inproc_post(p->peer, w->red_out, (size_t)w->RB, (void*)st, &w->peers);
            w->red_out = nullptr;
            if (bad) { w->peers.n = 0; ctx->err = "in-process transport: a peer shard failed or did not arrive"; return -CCAL_ERR_HIP; }
        } else {
            HIP_TRYN(ctx, launch_reduce(p, st, ds));
            if (int e = allreduce(p, w->red, (size_t)w->RB); e != CCAL_OK) return -e;
        }
        HIP_TRYN(ctx, launch_solve(p, 0.0, min_d, max_d, st, ds, w->h_gstatus, ++seq, o->verbose != 0));
        w->peers.n = 0;
        if (!w->gen_backsub) HIP_TRYN(ctx, launch_backsub(p, 0.0, min_d, max_d, st, ds));
        return seq;
    }
    int end(ccal_report* rep) override {
        HostStatus* hst = w->h_gstatus;
        // the deciding k_solve published after a system-scope fence: the result is complete; it is downloaded through a
        // side stream so that it does not queue behind the early-exit group enqueued ahead (drained before the next solve)
        hipStream_t dl = st;
        if (status_done(hst->word) && !pending.empty()) { dl = w->side; w->tail_pending = true; }
        else HIP_TRY(ctx, hipStreamSynchronize(st));
        const int status = status_done(hst->word) ? status_done(hst->word) - 1 : CCAL_ERR_NO_CONVERGENCE;
        if (hst->cur == 1) {             // the accepted point lives in set 1: make it set 0 for whoever comes next
            std::swap(p->d_intr, p->d_intr_c); std::swap(p->d_poses, p->d_poses_c); std::swap(p->d_extr, p->d_extr_c);
            w->cur ^= 1;
        }
        ccal_report R = {};
        R.status = status; R.iterations = hst->iter; R.lm_accepted = hst->lm_accepted; R.lm_rejected = hst->lm_rejected;
        R.lm_spec_hits = hst->spec_hits; R.lm_spec_misses = hst->spec_misses;
        R.initial_cost = hst->initial_cost; R.final_cost = hst->cur_cost;
        if (host_io) {
            HIP_TRY(ctx, hipMemcpyAsync(intr_io, p->d_intr, sizeof(double) * p->n_cams * CCAL_PMAX, hipMemcpyDeviceToHost, dl));
            if (poses_io && p->n_slots) HIP_TRY(ctx, hipMemcpyAsync(poses_io, p->d_poses, sizeof(double) * p->n_slots * 6, hipMemcpyDeviceToHost, dl));
            if (extr_io) HIP_TRY(ctx, hipMemcpyAsync(extr_io, p->d_extr, sizeof(double) * p->n_cams * 6, hipMemcpyDeviceToHost, dl));
            HIP_TRY(ctx, hipStreamSynchronize(dl));
            if (p->one_focal) for (int c = 0; c < p->n_cams; ++c) intr_io[c * CCAL_PMAX + 1] = intr_io[c * CCAL_PMAX];   // fy = f (src/util.rs:467-470)
        }
        R.solve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep) *rep = R;
        if (status == CCAL_ERR_NOT_PD) ctx->err = "normal equations are not positive definite";
        return status;
    }
};

// single camera: its own device-resident loop (GN and LM, sharded or not, empty shards included); the choice must not
// depend on anything rank-local, or sharded ranks would issue different collectives
static bool use_fused_path(const ccal_problem* p) { return p->n_cams == 1 && !dev_env("CCAL_DISABLE_FUSED"); }

extern "C" {

// developer hook of -DCCAL_STAMPS builds only (not part of include/ccal.h, not exported by the product build): copy the
// per-frame scratch of the fast path to the host; the diagnostic builds (tools/stamps_*.py) park in-kernel timestamps there
#ifdef CCAL_STAMPS
int ccal_debug_fcbuf(ccal_problem* p, double* out, int64_t n) {
    if (!p || !p->nws || !out) return CCAL_ERR_INVALID_ARG;
    if (!p->nws->fws) {          // general loop: what a -DCCAL_STAMPS build of k_schur4 leaves behind the partial sums
        const NormalWs* w = p->nws;
        if (!w->general_ready) return CCAL_OK;
        const int64_t room = (int64_t)w->RB * (std::max(w->n_pw, w->n_rows) - w->n_rows), m = std::min(n, room);
        if (m > 0 && hipMemcpy(out, w->partial + (int64_t)w->RB * w->n_rows, m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return CCAL_ERR_HIP;
        return CCAL_OK;
    }
    const int64_t m = std::min<int64_t>(n, (int64_t)std::max(p->n_obs, 1) * 40);
    if (hipMemcpy(out, p->nws->fws->fcbuf, m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return CCAL_ERR_HIP;
    return CCAL_OK;
}
#endif

int ccal_build_normal_dev(ccal_problem* p, double lambda) {
    if (!p) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    HIP_TRY(p->ctx, hipSetDevice(p->ctx->device));
    int rc = normal_ws_ensure(p);
    if (rc != CCAL_OK) return rc;
    NormalWs* w = p->nws;
    w->red_fused = false;
    if (p->n_cams == 1 && p->n_obs > 0 && !p->sharded() && !dev_env("CCAL_DISABLE_FUSED")) {
        // single camera: the device loop's own kernels (register / LDS Gram + per-frame elimination), evaluated at the
        // current parameters with the state set to "first evaluation"; fws->red = [A_dir | Y^T Y | . | failed blocks]
        ccal_ctx* ctx = p->ctx;
        if ((rc = fused_ws_ensure(p)) != CCAL_OK) return rc;
        FusedWs* f = w->fws;
        hipStream_t st = ctx->stream;
        if (f->tail_pending) { HIP_TRY(ctx, hipStreamSynchronize(st)); f->tail_pending = false; }
        FusedArgs fa = make_fused_args(p, 1e-6, 1e32);
        const bool schur_m = fused_use_schur1m(p, fa);
        // the device state only has to say "first evaluation of set 0 with this damping" - it still does after a build
        // (nothing decides), so back-to-back builds set it once; a solve invalidates the note
        if (!(f->state_is_eval && f->state_eval_lambda == lambda)) {
            HIP_TRY(ctx, launch_state_eval(f->d_state, lambda, st));
            f->state_is_eval = true; f->state_eval_lambda = lambda;
        }
        HIP_TRY(ctx, enqueue_fused_system(p, fa, schur_m, st));
        w->red_fused = true;
        return CCAL_OK;
    }
    if ((rc = normal_ws_ensure_general(p)) != CCAL_OK) return rc;
    if (w->tail_pending) { HIP_TRY(p->ctx, hipStreamSynchronize(p->ctx->stream)); w->tail_pending = false; }
    if ((rc = enqueue_gram(p, false, w->cur)) != CCAL_OK) return rc;
    return enqueue_reduce_system(p, w->cur, lambda, 1e-6, 1e32);
    CCAL_API_CATCH(p->ctx)
}

int ccal_build_normal(ccal_problem* p, const double* intr, const double* poses, const double* extr,
                      double lambda, double* S, double* b, double* cost) {
    if (!p || !intr || (!poses && p->n_slots) || (!extr && p->n_cams > 1)) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    ccal_ctx* ctx = p->ctx;
    int rc = ccal_upload_params(p, intr, poses, extr);
    if (rc != CCAL_OK) return rc;
    if ((rc = normal_ws_ensure(p)) != CCAL_OK) return rc;
    NormalWs* w = p->nws;
    if ((rc = ccal_build_normal_dev(p, lambda)) != CCAL_OK) return rc;
    const int K = w->K, K1 = K + 1;
    double failed = 0.0;
    int ec[CCAL_KMAX];
    caller_columns(p, ec);                   // S and b leave in the caller's parameter order
    if (w->red_fused) {
        // [A_dir | Y^T Y | . | failed]: S = A_dir - Y^T Y on the camera block, b its last column, cost = the r x r corner of A_dir
        double* h = w->fws->h_stage;
        HIP_TRY(ctx, hipMemcpyAsync(h, w->fws->red, (size_t)w->fws->RB1 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double* A = h; const double* Y = h + K1 * K1;
        if (S) for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
            double v = A[i * K1 + j] - Y[i * K1 + j];
            if (i == j && lambda > 0.0) v += lambda * std::min(std::max(A[i * K1 + i], 1e-6), 1e32);
            S[ec[i] * K + ec[j]] = v;
        }
        if (b) for (int i = 0; i < K; ++i) b[ec[i]] = A[i * K1 + K] - Y[i * K1 + K];
        if (cost) *cost = A[K * K1 + K];
        failed = h[2 * K1 * K1 + 1];
    } else {
        double* h = w->h_pinned;
        HIP_TRY(ctx, hipMemcpyAsync(h, w->red, w->RB * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double* hd = h + K1 * K1;
        // the general path keeps the lower triangle of the (K + 1) x (K + 1) system (row K = b^T)
        if (S) for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) {
            double v = h[i >= j ? i * K1 + j : j * K1 + i];
            if (i == j && lambda > 0.0) v += lambda * std::min(std::max(hd[i], 1e-6), 1e32);
            S[ec[i] * K + ec[j]] = v;
        }
        if (b) for (int i = 0; i < K; ++i) b[ec[i]] = h[K * K1 + i];
        if (cost) *cost = h[w->RB - 3];
        failed = h[w->RB - 1];
    }
    if (failed > 0.0) { ctx->err = "a frame's pose block is not positive definite"; return CCAL_ERR_NOT_PD; }
    return CCAL_OK;
    CCAL_API_CATCH(p->ctx)
}

static std::unique_ptr<SolveJob> make_job(ccal_problem* p, const ccal_solver_opts* o, bool host_io, double* intr_io, double* poses_io, double* extr_io) {
    if (use_fused_path(p)) return std::unique_ptr<SolveJob>(new FusedJob(p, o, host_io, intr_io, poses_io, extr_io));
    return std::unique_ptr<SolveJob>(new GeneralJob(p, o, host_io, intr_io, poses_io, extr_io));
}
static int solve_entry(ccal_problem* p, const ccal_solver_opts* o, bool host_io, double* intr_io, double* poses_io, double* extr_io,
                       ccal_report* rep, const std::atomic<int>* cancel = nullptr, int share = 1) {
    ccal_ctx* ctx = p->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = normal_ws_ensure(p);
    if (rc != CCAL_OK) return rc;
    auto job = make_job(p, o, host_io, intr_io, poses_io, extr_io);
    job->cancel = cancel;
    job->share = share;
    if ((rc = job->begin()) != CCAL_OK) return rc;
    bool fin = false;
    while (!fin) if ((rc = job->poll(&fin)) != CCAL_OK) return rc;
    return job->end(rep);
}

}  // extern "C"

// A context's helper thread for ccal_solve_batch.  Persistent: a session-sized solve is ~0.15 ms, creating a thread per call
// costs a third of that (measured: four problems per call 1.35x over one at a time with threads made per call).  It spins for
// a short while after a task (the next batch usually follows at once), then sleeps on a condition variable.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
struct ccal_ctx_worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv, cv_done;
    std::function<void()> task;
    std::atomic<int> has_task{0}, done{0}, stop{0};
    // how long a helper spins for its next task / the caller for a helper's result before either sleeps on its condition variable:
    // the next batch of session-sized solves usually follows within this window, a 50 000-frame upload does not - and must not
    // burn a core while it lasts
    static constexpr double kSpinSeconds = 100e-6;
    static bool spun_out(std::chrono::steady_clock::time_point t0, int spins) {
        return (spins & 0x3F) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kSpinSeconds;
    }
    void loop() {
        for (;;) {
            int spins = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (!has_task.load(std::memory_order_acquire) && !stop.load(std::memory_order_acquire)) {
                if (!spun_out(t0, ++spins)) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_task.load() || stop.load(); });
            }
            if (stop.load()) return;
            task();
            has_task.store(0, std::memory_order_release);
            { std::lock_guard<std::mutex> lk(m); done.store(1, std::memory_order_release); }
            cv_done.notify_all();
        }
    }
    void submit(std::function<void()> fn) {
        { std::lock_guard<std::mutex> lk(m); task = std::move(fn); done.store(0); has_task.store(1, std::memory_order_release); }
        cv.notify_one();
    }
    // spin briefly (the helpers of a batch finish within microseconds of the caller's own solve), then sleep on the condition variable
    void wait() {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 1;; ++spins) { if (done.load(std::memory_order_acquire)) return; if (spun_out(t0, spins)) break; cpu_relax(); }
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return done.load(std::memory_order_acquire) != 0; });
    }
};
namespace ccal {
void ctx_worker_destroy(ccal_ctx* ctx) {
    ccal_ctx_worker* w = ctx->worker;
    if (!w) return;
    { std::lock_guard<std::mutex> lk(w->m); w->stop.store(1); }
    w->cv.notify_one();
    if (w->th.joinable()) w->th.join();
    delete w;
    ctx->worker = nullptr;
}
}  // namespace ccal
// The helper is published to the context only once its thread runs: if std::thread throws (EAGAIN) the context keeps no
// worker and the exception becomes the call's CCAL_ERR_HIP - a later call tries again instead of waiting on a thread
// that never existed.
static ccal_ctx_worker* ctx_worker(ccal_ctx* ctx) {
    if (!ctx->worker) {
        std::unique_ptr<ccal_ctx_worker> w(new ccal_ctx_worker());
        ccal_ctx_worker* raw = w.get();
        w->th = std::thread([raw] { raw->loop(); });
        ctx->worker = w.release();
    }
    return ctx->worker;
}

// ---- ccal_solve_batch at session size: ONE launch per optimizer step for a whole group of problems ---------------------------
// A session-sized single-camera problem solved through single-launch groups costs the host four launches and their polls; n of
// them driven by n host threads were serialised by the runtime's launch path - eight 625-frame sessions took 0.48 ms per batch for
// ~0.1 ms of device work, whatever the kernels did.  Problems of one device, model, focal mode and lane mapping therefore advance in
// LOCKSTEP through k_gram1v_batch: launch s serves all of them (blockIdx.y = the problem; its argument block is an entry of a table
// written once per batch), the host waits for every member's word of launch s and enqueues launch s + 1 while any member is
// still running; a member that has finished leaves its later workgroups at their first look at the state.  Same arithmetic per
// problem as its own single-launch groups with this lane mapping: same verdicts, iteration counts, bits.
struct IterGroupKey { int device, model, one_focal, lpf; bool operator==(const IterGroupKey& o) const { return device == o.device && model == o.model && one_focal == o.one_focal && lpf == o.lpf; } };
// the lane mapping of the problem's single-launch groups when it is one of `share` problems side by side (0: that form does not apply)
static int batch_iter_lpf(const ccal_problem* p, int share) {
    if (p->n_cams != 1 || p->sharded() || p->n_obs <= 0 || p->K > kFusedMaxK || dev_env("CCAL_DISABLE_FUSED") || dev_env("CCAL_BATCH_LOCKSTEP_OFF")) return 0;
    const int avg = (int)(p->n_corners / std::max(p->n_obs, 1));
    const int rows = fused_iter_rows(p->cams[0].model, p->one_focal, p->n_obs, avg, p->K, share, true);
    if (rows <= 0 || 2 * rows > fused_partial_rows(p->n_obs)) return 0;
    return fused_iter_lpf(p->n_obs, avg, share);
}
// solve the members of one group; rc / reps per member.  Runs on the caller's thread.
// A member whose own preparation fails (workspace allocation, staging) takes its own code and error text and leaves the group; its
// neighbours go on (include/ccal.h: every problem's own verdict in reports[i]).  `retry`: members that are fine but do not fit this
// group after all - the caller solves them on their own.
static void run_iter_group(const IterGroupKey& key, const std::vector<int>& members_in, ccal_problem** ps, const ccal_solver_opts* o, bool host_io,
                           double** intr_io, double** poses_io, int share, std::vector<int>& rc, ccal_report* reps, std::vector<int>& retry) {
    std::vector<int> members;                                   // the members that made it into the launches
    ccal_ctx* c0 = ps[members_in[0]]->ctx;
    hipStream_t st = c0->stream;
    // failures of the GROUP (its device, its table): every member that is in the launches shares them
    auto fail_all = [&](const std::vector<int>& who, int code, const char* why) { for (int i : who) { rc[i] = code; note_error(ps[i]->ctx, why); } };
    if (hipSetDevice(c0->device) != hipSuccess) { fail_all(members_in, CCAL_ERR_HIP, "hipSetDevice failed"); return; }
    std::vector<std::unique_ptr<FusedJob>> jobs;
    jobs.reserve(members_in.size());
    int max_rows = 0, max_groups = 0;
    for (int i : members_in) {
        ccal_problem* p = ps[i];
        int r = normal_ws_ensure(p);
#ifdef CCAL_TEST_HOOKS      // tests/test_gpu_iter.py (second library only): problem number CCAL_TEST_FAIL_BATCH_MEMBER of the batch fails its preparation
        if (const char* e = std::getenv("CCAL_TEST_FAIL_BATCH_MEMBER"); e && std::atoi(e) == i) { r = CCAL_ERR_NO_MEMORY; note_error(p->ctx, "injected failure of a batch member (test hook)"); }
#endif
        std::unique_ptr<FusedJob> j;
        if (r == CCAL_OK) {
            // a stale launch of this problem's last solve may still sit in ITS OWN stream: the group's launches run elsewhere
            if (p->nws->fws && p->nws->fws->tail_pending && p->ctx->stream != st) { (void)hipStreamSynchronize(p->ctx->stream); p->nws->fws->tail_pending = false; }
            const bool fresh = !p->nws->fws;                     // begin() makes the workspace: its clears are ordered on the problem's own stream
            j.reset(new FusedJob(p, o, host_io, host_io ? intr_io[i] : nullptr, host_io && poses_io ? poses_io[i] : nullptr, nullptr));
            j->w = p->nws; j->st = st; j->share = share; j->batch_member = true;
            r = j->begin();
            if (r == CCAL_OK && fresh && p->ctx->stream != st && hipStreamSynchronize(p->ctx->stream) != hipSuccess) { r = CCAL_ERR_HIP; note_error(p->ctx, "hipStreamSynchronize failed"); }
            if (r == CCAL_OK && (j->iter_rows <= 0 || fused_iter_lpf(p->n_obs, j->fa.avg_corners, share) != key.lpf)) {
                // prepared, but not in this group's shape after all: solved on its own behind the group (nothing of it was enqueued
                // on the group's stream but begin()'s staging, which its own solve repeats)
                (void)hipStreamSynchronize(st);
                j.reset();
                retry.push_back(i);
                continue;
            }
        }
        if (r != CCAL_OK) { rc[i] = r; (void)hipStreamSynchronize(st); continue; }      // this member's own code; its context holds the reason
        max_rows = std::max(max_rows, j->iter_rows); max_groups = std::max(max_groups, j->max_groups);
        jobs.push_back(std::move(j));
        members.push_back(i);
    }
    const int n = (int)members.size();
    if (n == 0) return;
    // the table: grown on demand, kept by the group's first context
    const size_t bytes = (size_t)n * sizeof(FusedArgs);
    if (c0->batch_tab_bytes < bytes) {
        if (c0->d_batch_tab) (void)hipFree(c0->d_batch_tab);
        if (c0->h_batch_tab) (void)hipHostFree(c0->h_batch_tab);
        c0->d_batch_tab = nullptr; c0->h_batch_tab = nullptr; c0->batch_tab_bytes = 0;
        const size_t want = std::max(bytes, (size_t)16 * sizeof(FusedArgs));
        if (hipMalloc((void**)&c0->d_batch_tab, want) != hipSuccess || hipHostMalloc((void**)&c0->h_batch_tab, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); fail_all(members, CCAL_ERR_NO_MEMORY, "ccal_solve_batch: out of memory for the batch table"); (void)hipStreamSynchronize(st); return;
        }
        c0->batch_tab_bytes = want;
    }
    FusedArgs* h_tab = reinterpret_cast<FusedArgs*>(c0->h_batch_tab);
    for (int k = 0; k < n; ++k) h_tab[k] = jobs[(size_t)k]->batch_entry();
    bool ok = hipMemcpyAsync(c0->d_batch_tab, h_tab, bytes, hipMemcpyHostToDevice, st) == hipSuccess;
    std::vector<char> fin((size_t)n, 0);
    int n_fin = 0, err = CCAL_OK;
    for (int s = 1; ok && n_fin < n && s <= max_groups; ++s) {
        if (launch_gram_iter_batch(key.model, key.one_focal != 0, key.lpf, reinterpret_cast<const FusedArgs*>(c0->d_batch_tab), n, max_rows, s, st) != hipSuccess) { ok = false; break; }
        const auto t_wait = std::chrono::steady_clock::now();
        std::vector<long> spins((size_t)n, 0);
        int arrived_n = n_fin;
        std::vector<char> got(fin);
        while (arrived_n < n) {
            for (int k = 0; k < n; ++k) {
                if (got[(size_t)k]) continue;
                FusedJob* j = jobs[(size_t)k].get();
                j->seq = s;
                bool arrived = false;
                const int r = check_status(j->ctx, st, j->f->h_status, j->dev_state(), s, j->timeout_s, t_wait, &spins[(size_t)k], &arrived);
                if (r != CCAL_OK) { err = r; ok = false; break; }
                if (!arrived) continue;
                got[(size_t)k] = 1; ++arrived_n;
                const uint64_t wd = j->f->h_status->word;
                if (status_done(wd) && status_done_seq(wd) <= s) { fin[(size_t)k] = 1; ++n_fin; }
                else if (o->verbose) std::printf("[ccal %s] problem %d iter %d cost %.12g\n", o->method == CCAL_METHOD_LM ? "LM" : "GN", members[(size_t)k], j->f->h_status->iter, j->f->h_status->cur_cost);
            }
            if (!ok) break;
        }
    }
    // nothing of the group may still run when the members' buffers change hands (end() swaps parameter sets; the next solve of a
    // member may use its own stream)
    if (hipStreamSynchronize(st) != hipSuccess) ok = false;
    for (int k = 0; k < n; ++k) {
        FusedJob* j = jobs[(size_t)k].get();
        const int i = members[(size_t)k];
        if (!ok) { rc[i] = err != CCAL_OK ? err : CCAL_ERR_HIP; if (err == CCAL_OK) note_error(j->ctx, "ccal_solve_batch: a launch of the lockstep group failed"); j->f->tail_pending = false; continue; }
        j->finished = true;
        rc[i] = j->end(reps ? &reps[i] : nullptr);
        j->f->tail_pending = false;
    }
}

namespace ccal {
void ctx_worker_submit(ccal_ctx* ctx, std::function<void()> fn) { ctx_worker(ctx)->submit(std::move(fn)); }
void ctx_worker_wait(ccal_ctx* ctx) { if (ctx->worker) ctx->worker->wait(); }
}  // namespace ccal

extern "C" {

// Several independent problems at once.  The problems of one context share its stream: they are solved one after the other
// by one host thread; every further context is driven by its own (persistent) helper thread, the first one by the caller's,
// so the contexts' streams are fed - a group is three to seven launches, ~10 us of host time, against ~35 us on the device
// for a session-sized problem: ONE thread feeding four streams is host-bound (measured: 1.3x over one problem at a time) -
// and their kernels overlap on the GPU.  Every problem's own verdict goes to its report; the return value is CCAL_OK unless a
// call failed for a reason other than the solver's verdicts (then the first such code, in problem order).
int ccal_solve_batch(ccal_problem** ps, int n, const ccal_solver_opts* o, double** intr_io, double** poses_io, double** extr_io,
                     ccal_report* reps) {
    if (!ps || n < 0 || !o) return CCAL_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) {
        if (!ps[i]) return CCAL_ERR_INVALID_ARG;
        for (int k = 0; k < i; ++k) if (ps[k] == ps[i]) return CCAL_ERR_INVALID_ARG;
        if (intr_io && (!intr_io[i] || (!(poses_io && poses_io[i]) && ps[i]->n_slots) || (!(extr_io && extr_io[i]) && ps[i]->n_cams > 1))) return CCAL_ERR_INVALID_ARG;
        if (ps[i]->sharded()) { ps[i]->ctx->err = "ccal_solve_batch: sharded problems are solved one at a time (their collectives order the ranks)"; return CCAL_ERR_UNSUPPORTED; }
    }
    ccal_ctx* c0 = n ? ps[0]->ctx : nullptr;
    CCAL_API_TRY
    const bool host_io = intr_io != nullptr;
    std::vector<int> rc(n, CCAL_ERR_HIP);
    std::vector<ccal_ctx*> ctxs;                                  // distinct contexts, in order of first appearance
    for (int i = 0; i < n; ++i) if (std::find(ctxs.begin(), ctxs.end(), ps[i]->ctx) == ctxs.end()) ctxs.push_back(ps[i]->ctx);
    // how many of the batch's problems sit on each GPU: a problem's launches are sized for its share of that chip (FusedArgs::share)
    auto side_by_side = [&](const ccal_ctx* c) { int k = 0; for (int i = 0; i < n; ++i) k += ps[i]->ctx->device == c->device ? 1 : 0; return k; };
    // session-sized single-camera problems of one device, model, focal mode and lane mapping advance in lockstep, one launch per
    // step for the whole group (run_iter_group); everything else is driven per context as before
    std::vector<IterGroupKey> gkeys;
    std::vector<std::vector<int>> gmembers;
    std::vector<char> grouped((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        const int share = side_by_side(ps[i]->ctx);
        const int lpf = share >= 2 ? batch_iter_lpf(ps[i], share) : 0;
        if (!lpf) continue;
        const IterGroupKey key = { ps[i]->ctx->device, ps[i]->cams[0].model, ps[i]->one_focal ? 1 : 0, lpf };
        size_t g = 0;
        while (g < gkeys.size() && !(gkeys[g] == key)) ++g;
        if (g == gkeys.size()) { gkeys.push_back(key); gmembers.emplace_back(); }
        gmembers[g].push_back(i);
    }
    for (size_t g = 0; g < gkeys.size(); ++g) if (gmembers[g].size() >= 2) for (int i : gmembers[g]) grouped[(size_t)i] = 1;
    ctxs.clear();                                                  // contexts that still have problems of their own to drive
    std::vector<ccal_ctx*> late;                                   // ... those among them that also hold members of a lockstep group
    auto holds_grouped = [&](const ccal_ctx* c) { for (int i = 0; i < n; ++i) if (grouped[(size_t)i] && ps[i]->ctx == c) return true; return false; };
    for (int i = 0; i < n; ++i) {
        if (grouped[(size_t)i]) continue;
        std::vector<ccal_ctx*>& dst = holds_grouped(ps[i]->ctx) ? late : ctxs;
        if (std::find(dst.begin(), dst.end(), ps[i]->ctx) == dst.end()) dst.push_back(ps[i]->ctx);
    }
    auto run_ctx = [&](ccal_ctx* c) noexcept {
        const int n_side = side_by_side(c);
        for (int i = 0; i < n; ++i) {
            if (ps[i]->ctx != c || grouped[(size_t)i]) continue;
            try {
                rc[i] = solve_entry(ps[i], o, host_io, host_io ? intr_io[i] : nullptr, host_io && poses_io ? poses_io[i] : nullptr,
                                    host_io && extr_io ? extr_io[i] : nullptr, reps ? &reps[i] : nullptr, nullptr, n_side);
            } catch (const std::bad_alloc&) { rc[i] = CCAL_ERR_NO_MEMORY; note_error(c, "out of host memory");
            } catch (...) { rc[i] = CCAL_ERR_HIP; note_error(c, "C++ exception in ccal_solve_batch"); }
            if (reps && rc[i] != CCAL_OK && rc[i] != reps[i].status) { reps[i] = ccal_report{}; reps[i].status = rc[i]; }
        }
    };
    bool any_group = false;
    for (size_t g = 0; g < gkeys.size(); ++g) any_group = any_group || gmembers[g].size() >= 2;
    std::vector<ccal_ctx_worker*> busy;
    busy.reserve(ctxs.size());
    try {
        // (the caller's thread drives the lockstep groups when there are any, else the first context)
        for (size_t k = any_group ? 0 : 1; k < ctxs.size(); ++k) {
            ccal_ctx_worker* wk = ctx_worker(ctxs[k]);
            ccal_ctx* c = ctxs[k];
            wk->submit([&run_ctx, c] { run_ctx(c); });
            busy.push_back(wk);
        }
    } catch (...) {                                               // a helper could not be made: what was handed out finishes first
        for (ccal_ctx_worker* wk : busy) wk->wait();
        throw;
    }
    if (any_group) {
        for (size_t g = 0; g < gkeys.size(); ++g) {
            if (gmembers[g].size() < 2) continue;
            std::vector<int> retry;
            try {
                run_iter_group(gkeys[g], gmembers[g], ps, o, host_io, intr_io, poses_io, side_by_side(ps[gmembers[g][0]]->ctx), rc, reps, retry);
            } catch (const std::bad_alloc&) { for (int i : gmembers[g]) rc[i] = CCAL_ERR_NO_MEMORY;
            } catch (...) { for (int i : gmembers[g]) rc[i] = CCAL_ERR_HIP; }
            for (int i : retry) { grouped[(size_t)i] = 0; if (std::find(late.begin(), late.end(), ps[i]->ctx) == late.end()) late.push_back(ps[i]->ctx); }
            for (int i : gmembers[g]) if (grouped[(size_t)i] && reps && rc[i] != CCAL_OK && rc[i] != reps[i].status) { reps[i] = ccal_report{}; reps[i].status = rc[i]; }
        }
        // contexts that hold members of a group AND problems of their own (or a member sent back): driven here, behind the groups -
        // never by a helper thread while the group's thread may write the same context's error text
        for (ccal_ctx* c : late) run_ctx(c);
    } else if (!ctxs.empty()) run_ctx(ctxs[0]);
    for (ccal_ctx_worker* wk : busy) wk->wait();
    for (int i = 0; i < n; ++i)
        if (rc[i] == CCAL_ERR_HIP || rc[i] == CCAL_ERR_INVALID_ARG || rc[i] == CCAL_ERR_NO_MEMORY || rc[i] == CCAL_ERR_UNSUPPORTED) return rc[i];
    return CCAL_OK;
    CCAL_API_CATCH(c0)
}

// ---- ONE problem whose frame slots are sharded over n contexts of THIS process (include/ccal.h) --------------------
// Every shard is driven by a host thread of its own - the caller's thread takes shard 0, the contexts' persistent helper
// threads the others - exactly like the ranks of a multi-process run: each thread runs the ordinary device-resident loop
// on its shard and issues the step's collective on its context's stream (native RCCL: one communicator per device made
// by ncclCommInitAll; or a stream-ordered callback transport).  The decisions are functions of all-reduced sums only, so
// every shard ends with the same status, iteration count and - bit for bit - the same camera block.
static bool verdict(int rc) { return rc == CCAL_OK || rc == CCAL_ERR_NONFINITE || rc == CCAL_ERR_NOT_PD || rc == CCAL_ERR_NO_CONVERGENCE; }
}  // extern "C"
namespace ccal {
int solve_sharded_run(ccal_problem** ps, int n, const ccal_solver_opts* o, double* intr_io, double* const* poses_io, double* extr_io,
                      ccal_report* rep, InprocComm* inproc) {
    const int n_cams = ps[0]->n_cams;
    const size_t ni = (size_t)n_cams * CCAL_PMAX, ne = (size_t)n_cams * 6;
    std::vector<double> intr((size_t)n * ni), extr((size_t)n * ne, 0.0);
    for (int i = 0; i < n; ++i) {
        std::memcpy(&intr[i * ni], intr_io, ni * sizeof(double));
        if (extr_io) std::memcpy(&extr[i * ne], extr_io, ne * sizeof(double));
    }
    std::vector<int> rc((size_t)n, CCAL_ERR_HIP);
    std::vector<ccal_report> reps((size_t)n);
    // fault injection for tests/test_gpu_multi.py - in builds with -DCCAL_TEST_HOOKS only (libccal_hip_legacy.so, which the product
    // never loads): CCAL_TEST_FAIL_SHARD=r makes shard r fail before it enqueues anything - its peers are then inside the step's
    // collective with no partner, which is what the transport's abort / timeout path is for
#ifdef CCAL_TEST_HOOKS
    const int fail_shard = [] { const char* e = std::getenv("CCAL_TEST_FAIL_SHARD"); return e ? std::atoi(e) : -1; }();
#else
    constexpr int fail_shard = -1;
#endif
    std::atomic<int> cancel{0}, first_failed{-1};        // first_failed: the shard that raised `cancel` (its peers fail BECAUSE of it)
    auto run = [&](int i) noexcept {
        try {
            if (i == fail_shard) { rc[i] = CCAL_ERR_HIP; note_error(ps[i]->ctx, "injected failure (test hook)"); }
            else rc[i] = solve_entry(ps[i], o, true, &intr[i * ni], poses_io ? poses_io[i] : nullptr, &extr[i * ne], &reps[i], &cancel);
        } catch (const std::bad_alloc&) { rc[i] = CCAL_ERR_NO_MEMORY; note_error(ps[i]->ctx, "out of host memory");
        } catch (...) { rc[i] = CCAL_ERR_HIP; note_error(ps[i]->ctx, "C++ exception in ccal_solve_sharded"); }
        if (!verdict(rc[i])) {
            // this shard has left the shared sequence of collectives: its peers must not wait for it - neither in the in-process
            // transport's host barrier nor, with RCCL, for a step whose ncclAllReduce will never find its partner (the caller
            // aborts the communicators once every thread is back)
            int none = 0;
            if (cancel.compare_exchange_strong(none, 1, std::memory_order_acq_rel)) first_failed.store(i, std::memory_order_release);
            if (inproc) inproc_abort(inproc);
        }
    };
    std::vector<ccal_ctx_worker*> busy;
    busy.reserve((size_t)n);
    try {
        for (int i = 1; i < n; ++i) {
            ccal_ctx_worker* wk = ctx_worker(ps[i]->ctx);
            wk->submit([&run, i] { run(i); });
            busy.push_back(wk);
        }
    } catch (...) {          // a helper thread could not be made: the ranks already started wait for the missing one
        if (inproc) inproc_abort(inproc);
        for (ccal_ctx_worker* wk : busy) wk->wait();
        throw;
    }
    run(0);
    for (ccal_ctx_worker* wk : busy) wk->wait();
    // the shard that failed FIRST is the one to report (its peers then gave up with "a peer shard failed")
    int first_bad = first_failed.load(std::memory_order_acquire);
    for (int i = 0; i < n && first_bad < 0; ++i) if (!verdict(rc[i])) first_bad = i;
    if (first_bad >= 0) {
        if (first_bad != 0) note_error(ps[0]->ctx, (std::string("shard ") + std::to_string(first_bad) + ": " + ps[first_bad]->ctx->err).c_str());
        if (rep) { *rep = ccal_report{}; rep->status = rc[first_bad]; }
        return rc[first_bad];
    }
    // one decision sequence, one camera block: anything else is a defect of the transport (sums that differ between ranks)
    for (int i = 1; i < n; ++i) {
        if (rc[i] != rc[0] || reps[i].iterations != reps[0].iterations ||
            std::memcmp(&intr[i * ni], &intr[0], ni * sizeof(double)) != 0 || std::memcmp(&extr[i * ne], &extr[0], ne * sizeof(double)) != 0) {
            note_error(ps[0]->ctx, "ccal_solve_sharded: the shards disagree on the result (transport defect)");
            if (rep) { *rep = ccal_report{}; rep->status = CCAL_ERR_HIP; }
            return CCAL_ERR_HIP;
        }
    }
    std::memcpy(intr_io, &intr[0], ni * sizeof(double));
    if (extr_io) std::memcpy(extr_io, &extr[0], ne * sizeof(double));
    if (rep) {
        *rep = reps[0];
        for (int i = 1; i < n; ++i) rep->solve_ms = std::max(rep->solve_ms, reps[i].solve_ms);
    }
    return rc[0];
}
}  // namespace ccal
extern "C" {

int ccal_solve_sharded(ccal_problem** ps, int n, const ccal_solver_opts* o, double* intr_io, double** poses_io, double* extr_io,
                       ccal_report* rep) {
    if (!ps || n < 1 || !o || !intr_io) return CCAL_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i) if (!ps[i]) return CCAL_ERR_INVALID_ARG;
    ccal_ctx* c0 = ps[0]->ctx;
    CCAL_API_TRY
    int n_native = 0, n_cb = 0, n_peer = 0;
    for (int i = 0; i < n; ++i) {
        const ccal_problem* p = ps[i];
        if ((p->n_slots && !(poses_io && poses_io[i])) || (p->n_cams > 1 && !extr_io)) { c0->err = "ccal_solve_sharded: null parameter array"; return CCAL_ERR_INVALID_ARG; }
        for (int k = 0; k < i; ++k) if (ps[k]->ctx == p->ctx) { c0->err = "ccal_solve_sharded: every shard needs a context of its own (one host thread and one stream per shard)"; return CCAL_ERR_INVALID_ARG; }
        if (p->n_cams != ps[0]->n_cams || p->K != ps[0]->K || p->one_focal != ps[0]->one_focal || p->huber_delta != ps[0]->huber_delta) { c0->err = "ccal_solve_sharded: the shards describe different problems"; return CCAL_ERR_INVALID_ARG; }
        for (int c = 0; c < p->n_cams; ++c) if (p->cams[c].model != ps[0]->cams[c].model) { c0->err = "ccal_solve_sharded: the shards describe different cameras"; return CCAL_ERR_INVALID_ARG; }
        if (std::memcmp(p->lo.data(), ps[0]->lo.data(), p->lo.size() * sizeof(double)) || std::memcmp(p->hi.data(), ps[0]->hi.data(), p->hi.size() * sizeof(double)) ||
            p->has_bound != ps[0]->has_bound || p->fixed != ps[0]->fixed) { c0->err = "ccal_solve_sharded: the shards carry different bounds / fixed parameters"; return CCAL_ERR_INVALID_ARG; }
        n_native += p->rccl_comm ? 1 : 0; n_cb += (!p->rccl_comm && p->allreduce) ? 1 : 0; n_peer += p->peer ? 1 : 0;
    }
    if (n == 1 && !ps[0]->sharded()) return solve_entry(ps[0], o, true, intr_io, poses_io ? poses_io[0] : nullptr, extr_io, rep);
    if ((n_native && n_native != n) || (n_cb && n_cb != n) || (n_peer && n_peer != n) || (n_native > 0) + (n_cb > 0) + (n_peer > 0) > 1) { c0->err = "ccal_solve_sharded: a transport on some shards only"; return CCAL_ERR_INVALID_ARG; }
    if (n_peer) { c0->err = "ccal_solve_sharded: the shards of a ccal_multi_problem are solved by ccal_multi_solve"; return CCAL_ERR_INVALID_ARG; }
    if (n_native || n_cb) return solve_sharded_run(ps, n, o, intr_io, poses_io, extr_io, rep, nullptr);
    // no transport set: the library's own in-process one for the duration of the call (shards on one GPU, or on GPUs with
    // peer access); ccal_multi_* keeps a transport - RCCL when the devices differ - for the lifetime of the device set
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) devs[i] = ps[i]->ctx->device;
    std::string err;
    InprocComm* ic = inproc_create(n, devs.data(), &err);
    if (!ic) { c0->err = "ccal_solve_sharded: " + err; return CCAL_ERR_UNSUPPORTED; }
    for (int i = 0; i < n; ++i) ps[i]->peer = inproc_rank_handle(ic, i);
    inproc_set_timeout(ic, wait_timeout(ps[0], o));
    int rc = CCAL_ERR_HIP;
    try { rc = solve_sharded_run(ps, n, o, intr_io, poses_io, extr_io, rep, ic); } catch (...) { rc = CCAL_ERR_NO_MEMORY; }
    // the early-exit groups enqueued ahead reference the transport's events and buffers: drain them before it goes
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(ps[i]->ctx->device);
        (void)hipStreamSynchronize(ps[i]->ctx->stream);
        if (ps[i]->nws) { ps[i]->nws->tail_pending = false; if (ps[i]->nws->fws) ps[i]->nws->fws->tail_pending = false; }
        ps[i]->peer = nullptr;
    }
    inproc_destroy(ic);
    return rc;
    CCAL_API_CATCH(c0)
}

int ccal_solve(ccal_problem* p, const ccal_solver_opts* o, double* intr_io, double* poses_io, double* extr_io, ccal_report* rep) {
    if (!p || !o || !intr_io || (!poses_io && p->n_slots) || (!extr_io && p->n_cams > 1)) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    return solve_entry(p, o, true, intr_io, poses_io, extr_io, rep);
    CCAL_API_CATCH(p->ctx)
}

int ccal_solve_dev(ccal_problem* p, const ccal_solver_opts* o, ccal_report* rep) {
    if (!p || !o) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    return solve_entry(p, o, false, nullptr, nullptr, nullptr, rep);
    CCAL_API_CATCH(p->ctx)
}

}  // extern "C"
