// Device-resident optimizer state, the decision rule shared by both device loops, and the argument blocks of the
// single-camera fast path (ccal_kernels_fused.hip, driven by solve_fused() in ccal_solver.hip).
//
// One GROUP of launches = one evaluation + one reduced system + ONE all-reduce + one decision/solve, for Gauss-Newton
// and for Levenberg-Marquardt alike:
//     Gram at the evaluated set  ->  pose elimination (Schur)  ->  reduce  ->  [all-reduce]  ->  decide + camera solve
// GN accepts every step, so the system eliminated at the candidate is always the next one to solve.  LM does not know
// the next damping before the (all-reduced) cost of the candidate is known; it eliminates SPECULATIVELY with
// lambda_spec = the damping an accepted step gets when the gain ratio is near 1 (radius * 3, the trust-region rule's
// cap - the usual case near convergence).  A hit: the sums at hand are the next system, the step costs one group like
// GN.  A miss (another radius factor) or a rejected step: the decision sets `redo`, and the next group skips its Gram,
// re-eliminates the accepted set from its stored per-frame records with the right damping and solves - the same work
// the previous two-phase LM group did for every step, now only for those.  Every group carries exactly one
// collective, whatever the decisions: sharded ranks cannot disagree on the sequence.
#pragma once
#include "ccal_normal.hpp"

namespace ccal {

struct DevState {                 // lives in device memory; updated by the decision kernels only (sizeof % 8 == 0)
    double lambda;                // damping of the system to solve next (0 for GN)
    double lambda_spec;           // LM: damping of the speculative elimination at the candidate
    double lambda_solve;          // damping that produced the current dc (model decrease of the pose blocks)
    double radius, dec;
    double cur_cost, last_cost, initial_cost;
    double mc_cam;                // model decrease of the camera block for the current dc
    double min_error, min_abs, min_rel;
    int32_t cur;                  // parameter set (0/1) holding the accepted point
    int32_t first;                // 1 until the starting point has been evaluated
    int32_t redo;                 // 1: the next group re-eliminates the accepted set with `lambda` (no evaluation, no decision)
    int32_t done;                 // 0 = running, else ccal_status + 1
    int32_t iter, max_iter, method;
    int32_t lm_accepted, lm_rejected;
    int32_t sys_failed;           // a pose block of the system that produced the candidate was not positive definite (any rank)
    int32_t cam_failed;           // the camera system that produced the candidate was not positive definite
    int32_t done_seq;             // sequence number of the step that set `done` (0 while running)
    int32_t spec_hits, spec_misses;   // LM: accepted steps whose speculative elimination was / was not the next system
    int32_t error_metric;         // ccal_solver_opts::error_metric: what the stop rules compare (0: cost, 1: its square root)
    int32_t pad_;
};
static_assert(sizeof(DevState) % 8 == 0, "DevState is staged as doubles");

struct HostStatus {               // pinned, host-coherent; written at the end of a decision kernel
    // what the polling host needs after EVERY group, in one 8-byte store (no fence, no second word to order against):
    // bits 0-23 sequence number of the group, 24-31 `done` (0 = running, else ccal_status + 1), 32-55 the sequence number
    // of the step that set `done` - the host acts on `done` only once it has waited for that step
    volatile uint64_t word;
    // the rest is published (behind a system-scope fence, before `word`) only by the group that finishes the solve
    volatile int32_t iter, cur, lm_accepted, lm_rejected;
    volatile int32_t spec_hits, spec_misses;
    volatile double cur_cost, initial_cost, radius;
};
constexpr int kMaxGroups = (1 << 23);             // sequence numbers fit the 24-bit fields of HostStatus::word
__host__ __device__ inline uint64_t status_word(int seq, int done, int done_seq) {
    return (uint64_t)(uint32_t)seq | ((uint64_t)(uint32_t)(done & 0xff) << 24) | ((uint64_t)(uint32_t)done_seq << 32);
}
__host__ __device__ inline int status_seq(uint64_t w) { return (int)(w & 0xffffff); }
__host__ __device__ inline int status_done(uint64_t w) { return (int)((w >> 24) & 0xff); }
__host__ __device__ inline int status_done_seq(uint64_t w) { return (int)((w >> 32) & 0xffffff); }

// What an accepted LM step whose gain ratio is near 1 does to the radius: the rule's cap.  The same expression as in
// optimizer_decide, so that a hit is a bitwise comparison.
__host__ __device__ inline double lm_radius_cap(double radius) { return fmin(1e16, radius / (1.0 / 3.0)); }

// The optimizer's decision for one group, from all-reduced sums only (identical on every rank of a sharded solve):
//   cost      sum rho'(s) s of the evaluated set (the starting point in the first group, the candidate afterwards)
//   mc_pose   model decrease of the pose blocks for the step that produced the candidate
//   sys_fail  a pose block of the system at hand was not positive definite (count over all ranks > 0)
// tiny-solver's Gauss-Newton rules (src/util.rs:455: accept unconditionally; stop on min_error, |d| < 1e-5,
// |d| / last < 1e-5, max_iterations) or the Ceres-style trust region of the LM mode.
// Returns true when the reduced system at hand is the one to solve now.
// The quantity the stop rules compare (ccal_solver_opts::error_metric): tiny-solver's compute_error is the squared L2 norm of the
// loss-corrected residuals - or, the other reading of the absent crate, the norm itself.  Metric 0 leaves every expression as it was.
__host__ __device__ inline double error_of(double cost, int metric) { return metric ? sqrt(fmax(cost, 0.0)) : cost; }
// LM's predicted decrease in the same metric
__host__ __device__ inline double model_decrease_of(double cur_cost, double mc, int metric) {
    return metric ? sqrt(fmax(cur_cost, 0.0)) - sqrt(fmax(cur_cost - mc, 0.0)) : mc;
}
__device__ inline bool optimizer_decide(DevState* st, double cost, double mc_pose, bool sys_fail, int seq) {
    const bool lm = st->method == CCAL_METHOD_LM;
    const int em = st->error_metric;
    int done = 0;
    bool solve = false;
    if (st->redo) {                                   // re-elimination group: the decision was taken one group ago
        st->redo = 0;
        solve = true;
    } else if (st->first) {
        st->cur_cost = cost; st->initial_cost = cost; st->first = 0;
        if (!(cost == cost)) done = CCAL_ERR_NONFINITE + 1;
        else if (!(fabs(cost) < 1.7e308)) done = (lm ? CCAL_ERR_NONFINITE : CCAL_ERR_NOT_PD) + 1;
        solve = true;
    } else if (!lm) {
        // Gauss-Newton: the candidate is the new point (tiny-solver applies dx unconditionally)
        st->cur ^= 1;
        const double last = st->cur_cost, cur = cost;
        st->last_cost = last; st->cur_cost = cur; st->iter += 1;
        const double le = error_of(last, em), ce = error_of(cur, em);
        if (ce < st->min_error) done = CCAL_OK + 1;
        else if (!(cur == cur)) done = CCAL_ERR_NONFINITE + 1;
        else if (!(fabs(cur) < 1.7e308)) done = CCAL_ERR_NOT_PD + 1;
        else if (fabs(le - ce) < st->min_abs) done = CCAL_OK + 1;
        else if (fabs(le - ce) / le < st->min_rel) done = CCAL_OK + 1;
        else if (st->iter >= st->max_iter) done = CCAL_ERR_NO_CONVERGENCE + 1;
        solve = true;
    } else {
        const double mc = st->mc_cam + mc_pose;
        const double rho = (st->cur_cost - cost) / mc;
        st->iter += 1;
        const bool lin_fail = st->sys_failed || st->cam_failed;      // the candidate came out of a failed linear solve
        const bool fin = fabs(cost) < 1.7e308;
        const double mce = model_decrease_of(st->cur_cost, mc, em);
        if (!lin_fail && fin && mc >= 0.0 && (mce < st->min_abs || mce < st->min_rel * error_of(st->cur_cost, em))) {
            // predicted decrease below the thresholds: converged (the re-weighted Huber cost is not monotone at the optimum)
            if (cost < st->cur_cost) { st->cur ^= 1; st->last_cost = st->cur_cost; st->cur_cost = cost; st->lm_accepted += 1; }
            done = CCAL_OK + 1;
        } else if (!lin_fail && fin && mc > 0.0 && rho > 0.0) {
            st->cur ^= 1;
            const double last = st->cur_cost, cur = cost;
            st->last_cost = last; st->cur_cost = cur; st->lm_accepted += 1;
            const double t = 2.0 * rho - 1.0;
            st->radius = fmin(1e16, st->radius / fmax(1.0 / 3.0, 1.0 - t * t * t));
            st->dec = 2.0;
            const double le = error_of(last, em), ce = error_of(cur, em);
            if (ce < st->min_error) done = CCAL_OK + 1;
            else if (fabs(le - ce) < st->min_abs) done = CCAL_OK + 1;
            else if (fabs(le - ce) / le < st->min_rel) done = CCAL_OK + 1;
            st->lambda = 1.0 / st->radius;
            if (!done) {
                if (st->lambda == st->lambda_spec) { solve = true; st->spec_hits += 1; }      // the sums at hand are the next system
                else { st->redo = 1; st->spec_misses += 1; }
            }
        } else {
            st->lm_rejected += 1;
            st->radius /= st->dec; st->dec *= 2.0;
            if (st->radius < 1e-32) done = CCAL_ERR_NO_CONVERGENCE + 1;
            st->lambda = 1.0 / st->radius;
            if (!done) st->redo = 1;
        }
        st->sys_failed = 0; st->cam_failed = 0;
        if (!done && st->iter >= st->max_iter) done = CCAL_ERR_NO_CONVERGENCE + 1;
    }
    if (!done && solve && sys_fail) {
        // the system about to be solved has a pose block that is not positive definite: Gauss-Newton has no step
        // (tiny-solver: None); LM solves with that block frozen and rejects the candidate at the next decision
        if (!lm) done = CCAL_ERR_NOT_PD + 1;
        else st->sys_failed = 1;
    }
    if (lm) st->lambda_spec = 1.0 / lm_radius_cap(st->radius);
    st->done = done;
    if (done && !st->done_seq) st->done_seq = seq;
    return solve && !done;
}

// `full`: publish the whole report (the finishing group always does; verbose solves do after every group).  Otherwise one
// 8-byte store: a system-scope fence costs ~3 us of a 10-us single-wavefront kernel.
__device__ inline void publish_host_status(HostStatus* hs, const DevState* s, int seq, bool full = false) {
    if (full || s->done) {
        hs->iter = s->iter; hs->cur = s->cur;
        hs->lm_accepted = s->lm_accepted; hs->lm_rejected = s->lm_rejected;
        hs->spec_hits = s->spec_hits; hs->spec_misses = s->spec_misses;
        hs->cur_cost = s->cur_cost; hs->initial_cost = s->initial_cost; hs->radius = s->radius;
        __threadfence_system();
    }
    hs->word = status_word(seq, s->done, s->done_seq);      // the host polls this word; kernel completion flushes it at the latest
}

// Deterministic sum of the elimination kernel's per-workgroup partial sums, laid out [workgroup][entry]: a 256-thread
// workgroup sums 64 consecutive entries - lane = entry (coalesced rows, every load independent of the last), the four
// wavefronts take the four contiguous quarters of the rows (four interleaved accumulators each), combined in LDS in a
// fixed order.  The same function - hence the same order, bit for bit - whether k_reduce1 runs it (one workgroup per 64
// entries; the sharded loop all-reduces its result) or k_head does before it decides (single-GPU loop: one launch less).
__device__ __forceinline__ double reduce_partial_rows(const double* partial, int n_part, int rb, int e, double (*sh)[64]) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (n_part + 3) >> 2, r0 = wv * per, r1 = min(n_part, r0 + per);
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (e < rb) {
        const double* src = partial + e;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            v0 += src[(int64_t)r * rb]; v1 += src[(int64_t)(r + 1) * rb]; v2 += src[(int64_t)(r + 2) * rb]; v3 += src[(int64_t)(r + 3) * rb];
        }
        for (; r < r1; ++r) v0 += src[(int64_t)r * rb];
    }
    sh[wv][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    const double t = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    __syncthreads();
    return t;                                     // every wavefront holds the sum of entry e in lane (e & 63)
}

// Which parameter / record set a group's kernels work on, and with which damping.
__device__ __forceinline__ int eval_set(const DevState* st) { return st->first ? st->cur : (st->cur ^ 1); }
__device__ __forceinline__ int schur_set(const DevState* st) { return st->redo ? st->cur : eval_set(st); }
__device__ __forceinline__ double schur_lambda(const DevState* st) {
    return (st->method == CCAL_METHOD_LM && !st->first && !st->redo) ? st->lambda_spec : st->lambda;
}

// Single-launch groups (round 4; session sizes - the launcher decides, fused_iter_rows): k_gram1v<.., ITER> sums the PREVIOUS
// launch's rows of partial sums itself (every workgroup, the same rows in the same order), takes the decision and solves the
// camera system in front of its own evaluation (every workgroup the same arithmetic on the same sums: the same decision, the
// same candidate, bit for bit; workgroup 0 alone writes state, candidate, status word and result).  A group is ONE launch
// instead of Gram + reduce + head: two kernel boundaries (~3 us each) less per step.  What a launch reads and what the same
// launch's workgroup 0 writes must not alias - another workgroup may start later: the state and the rows alternate between
// two buffers by launch number.
constexpr int kFusedMaxK = 9;     // the single-camera loop: at most 9 camera columns (OPENCV5, two focal lengths)
struct IterArgs {
    int32_t on;                    // this launch is of the single-launch form
    int32_t skip_head;             // the solve's first launch: no sums to decide on yet, the state passes through
    int32_t seq, publish_all;
    const DevState* st_in; DevState* st_out;
    const double* partial_in; int32_t n_part_in;       // the previous launch's rows (this launch's go to FusedArgs::partial)
    HostStatus* hs; const ColInfo* cols; double* dc_out;
    double* result_host; int64_t np6;                  // see HeadArgs
    double* result_poses; int32_t* done_cnt;           // != NULL: EVERY workgroup of the finishing launch writes its slice of the poses there (head_finish, SPREAD)
    // The solve's FIRST launch as its own k_unpack1 (`fold`; every slot has an observation frame): the starting state, the column
    // table and the intrinsics ride in THIS argument block, every frame's lanes read their slot's starting pose where the caller
    // left it (pinned host memory, or parameter set 0 for device-resident solves) and write it to the parameter sets; workgroup 0
    // writes state, columns and intrinsics.  One launch (~5 us) and the host's gap behind it (~5 us) less per solve.
    int32_t fold, n_cols, poses_on_device;
    const double* poses_src; ColInfo* cols_out;
    DevState st0; ColInfo col0[kFusedMaxK]; double intr_h[CCAL_PMAX];
};

// A workgroup's row of partial sums in the single-launch form: the two symmetric blocks as their upper triangles (the mirrored
// entries are the same sums bit for bit: 58 doubles instead of 100 for six camera columns - every workgroup reads every row),
//   [A_dir (i <= j) | Y^T Y (i <= j) | mc_pose | failed pose blocks]
__host__ __device__ constexpr int iter_row_len(int K) { return (K + 1) * (K + 2) + 2; }
// packed entry -> entry of the full row [A_dir (K1 x K1) | Y^T Y (K1 x K1) | mc_pose | failed] (fused_red_size)
__host__ __device__ inline int iter_row_src(int K, int pe) {
    const int K1 = K + 1, T = K1 * (K1 + 1) / 2, NA = K1 * K1;
    if (pe >= 2 * T) return 2 * NA + (pe - 2 * T);
    const int blk = pe >= T ? 1 : 0;
    int t = pe - blk * T, i = 0;
    while (t >= K1 - i) { t -= K1 - i; ++i; }            // row i holds K1 - i entries (j = i .. K)
    return blk * NA + i * K1 + (i + t);
}
// Every workgroup (256 threads) of a single-launch group sums ALL rows of the launch before - the same rows in the same order in
// every workgroup: lane = packed entry, the four wavefronts take the four contiguous quarters of the rows, 48 loads in flight per
// lane (with 150 workgroups on the same lines an L2 round trip is ~1.3 us: the loop of reduce_partial_rows - four loads, wait,
// add - took 5.3 us for 157 rows), four interleaved accumulators, combined in a fixed order; the full (mirrored) layout goes to
// `red` in LDS, where head_wave expects it.
// Two halves, so that the caller can put loads of its own between them (issued while these sums' loads are in flight, consumed
// after the decision): iter_reduce_load = this wavefront's quarter of the rows into four accumulators per 64 packed entries;
// iter_reduce_combine = the four wavefronts' sums through LDS (two workgroup barriers).
template <int K>
__device__ __forceinline__ void iter_reduce_load(const double* partial, const int n_rows, double (*v)[4]) {
    constexpr int PROW = iter_row_len(K), NCH = (PROW + 63) / 64, B = NCH == 1 ? 48 : 24;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (n_rows + 3) >> 2, r0 = wv * per, r1 = min(n_rows, r0 + per);
#pragma unroll
    for (int c = 0; c < NCH; ++c) { v[c][0] = 0.0; v[c][1] = 0.0; v[c][2] = 0.0; v[c][3] = 0.0; }
    for (int r = r0; r < r1; r += B) {
        double t[NCH][B];
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const double* row = partial + (int64_t)min(r + i, r1 - 1) * PROW;       // (past the end: a valid row, not added)
#pragma unroll
            for (int c = 0; c < NCH; ++c) t[c][i] = row[min(c * 64 + lane, PROW - 1)];
        }
#pragma unroll
        for (int i = 0; i < B; ++i) {
            if (r + i < r1) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) v[c][i & 3] += t[c][i];
            }
        }
    }
}
template <int K>
__device__ __forceinline__ void iter_reduce_combine(const double (*v)[4], double* red, double (*sh)[(iter_row_len(K) + 63) / 64][64]) {
    constexpr int PROW = iter_row_len(K), NCH = (PROW + 63) / 64, K1 = K + 1, NA = K1 * K1;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wv < 4) {                     // (workgroups of eight wavefronts - k_gram2i - sum with their first four, like everybody: the same order)
#pragma unroll
        for (int c = 0; c < NCH; ++c) sh[wv][c][lane] = (v[c][0] + v[c][1]) + (v[c][2] + v[c][3]);
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int pe = c * 64 + lane;
            if (pe < PROW) {
                const double sum = (sh[0][c][lane] + sh[1][c][lane]) + (sh[2][c][lane] + sh[3][c][lane]);
                const int e = iter_row_src(K, pe);
                red[e] = sum;
                if (e < 2 * NA) {                      // the mirrored entry
                    const int blk = e >= NA ? 1 : 0, q = e - blk * NA, i = q / K1, j = q - i * K1;
                    red[blk * NA + j * K1 + i] = sum;
                }
            }
        }
    }
    __syncthreads();
}

struct FusedArgs {
    const float* x; const float* y; const float* z; const float* u; const float* v;
    const int64_t* obs_off; const int32_t* obs_slot;
    int32_t n_obs, K, PF, PRAW, n_pw;
    int32_t n_part;                // workgroups of the elimination kernel = partial sums per entry (set by the launcher)
    double huber_delta, min_diag, max_diag;
    ModelRt rt;                    // the context's run-time conventions (KB4 threshold, OPENCV5 coefficient order)
    double* intr[2]; double* poses[2]; double* pf[2]; double* praw[2];
    double* fcbuf; double* mc_f; double* cost_f;
    const double* dc; const DevState* st;
    double* partial; double* red;
    // general (multi-camera) loop, one camera's blocks through the register Gram kernels (GEN): the camera's observation
    // frames, where each frame's record goes inside the Gram buffer praw[set] (doubles), the camera and the extrinsics sets
    const int32_t* list; const int64_t* rec_off;
    int32_t cam; const double* extr[2];
    int32_t avg_corners;           // corners per observation frame on average (lanes-per-frame choice of the Gram launchers)
    int32_t slot_ident;            // obs_slot[f] == f for every frame (one camera that saw every slot): the prologue does not wait for the table
    // single-camera loop: k_gram1w eliminates the pose blocks of its frames itself (gram_fused_tail): no k_schur1m launch,
    // one row of partial sums per wavefront.  Set by launch_gram1v when the kernel it picks supports it (elim_fused = 1)
    int32_t fuse_elim, elim_fused;
    int32_t part_cap;              // rows the partial-sum buffer holds
    // GEN, all cameras of a rig in ONE launch (same model and focal mode): the camera of every observation frame; `list` then
    // holds every camera's frames, `intr` / `extr` point at camera 0.  NULL: one camera per launch (`cam`)
    const int32_t* obs_cam;
    int32_t lpf_force;             // lanes per frame forced by a developer switch of the second library (0: the launchers' cost model)
    // ccal_solve_batch: how many problems are being solved side by side on this GPU (contexts of the batch; 0 / 1: alone).  The
    // lanes-per-frame cost model then counts 1 / share of the chip's SIMDs as this problem's: fewer, longer wavefronts per launch
    // (625 frames alone: one frame per wavefront, 625 wavefronts that each hold a SIMD; eight side by side: five frames each)
    int32_t share;
    IterArgs it;
    // general (multi-camera) loop, GEN kernels: the candidate pose of the frame's slot is formed HERE, in the prologue - k_backsub's
    // work (dp = -L^-T (y_r + Y dc), model decrease of the pose block) with the slot's elimination record of the general loop and
    // the reduced system's camera step - instead of by a launch of its own behind k_solve (gen_backsub_pose, ccal_gram_common.hpp);
    // the slot's first observation frame (g_owner) writes the candidate pose and the model decrease
    int32_t gen_backsub, g_K, g_PF;
    const double* g_pf; const double* g_dc; double* g_mc_slot; const int8_t* g_owner;
    // ragged single-camera problems (k_gram2b): the sorted table - int4 { frame, first corner, corners, slot } per position - and the bins
    const int32_t* bin_tab;
    int32_t n_bins, bin_lpf[kGramMaxBins], bin_first[kGramMaxBins], bin_count[kGramMaxBins], bin_wg0[kGramMaxBins + 1];
};

// per-frame record of the single-camera Gram kernels (doubles), rotation columns in the phi basis:
//   C = H_pp packed lower (21) | [B = H_pc | g_p] (6 x (K+1)) | A = [J_c | r]^T W [J_c | r] ((K+1)^2) | J_l, the frame's left Jacobian (9)
__host__ __device__ constexpr int praw_jl_off(int K) { return 21 + 6 * (K + 1) + (K + 1) * (K + 1); }
__host__ __device__ constexpr int praw_size(int K) { return (praw_jl_off(K) + 9 + 1) & ~1; }
// GEN record of one observation frame (general loop), laid out for k_schur's expansion: every row it multiplies with E is
// six contiguous doubles:  C (6 x 6, full symmetric) | [B|g]^T (K1 x 6: row = camera column, r last) | A (K1 x K1) |
// E^T dense (12 x 6, frame_setup_composed);  K = the camera's P_eff
// (round 3: A as its packed lower triangle - row i, column j <= i at i (i + 1) / 2 + j - and E as its 36 structural non-zeros
//  E_c = [M | N | J1 | SJ], each 3 x 3 with X(m, b) at 3 b + m:  E^T row b = [M(.,b) 0], row 3+b = [0 N(.,b)], row 6+b = [J1(.,b) SJ(.,b)],
//  row 9+b = [0 e_b]  (frame_setup_composed); 142 doubles for EUCM where the dense form had 230: the records are written once
//  and read once per group, 37 -> 23 MB each way for two cameras x 10 000 frames)
__host__ __device__ constexpr int gen_a_off(int K) { return 36 + 6 * (K + 1); }
__host__ __device__ constexpr int gen_e_off(int K) { return (36 + 6 * (K + 1) + (K + 1) * (K + 2) / 2 + 1) & ~1; }    // 16-byte aligned
constexpr int GEN_EC = 36;
__host__ __device__ constexpr int gen_rec_size(int K) { return gen_e_off(K) + GEN_EC; }
// entry e (row-major 12 x 6) of the dense E^T from the compact form
__device__ __forceinline__ double gen_et_dense(const double* ec, int e) {
    const int row = e / 6, k = e - 6 * row, blk = row / 3, b = row - 3 * blk;
    if (blk == 0) return k < 3 ? ec[3 * b + k] : 0.0;
    if (blk == 1) return k >= 3 ? ec[9 + 3 * b + (k - 3)] : 0.0;
    if (blk == 2) return k < 3 ? ec[18 + 3 * b + k] : ec[27 + 3 * b + (k - 3)];
    return (k - 3) == b ? 1.0 : 0.0;
}
// where stored value v (0 .. 35) sits in the dense (row-major 12 x 6) E^T
__host__ __device__ constexpr int gen_et_pos(int v) {
    const int blk = v / 9, r = v - 9 * blk, b = r / 3, m = r - 3 * b;
    return blk == 0 ? 6 * b + m : (blk == 1 ? 6 * (3 + b) + 3 + m : (blk == 2 ? 6 * (6 + b) + m : 6 * (6 + b) + 3 + m));
}
// the compact form from frame_setup_composed's dense E^T
__device__ __forceinline__ void gen_et_compact(const double* ept, double* ec) {
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            ec[3 * b + m] = ept[6 * b + m]; ec[9 + 3 * b + m] = ept[6 * (3 + b) + 3 + m];
            ec[18 + 3 * b + m] = ept[6 * (6 + b) + m]; ec[27 + 3 * b + m] = ept[6 * (6 + b) + 3 + m];
        }
}
// rows of partial sums up to which k_head adds them up itself (session-sized problems: <= 1 280 frames); see solve_fused
constexpr int kHeadReduceRows = 40;
// red / partial rows of the single-camera path: [A_dir (K1 x K1) | Y^T Y (K1 x K1) | mc_pose | failed pose blocks]
__host__ __device__ constexpr int fused_red_size(int K) { return 2 * (K + 1) * (K + 1) + 2; }

struct HeadArgs {
    DevState* st; HostStatus* hs; const double* red; const ColInfo* cols;
    double* intr[2]; double* dc;
    int32_t K, seq;
    double min_diag, max_diag;
    // session-sized solves with host pointers: the group that finishes the solve writes the result - intrinsics and the
    // accepted poses - straight into pinned host memory before it publishes `done`: ccal_solve then returns from its poll
    // with no copy and no synchronise (result_host == NULL: the host fetches the result itself)
    double* result_host; const double* poses[2]; int64_t np6;
    // single-GPU loop: the head reduces the elimination kernel's partial sums itself (no k_reduce1 launch);
    // partial == NULL: `red` holds the (all-)reduced sums
    const double* partial; int32_t n_part;
    int32_t publish_all;           // verbose solves: the whole report after every group
    // single-process sharded solves over the in-process transport (ccal_multi.hip): every rank's reduced sums of this step, rank
    // order; the head adds them itself (the same order - the same bits - on every rank).  peers.n == 0: `red` / `partial` as above
    PeerView peers;
};

// one element of a peer rank's buffer: a system-scope load that bypasses this GPU's caches (the buffer lives on another device
// when the shards sit on different GPUs; what an earlier step left of the same address in L2 must not be served again)
__device__ __forceinline__ double ld_peer(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ double peer_sum(const PeerView& pv, const int e) {
    double v = ld_peer(pv.src[0] + e);
    for (int q = 1; q < pv.n; ++q) v += ld_peer(pv.src[q] + e);
    return v;
}

struct UnpackArgs {               // the starting point of a single-camera solve
    // the small part travels in the kernel-argument block itself (~600 bytes: no staging copy, no dependent load)
    DevState st0; ColInfo col0[kFusedMaxK]; double intr_h[CCAL_PMAX]; int32_t n_cols;
    const double* poses_src;      // ccal_solve: the caller's poses in pinned host memory (read by the kernel: zero-copy) or their
    int64_t np6;                  // staged device copy (large problems); ccal_solve_dev: unused
    int32_t poses_on_device;      // ccal_solve_dev: the starting point is already in intr0 / poses0, only state + columns arrive
    double* intr0; double* intr1; double* poses0; double* poses1;
    DevState* st; ColInfo* cols;
};
hipError_t launch_unpack1(const UnpackArgs& a, hipStream_t s);
hipError_t launch_gram1(int model, bool one_focal, const FusedArgs& a, hipStream_t s);     // MFMA Gram (any model)
hipError_t launch_gram1v(int model, bool one_focal, FusedArgs& a, hipStream_t s);    // register (VALU) Gram, any model
// rows of partial sums (= workgroups) a single-launch group of this problem would have; 0: that form does not apply
int fused_iter_rows(int model, bool one_focal, int n_obs, int avg_corners, int K, int share, bool batch = false);     // batch: a member of a lockstep batch (OPENCV5 takes the form only there)
hipError_t launch_gram_iter(int model, bool one_focal, FusedArgs& a, hipStream_t s);  // k_gram1v<.., ITER>; a.it filled in
int fused_iter_lpf(int n_obs, int avg_corners, int share);                           // the lane mapping of that launch
// ccal_solve_batch: launch number s_no of a whole batch of session-sized problems (one model, focal mode and lane
// mapping); tab: n argument blocks in DEVICE memory with it.st_in = the first of the two state buffers, partial = the partial-sum
// buffer's base, n_part = the problem's rows, it.fold = the first launch unpacks; max_rows = the largest n_part
hipError_t launch_gram_iter_batch(int model, bool one_focal, int lpf, const FusedArgs* tab, int n, int max_rows, int s_no, hipStream_t s);
hipError_t launch_gram1v_general(int model, bool one_focal, const FusedArgs& a, hipStream_t s);   // the same for camera 0 of the general loop
// ccal_kernels_gram2.hip: a corner's two rows on two lanes (row-local columns), same records, same fused tail
hipError_t launch_gram2(int model, bool one_focal, FusedArgs& a, hipStream_t s);
// single-launch groups on the two-wavefronts-per-SIMD kernel (k_gram2i: UCM / EUCM, 2 000 .. ~10 000 frames): rows (= workgroups of eight
// wavefronts) such a launch of this problem has, 0: the form does not apply; launch: a.it filled in as for launch_gram_iter
int gram2_iter_rows(int model, bool one_focal, int n_obs, int avg_corners, int K);
hipError_t launch_gram2_iter(int model, bool one_focal, FusedArgs& a, hipStream_t s);
hipError_t launch_gram2_general(int model, bool one_focal, const FusedArgs& a, hipStream_t s);
hipError_t launch_schur1m(FusedArgs& a, hipStream_t s);   // second library only (-DCCAL_LEGACY_KERNELS): the separate elimination launch; sets a.n_part
hipError_t launch_reduce1(const FusedArgs& a, hipStream_t s);
hipError_t launch_head(const HeadArgs& a, hipStream_t s);
hipError_t launch_state_eval(DevState* st, double lambda, hipStream_t s);     // state := "first evaluation of set 0 with this lambda"

}  // namespace ccal
