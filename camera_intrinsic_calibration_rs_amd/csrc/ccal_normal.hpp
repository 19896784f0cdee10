// Mode N (normal equations + exact per-frame Schur complement) and solver-loop workspaces.
//
// Pipeline of one linear solve (all on the context stream, everything stays in HBM):
//   k_gram    per observation frame: G_o = [J|r]^T W [J|r] with v_mfma_f64_16x16x4_f64   (heavy)
//   k_schur   per frame slot: assemble C = H_pp, B = [H_pc | g_p], eliminate the pose block,
//             accumulate the reduced system in per-wave LDS accumulators, keep L and Y = L^-1 B
//   k_reduce  deterministic sum of the per-wave partials -> red[RB]      (the all-reduce buffer)
//   k_solve   one workgroup: damping, fixed mask, Cholesky K x K, dc, clamp to bounds
//   k_backsub per slot: dp = -L^-T (y_r + Y dc), candidate poses, model-decrease partials
#pragma once
#include "ccal_internal.hpp"

#ifndef CCAL_GRAM_TILE
#define CCAL_GRAM_TILE 32
#endif
#ifndef CCAL_GRAM_MINW
#define CCAL_GRAM_MINW 1          // launch_bounds: minimum wavefronts per SIMD the Gram kernels are compiled for
#endif

namespace ccal {

// corners whose weighted rows are staged in LDS at a time by the Gram kernels (32 or 64)
constexpr int GRAM_TILE_CORNERS = CCAL_GRAM_TILE;

// layout of the reduced buffer red[RB]:  A[(K+1)*(K+1)] | hdiag[K] | gc[K] | cost | mc_pose | failed pose blocks
//   A = [[S, b],[b^T, *]] undamped in the camera block (pose damping already inside the Schur terms)
inline int red_size(int K) { return (K + 1) * (K + 1) + 2 * K + 3; }
// per-slot record pf[PF]: L (21, row-major lower, diagonal stored inverted) | Y[6][K+1] | g_p[6] | dC[6]
inline int pf_size(int K) { return (21 + 6 * (K + 1) + 12 + 1) & ~1; }

struct ColInfo {          // one column of the reduced camera system
    double lo, hi;
    int32_t has_bound, fixed;
    int32_t is_extr;      // 0: intrinsic (index into intr, full layout), 1: extrinsic (index into extr)
    int32_t dst, dst2;    // element index in the destination array; dst2 >= 0 mirrors the value (fy = f)
};

struct NormalWs {
    int K = 0, RB = 0, PF = 0, n_pw = 0;
    // the candidate poses are formed in the GEN Gram kernels' prologue (FusedArgs::gen_backsub) instead of by k_backsub: set by the
    // device-resident general loop for its launches (CCAL_GEN_BACKSUB=0: the separate launch); d_obs_owner[o] = 1 for the first
    // observation frame of its slot (the one that writes the candidate pose and the model decrease)
    bool gen_backsub = false; double lm_min_diag = 1e-6, lm_max_diag = 1e32;
    int8_t* d_obs_owner = nullptr;
    bool all_slots_observed = false;           // no slot without an observation frame: the two pose sets need not start equal
    bool general_ready = false;                // the general loop's buffers exist (normal_ws_ensure_general); the single-camera loop never asks
    int schur_wpb = 4;                         // wavefronts per workgroup of k_schur: 1 for reduced systems of 64 .. 127 columns
    bool schurq = false;                       // two cameras with equal blocks: elimination with four lanes per slot (k_schurq) instead of k_schur<true>
    int schurq_slots = 16;                     // k_schurq: frame slots per wavefront (16 = four lanes each, 8 = eight lanes each)
    int n_rows = 0;                            // rows of partial sums the elimination kernel in use writes = what k_reduce adds up
    double* G[2] = { nullptr, nullptr };       // per-observation-frame Gram blocks (current / candidate)
    double* cost_o[2] = { nullptr, nullptr };  // per-observation-frame cost
    int64_t* d_goff = nullptr;                 // [n_obs] offset of G_o
    int32_t* d_slot_off = nullptr;             // [n_slots+1] CSR slot -> observation frames
    int32_t* d_slot_obs = nullptr;
    int32_t* d_all_obs = nullptr;              // merged Gram launch: every camera's observation frames, camera-major
    bool merged_gram = false;                  // all cameras share model and focal mode: their blocks in ONE launch of the register Gram kernel
    // ragged frames (round 6): a Gram launch's list of observation frames sorted by corner count + the bins of the launch (gram2_bin_plan);
    // [0]: the merged launch, [1 + c]: camera c's own launch.  n_bins == 0: the plain launch over the list in table order
    GramBins gen_bins[1 + CCAL_MAX_CAMS];
    int32_t* d_gen_sorted[1 + CCAL_MAX_CAMS] = {};
    int64_t* d_slot_rec = nullptr;             // k_schurq: [n_slots][2] record offset of camera 0 / 1 in that slot, -1 = none
    int64_t* d_slot_desc = nullptr;            // [n_obs] in slot order: goff * 8 + camera
    int32_t* d_obs_cam = nullptr;
    int32_t* d_caminfo = nullptr;              // [n_cams][4]: Peff, col_theta, col_extr, NCP
    double* partial = nullptr;                 // [RB][n_pw]
    double* red = nullptr;                     // [2][RB + 8]: the reduced sums; the in-process transport alternates between the two
    double* red_out = nullptr;                 // where this group's k_reduce writes (NULL: red)
    PeerView peers = {};                       // this group's k_solve: every rank's sums (in-process transport), n == 0: red
    double* pf = nullptr;                      // [n_slots][PF]
    double* dc = nullptr;                      // [K]
    double* mc_slot = nullptr;                 // [n_slots]
    double* scal = nullptr;                    // [8]: 2 = model decrease of the camera block (host-driven form)
    int32_t* flags = nullptr;                  // [4]: 1 = camera Cholesky failed (host-driven ccal_build_normal form only)
    ColInfo* cols = nullptr;                   // [K]
    double* h_pinned = nullptr;                // pinned staging (RB + 16 doubles)
    int cur = 0;                               // which G buffer holds the current point
    bool red_fused = false;                    // the last ccal_build_normal_dev left its sums in fws->red (single camera)
    bool gstate_is_eval = false;               // d_gstate already says "first evaluation of set 0" (ccal_build_normal)
    bool register_gram = false;                // general loop: every camera's blocks come from k_gram1v / k_gram1w (GEN record format, caminfo NCP = 0)
    int64_t g_len = 0;
    struct DevState* d_gstate = nullptr;       // general loop: optimizer state on the device,
    struct HostStatus* h_gstatus = nullptr;    //   its published copy (pinned, host-coherent)
    struct DevState* h_gstate = nullptr;       //   and the pinned staging of its initial value
    hipStream_t side = nullptr;                // result download past the early-exit group enqueued ahead
    bool tail_pending = false;
    struct FusedWs* fws = nullptr;             // single-camera fused path (ccal_fused.hpp)
};

struct FusedWs {
    int PRAW = 0, RB1 = 0, n_pw = 0;
    char* d_block = nullptr;                   // the ONE device allocation the buffers below are slices of (fcbuf apart)
    char* h_block = nullptr;                   // the ONE pinned, host-coherent allocation: h_status | h_result | h_stage
    double* pf[2] = { nullptr, nullptr };
    double* praw[2] = { nullptr, nullptr };
    double* partial = nullptr;
    double* red = nullptr;                     // [2][RB1 + 7 rounded]: the in-process transport alternates between the two (red_stride apart)
    size_t red_stride = 0;
    double* fcbuf = nullptr;                   // -DCCAL_STAMPS builds only: in-kernel timestamps (NULL in the product build)
    double* mc_f = nullptr;                    // [n_obs] model decrease of each pose block
    double* cost_f = nullptr;                  // [n_obs] cost of each frame
    struct DevState* d_state = nullptr;
    bool state_is_eval = false; double state_eval_lambda = 0.0;     // d_state already says "first evaluation of set 0" with this damping
    bool fuse_elim = true;                     // the Gram kernels eliminate their frames' pose blocks in their tail (second library, CCAL_FUSE_ELIM=0: separate launch)
    struct HostStatus* h_status = nullptr;     // pinned, host-coherent
    double* h_stage = nullptr;                 // pinned staging of the caller's poses (read by k_unpack1 in place when small)
    double* d_stage = nullptr;                 // their device image (large problems: one copy per solve, k_unpack1 distributes it)
    double* h_result = nullptr;                // pinned, host-coherent: [intr | poses] written by the k_head that finishes a session-sized solve
                                               // (up to kSpreadBytes of poses: by every workgroup of a finishing single-launch group, head_finish SPREAD)
    int32_t* done_cnt = nullptr;               // SPREAD: the workgroups that have written their slice (zero between solves)
    hipStream_t side = nullptr;                // result download: does not queue behind the early-exit groups
    bool tail_pending = false;                 // early-exit groups of the previous solve may still be in flight
    int all_slots_observed = -1;               // every frame slot has an observation frame (-1: not looked yet): the first single-launch group may unpack for itself
};

struct DevState;
// argument block of the per-slot elimination kernels (k_schur: ccal_kernels_normal.hip, k_schurq: ccal_kernels_schurq.hip)
struct SchurArgs {
    // the record buffers of parameter set 0 / 1 of the device-resident loop (host-driven form: [0] only).  An ARRAY indexed with
    // the set - one scalar load at a computed offset - and not `set == 1 ? G2 : G`: hipcc -O1 (ROCm 7.2) compiled that select
    // of two kernel-argument pointers behind `st != NULL && ...` into a branch that skips the load of G (every general solve
    // ended NOT_PD on records of zeros); at -O3 the same source happened to come out right - and stopped doing so when the
    // block grew by a pointer (the "136-byte" failures of this round).  Found with a -O1 host-sanitizer build
    const double* Gs[2];
    const int64_t* slot_desc;      // k_schur: per (slot, observation) in slot order: goff * 8 + camera - one load instead of three;
                                   // k_schurq: [n_slots][2] record offset (doubles) of camera 0 / 1 in the slot, -1 = none
    const int32_t* slot_off; const int32_t* caminfo; int32_t n_cams;
    int32_t n_slots, K, RB, PF, n_pw;
    int32_t STG;                   // per-wavefront staging (doubles) for record-format observations, 0 = none
    double lambda, min_diag, max_diag;
    double* partial; double* pf; const double* mc_slot;
    const DevState* st;                        // device-resident loop: Gram set and lambda come from the state
};


void normal_ws_destroy(ccal_problem* p);
int normal_ws_ensure(ccal_problem* p);          // sizes, column table, camera step: allocate on first use
int normal_ws_ensure_general(ccal_problem* p);  // + the general (multi-camera) loop's buffers
int normal_upload_cols(ccal_problem* p);        // bounds / fixed flags -> device

// launchers (ccal_kernels_normal.hip).  Host-driven form: `cand` / gbuf / lambda select parameter set, G buffer and
// damping.  Device-resident form (st != NULL, *_dev): set 0 = (p->d_*, G[w->cur]), set 1 = (p->d_*_c, G[w->cur ^ 1]),
// the kernels pick the current set by st->cur, take lambda from st->lambda and do nothing once st->done is set.
struct DevState;
struct HostStatus;
hipError_t launch_gram(const ccal_problem* p, int cam, bool use_candidate_params, int gbuf, hipStream_t s);
hipError_t launch_gram_dev(const ccal_problem* p, int cam, const DevState* st, hipStream_t s);
hipError_t launch_gram_dev_all(const ccal_problem* p, const DevState* st, hipStream_t s);     // every camera: one launch if w->merged_gram, else one per camera
hipError_t launch_schur(const ccal_problem* p, int gbuf, double lambda, double min_diag, double max_diag, hipStream_t s,
                        const DevState* st = nullptr);
bool schurq_fits(int n_cams, const int* peff, const int* col_theta, const int* col_extr);      // ccal_kernels_schurq.hip
int schurq_slots_per_wave(int n_slots);         // 16 (four lanes per slot) or 8 (eight lanes per slot)
int schurq_rows(int n_slots, int slots_per_wave);
hipError_t launch_schurq(const SchurArgs& a, int peff, int rows, int slots_per_wave, hipStream_t s);
hipError_t launch_reduce(const ccal_problem* p, hipStream_t s, const DevState* st = nullptr);
hipError_t launch_solve(const ccal_problem* p, double lambda, double min_diag, double max_diag, hipStream_t s, DevState* st = nullptr,
                        HostStatus* hs = nullptr, int seq = 0, bool publish_all = false);
hipError_t launch_backsub(const ccal_problem* p, double lambda, double min_diag, double max_diag, hipStream_t s, const DevState* st = nullptr);

}  // namespace ccal
