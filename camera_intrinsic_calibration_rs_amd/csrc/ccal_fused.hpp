// Device-resident optimizer state and argument blocks of the single-camera fast path
// (ccal_kernels_fused.hip, driven by solve_fused() in ccal_solver.hip).
#pragma once
#include "ccal_normal.hpp"

namespace ccal {

struct DevState {                 // lives in device memory; updated by k_head only (sizeof % 8 == 0)
    double lambda;                // damping of the NEXT pose elimination / camera solve (0 for GN)
    double lambda_solve;          // damping that produced the current dc (model decrease of the pose blocks)
    double radius, dec;
    double cur_cost, last_cost, initial_cost;
    double mc_cam;                // model decrease of the camera block for the current dc
    double min_error, min_abs, min_rel;
    int32_t cur;                  // parameter set (0/1) holding the accepted point
    int32_t first;                // 1 until the starting point has been evaluated
    int32_t done;                 // 0 = running, else ccal_status + 1
    int32_t iter, max_iter, method;
    int32_t lm_accepted, lm_rejected, accepted_now;
    int32_t done_seq;             // sequence number of the step that set `done` (0 while running)
};
static_assert(sizeof(DevState) % 8 == 0, "DevState is staged as doubles");

struct HostStatus {               // pinned, host-coherent; written at the end of k_head
    volatile int32_t seq;
    volatile int32_t done, iter, cur, lm_accepted, lm_rejected;
    volatile int32_t done_seq;    // which step finished the solve: the host acts on `done` only once it has waited for that step
    volatile double cur_cost, initial_cost, radius;
};

struct FusedArgs {
    const float* x; const float* y; const float* z; const float* u; const float* v;
    const int64_t* obs_off; const int32_t* obs_slot;
    int32_t n_obs, K, PF, PRAW, n_pw;
    double huber_delta, min_diag, max_diag;
    double* intr[2]; double* poses[2]; double* pf[2]; double* praw[2];
    double* fcbuf; double* mc_f; double* cost_f;
    const double* dc; const DevState* st; int32_t* st_flags;
    double* partial; double* red;
    int32_t* ticket;               // arrival counter of the fused reduce + head tail of k_schur1
};

struct HeadArgs {
    DevState* st; HostStatus* hs; const double* red; const ColInfo* cols; int32_t* flags;
    double* intr[2]; double* dc;
    int32_t K, phase, seq;
    double min_diag, max_diag;
};

struct UnpackArgs {               // staging block (doubles): [intr CCAL_PMAX | DevState | ColInfo x CCAL_KMAX | poses np6]
    const double* stage; int64_t small_doubles, np6;
    double* intr0; double* intr1; double* poses0; double* poses1;
    DevState* st; ColInfo* cols; int32_t* flags;
};
hipError_t launch_unpack1(const UnpackArgs& a, hipStream_t s);
hipError_t launch_gram1(int model, bool one_focal, const FusedArgs& a, hipStream_t s);     // MFMA Gram (any model)
hipError_t launch_gram1v(int model, bool one_focal, const FusedArgs& a, hipStream_t s);    // VALU Gram (<= 105 triangle entries)
hipError_t launch_schur1(const FusedArgs& a, int set_sel, const HeadArgs* fused_head, hipStream_t s);   // fused_head != NULL: last workgroup reduces + decides + solves
hipError_t launch_schur1m(const FusedArgs& a, int set_sel, hipStream_t s);   // four frames per wavefront; needs a.n_pw = 4 ceil(n_obs / 16)
hipError_t launch_reduce1(const FusedArgs& a, int first, int count, hipStream_t s);
hipError_t launch_cost1(const FusedArgs& a, hipStream_t s);
hipError_t launch_head(const HeadArgs& a, hipStream_t s);
hipError_t launch_state_eval(DevState* st, double lambda, hipStream_t s);     // state := "first evaluation of set 0 with this lambda"

}  // namespace ccal
