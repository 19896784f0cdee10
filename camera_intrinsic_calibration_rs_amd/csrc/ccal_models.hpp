// Model ids, block dimensions and the ONE table of model conventions shared by host and device code.
//
// The projection formulas of the path live in the crate camera-intrinsic-model 0.8 (Cargo.toml:25), whose source is
// not under /root/reference.  Everything this build ASSUMES about that crate is in this file; a maintainer with the
// crate at hand checks / flips it here (compile-time items) or through ccal_set_model_conventions (run-time items:
// distortion bounds and the KB4 branch threshold -- no kernel is recompiled for those).
#pragma once
#include <hip/hip_runtime.h>

namespace ccal {

constexpr int kUCM = 0, kEUCM = 1, kKB4 = 2, kOCV5 = 3;
constexpr int kNumModels = 4, kMaxDist = 5;

__host__ __device__ constexpr int model_np(int m) { return m == kUCM ? 5 : m == kEUCM ? 6 : m == kKB4 ? 8 : 9; }
// Jacobian width of one block: P_eff + 6 (camera 0) or + 12 (other cameras)
__host__ __device__ constexpr int block_dim(int model, bool one_focal, bool other) {
    return model_np(model) - (one_focal ? 1 : 0) + (other ? 12 : 6);
}

// Other-camera Gram blocks that fit one 16 x 16 matrix-core tile once the tvec_0_b columns (linear combinations of
// the tvec_c_0 columns) are left out: block_dim - 3 Jacobian columns + the residual <= 16.
__host__ __device__ constexpr bool gram_compact(int model, bool one_focal, bool other) {
    return other && block_dim(model, one_focal, true) - 2 <= 16;
}

// ---- the kernels' own (canonical) order of the OPENCV5 distortion coefficients ---------------------------------
// th[4 ..] = k1, k2, p1, p2, k3 inside every kernel and in the columns of its Jacobians / normal equations.  Where
// these sit in the CALLER's params() vector is a run-time convention (ccal_model_conventions.ocv5_order): load_theta
// gathers them, the camera solve scatters the step back, the API boundary permutes columns.
constexpr int OCV5_K1 = 4, OCV5_K2 = 5, OCV5_P1 = 6, OCV5_P2 = 7, OCV5_K3 = 8;
// what a kernel needs to know about the context's conventions beside the parameters themselves
struct ModelRt {
    double kb4_eps;                // KB4 small-radius threshold of project_one
    double unproject_eps;          // small-radius threshold of the unprojection
    uint32_t ocv5_perm;            // 3 bits per coefficient: position of k1, k2, p1, p2, k3 among the caller's five distortion slots
    uint32_t pad_;
};
constexpr uint32_t kOcv5IdentityPerm = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12);
__host__ __device__ constexpr int ocv5_pos(uint32_t perm, int i) { return (int)((perm >> (3 * i)) & 7u); }

// ---- run-time conventions (defaults; ccal_set_model_conventions overrides them per context) -------------------
// KB4 project_one: radius r = sqrt(x^2 + y^2) <= this -> pinhole limit (u = fx x / z + cx), else the atan polynomial
constexpr double kDefaultKb4SmallRadius = 1e-8;
// unprojection (pose initialisation, convert_model rays): image-plane radius below which the ray is the optical axis
constexpr double kDefaultUnprojectSmallRadius = 1e-8;
// distortion_params_bound() of the crate (src/util.rs:40-48 applies them at index 4 + i - shift): [model][i] lo / hi
constexpr double kDefaultDistLo[kNumModels][kMaxDist] = {
    { 1e-6, 0, 0, 0, 0 },               // UCM   alpha
    { 1e-6, 1e-6, 0, 0, 0 },            // EUCM  alpha, beta
    { -1.0, -1.0, -1.0, -1.0, 0 },      // KB4   k1..k4
    { -1.0, -1.0, -1.0, -1.0, -1.0 },   // OPENCV5 k1, k2, p1, p2, k3
};
constexpr double kDefaultDistHi[kNumModels][kMaxDist] = {
    { 1.0, 0, 0, 0, 0 },
    { 1.0, 100.0, 0, 0, 0 },
    { 1.0, 1.0, 1.0, 1.0, 0 },
    { 1.0, 1.0, 1.0, 1.0, 1.0 },
};

}  // namespace ccal
