// Model ids and block dimensions shared by host and device code.
#pragma once
#include <hip/hip_runtime.h>

namespace ccal {

constexpr int kUCM = 0, kEUCM = 1, kKB4 = 2, kOCV5 = 3;

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

}  // namespace ccal
