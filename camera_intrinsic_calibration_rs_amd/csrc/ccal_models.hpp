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

}  // namespace ccal
