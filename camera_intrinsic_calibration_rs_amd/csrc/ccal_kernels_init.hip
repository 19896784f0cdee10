// Per-frame pose initialisation: what calib_camera does before it builds the
// problem -- `generic_camera.unproject(p2ds)`, keep the valid ones, divide by z, planar PnP
// (src/util.rs:418-436; the reference calls sqpnp_simple, a third-party crate).  Any PnP that lands in
// the same basin is equivalent after the joint solve; this one is the classic plane-induced homography:
//   normalised image point (xn, yn) = unproject(u, v) / z,   board point (X, Y, 0)
//   least squares for H = [h11 h12 h13; h21 h22 h23; h31 h32 1]   (8 x 8 normal equations per frame)
//   H ~ [r1 r2 t]  ->  scale, Gram-Schmidt, r3 = r1 x r2, quaternion -> rvec.
// One wavefront per observation frame; lanes own corners, 44 lane-private sums, one shuffle reduction.
#include "ccal_device.hpp"
#include "ccal_internal.hpp"

namespace ccal {

// unproject + divide by z: the model inverse as published (UCM/EUCM closed form, Usenko et al. 2018;
// KB4 Newton on theta; OPENCV5 fixed-point undistortion).  Returns false where the reference's
// `unproject` yields None (outside the model's domain) or the ray is not in front of the camera.
template <int MODEL>
__device__ __forceinline__ bool unproject_normalized(const double* th, double small_radius, double u, double v, double& xn, double& yn) {
    const double mx = (u - th[2]) / th[0], my = (v - th[3]) / th[1];
    const double r2 = mx * mx + my * my;
    if constexpr (MODEL == kUCM || MODEL == kEUCM) {
        const double alpha = th[4], beta = (MODEL == kEUCM) ? th[5] : 1.0;
        if (alpha > 0.5 && r2 > 1.0 / (beta * (2.0 * alpha - 1.0))) return false;
        const double t1 = 1.0 - (2.0 * alpha - 1.0) * beta * r2;
        if (t1 < 0.0) return false;
        const double k = (1.0 - alpha * alpha * beta * r2) / (alpha * sqrt(t1) + (1.0 - alpha));
        if (!(k > 1e-3)) return false;
        xn = mx / k; yn = my / k;
        return true;
    } else if constexpr (MODEL == kKB4) {
        const double r = sqrt(r2);
        if (r < small_radius) { xn = mx; yn = my; return true; }
        double t = r;
        for (int it = 0; it < 10; ++it) {
            const double t2 = t * t;
            const double f = t * (1.0 + t2 * (th[4] + t2 * (th[5] + t2 * (th[6] + t2 * th[7])))) - r;
            const double fp = 1.0 + t2 * (3.0 * th[4] + t2 * (5.0 * th[5] + t2 * (7.0 * th[6] + t2 * 9.0 * th[7])));
            t -= f / fp;
        }
        if (!(t > 0.0) || !(t < 1.5)) return false;             // theta < ~86 deg: in front of the camera
        const double s = tan(t) / r;
        xn = mx * s; yn = my * s;
        return true;
    } else {
        const double k1 = th[OCV5_K1], k2 = th[OCV5_K2], p1 = th[OCV5_P1], p2 = th[OCV5_P2], k3 = th[OCV5_K3];
        double x = mx, y = my;
        for (int it = 0; it < 25; ++it) {
            const double q = x * x + y * y;
            const double rad = 1.0 + q * (k1 + q * (k2 + q * k3));
            const double dx = 2.0 * p1 * x * y + p2 * (q + 2.0 * x * x);
            const double dy = p1 * (q + 2.0 * y * y) + 2.0 * p2 * x * y;
            x = (mx - dx) / rad; y = (my - dy) / rad;
        }
        // accept only if re-projection reproduces the input
        const double q = x * x + y * y, rad = 1.0 + q * (k1 + q * (k2 + q * k3));
        const double ex = x * rad + 2.0 * p1 * x * y + p2 * (q + 2.0 * x * x) - mx;
        const double ey = y * rad + p1 * (q + 2.0 * y * y) + 2.0 * p2 * x * y - my;
        if (!(fabs(ex) + fabs(ey) < 1e-9)) return false;
        xn = x; yn = y;
        return true;
    }
}

struct InitArgs {
    const float* x; const float* y; const float* z; const float* u; const float* v;
    const int64_t* obs_off; const int32_t* list; int32_t n_list, cam;
    const double* intr;
    double* poses_obs;      // [n_obs][6]  T_cam_board
    int32_t* valid_obs;     // [n_obs]     number of corners used, 0 = no pose
    int32_t min_points;
    ModelRt rt;             // the context's run-time conventions
};

template <int MODEL>
__global__ __launch_bounds__(256) void k_pose_init(const InitArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int widx = blockIdx.x * WAVES_PER_BLOCK + wave;
    if (widx >= a.n_list) return;
    const int o = __builtin_amdgcn_readfirstlane(a.list[widx]);
    const int64_t start = a.obs_off[o];
    const int n = (int)(a.obs_off[o + 1] - start);
    double th[th_len<MODEL>()];
    load_theta<MODEL, false>(a.intr + a.cam * CCAL_PMAX, a.rt, th);

    double M[36], rhs[8];            // upper triangle of A^T A (row-major packed) and A^T b
#pragma unroll
    for (int i = 0; i < 36; ++i) M[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) rhs[i] = 0.0;
    int cnt = 0, nonplanar = 0;
    for (int c = lane; c < n; c += 64) {
        const int64_t g = start + c;
        const double X = a.x[g], Y = a.y[g];
        if (a.z[g] != 0.0f) nonplanar = 1;
        double xn, yn;
        if (!unproject_normalized<MODEL>(th, a.rt.unproject_eps, (double)a.u[g], (double)a.v[g], xn, yn)) continue;
        ++cnt;
        const double r1[8] = { X, Y, 1.0, 0.0, 0.0, 0.0, -xn * X, -xn * Y };
        const double r2[8] = { 0.0, 0.0, 0.0, X, Y, 1.0, -yn * X, -yn * Y };
        int k = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = i; j < 8; ++j) { M[k] += r1[i] * r1[j] + r2[i] * r2[j]; ++k; }
            rhs[i] += r1[i] * xn + r2[i] * yn;
        }
    }
    // wave reduction
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < 36; ++i) M[i] += __shfl_xor(M[i], off, 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) rhs[i] += __shfl_xor(rhs[i], off, 64);
        cnt += __shfl_xor(cnt, off, 64);
        nonplanar |= __shfl_xor(nonplanar, off, 64);
    }
    if (lane != 0) return;
    double* out = a.poses_obs + (int64_t)o * 6;
    bool ok = cnt >= a.min_points && !nonplanar;
    // Cholesky of the 8x8 normal matrix (packed upper -> full lower), solve for h
    double L[8][8], h[8];
    if (ok) {
        int k = 0;
        for (int i = 0; i < 8; ++i) for (int j = i; j < 8; ++j) { L[j][i] = M[k]; ++k; }
        for (int j = 0; j < 8 && ok; ++j) {
            double s = L[j][j];
            for (int q = 0; q < j; ++q) s -= L[j][q] * L[j][q];
            if (!(s > 0.0)) { ok = false; break; }
            const double l = sqrt(s);
            L[j][j] = l;
            for (int i = j + 1; i < 8; ++i) { double t = L[i][j]; for (int q = 0; q < j; ++q) t -= L[i][q] * L[j][q]; L[i][j] = t / l; }
        }
    }
    if (ok) {
        for (int i = 0; i < 8; ++i) { double t = rhs[i]; for (int q = 0; q < i; ++q) t -= L[i][q] * h[q]; h[i] = t / L[i][i]; }
        for (int i = 7; i >= 0; --i) { double t = h[i]; for (int q = i + 1; q < 8; ++q) t -= L[q][i] * h[q]; h[i] = t / L[i][i]; }
        // H ~ [r1 r2 t]
        double c1[3] = { h[0], h[3], h[6] }, c2[3] = { h[1], h[4], h[7] };
        const double n1 = sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
        const double n2 = sqrt(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
        const double lam = 2.0 / (n1 + n2);
        const double t[3] = { lam * h[2], lam * h[5], lam };
        for (int i = 0; i < 3; ++i) c1[i] /= n1;
        const double d = c1[0] * c2[0] + c1[1] * c2[1] + c1[2] * c2[2];
        for (int i = 0; i < 3; ++i) c2[i] -= d * c1[i];
        const double n2b = sqrt(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
        ok = n2b > 1e-12 && n1 > 1e-12 && lam == lam;
        if (ok) {
            for (int i = 0; i < 3; ++i) c2[i] /= n2b;
            const double c3[3] = { c1[1] * c2[2] - c1[2] * c2[1], c1[2] * c2[0] - c1[0] * c2[2], c1[0] * c2[1] - c1[1] * c2[0] };
            // rotation matrix R = [c1 c2 c3] (columns) -> unit quaternion (largest component first) -> rvec
            const double R00 = c1[0], R10 = c1[1], R20 = c1[2], R01 = c2[0], R11 = c2[1], R21 = c2[2], R02 = c3[0], R12 = c3[1], R22 = c3[2];
            const double tr = R00 + R11 + R22;
            double qw, qx, qy, qz;
            if (tr >= R00 && tr >= R11 && tr >= R22) {
                const double s = 2.0 * sqrt(fmax(tr + 1.0, 1e-300));
                qw = 0.25 * s; qx = (R21 - R12) / s; qy = (R02 - R20) / s; qz = (R10 - R01) / s;
            } else if (R00 >= R11 && R00 >= R22) {
                const double s = 2.0 * sqrt(fmax(1.0 + R00 - R11 - R22, 1e-300));
                qw = (R21 - R12) / s; qx = 0.25 * s; qy = (R01 + R10) / s; qz = (R02 + R20) / s;
            } else if (R11 >= R22) {
                const double s = 2.0 * sqrt(fmax(1.0 + R11 - R00 - R22, 1e-300));
                qw = (R02 - R20) / s; qx = (R01 + R10) / s; qy = 0.25 * s; qz = (R12 + R21) / s;
            } else {
                const double s = 2.0 * sqrt(fmax(1.0 + R22 - R00 - R11, 1e-300));
                qw = (R10 - R01) / s; qx = (R02 + R20) / s; qy = (R12 + R21) / s; qz = 0.25 * s;
            }
            if (qw < 0.0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
            const double vn = sqrt(qx * qx + qy * qy + qz * qz);
            const double ang = 2.0 * atan2(vn, qw);
            const double sc = vn > 1e-15 ? ang / vn : 0.0;
            out[0] = qx * sc; out[1] = qy * sc; out[2] = qz * sc;
            out[3] = t[0]; out[4] = t[1]; out[5] = t[2];
            ok = out[0] == out[0] && out[3] == out[3];
        }
    }
    if (!ok) for (int i = 0; i < 6; ++i) out[i] = 0.0;
    a.valid_obs[o] = ok ? cnt : 0;
}

hipError_t launch_pose_init(const ccal_problem* p, int cam, const double* d_intr, double* d_poses_obs, int32_t* d_valid, int min_points, hipStream_t s) {
    InitArgs a = {};
    a.x = p->d_x; a.y = p->d_y; a.z = p->d_z; a.u = p->d_u; a.v = p->d_v;
    a.obs_off = p->d_obs_off; a.list = p->cams[cam].d_obs; a.n_list = (int32_t)p->cams[cam].obs.size(); a.cam = cam;
    a.intr = d_intr; a.poses_obs = d_poses_obs; a.valid_obs = d_valid; a.min_points = min_points; a.rt = model_rt(p->ctx);
    const int blocks = (a.n_list + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (blocks == 0) return hipSuccess;
    switch (p->cams[cam].model) {
        case kUCM: hipLaunchKernelGGL(k_pose_init<kUCM>, dim3(blocks), dim3(256), 0, s, a); break;
        case kEUCM: hipLaunchKernelGGL(k_pose_init<kEUCM>, dim3(blocks), dim3(256), 0, s, a); break;
        case kKB4: hipLaunchKernelGGL(k_pose_init<kKB4>, dim3(blocks), dim3(256), 0, s, a); break;
        case kOCV5: hipLaunchKernelGGL(k_pose_init<kOCV5>, dim3(blocks), dim3(256), 0, s, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ccal
