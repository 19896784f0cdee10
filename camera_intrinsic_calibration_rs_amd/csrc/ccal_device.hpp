// Device-side math of the reprojection hot path (gfx950 / CDNA4, f64).
//
// What the reference does per corner with forward-mode dual numbers
// (src/optimization/factors.rs:152-173 and 204-228: rebuild the model, exp-map the rvec,
// transform, project, subtract) is re-derived here analytically and split by how often each
// piece changes:
//   per frame  : R(rvec), dR/drvec_k, composed transforms      -> frame_setup_*  (once per wave)
//   per corner : p = R X + t, projection + its 2x3 / 2xP partials, chain rule    (once per lane)
// Column order of a block Jacobian follows the reference's variable order
// [params, rvec, tvec(, rvec_i_0, tvec_i_0)] (src/util.rs:411, 621-627).
#pragma once
#include <hip/hip_runtime.h>

#include "ccal_models.hpp"

namespace ccal {

// Frame constants, laid out in LDS (doubles).  cam0 factor uses FC_RC/FC_TC/FC_A only (FC_N0 entries).
//   p      = RC X + TC
//   dp/drvec_0_b[k] = a_k x (RC X)         d(R X) = dphi x (R X) with dphi = J_l(w) dw: a_k is the k-th column of the
//                                          left Jacobian of SO(3) (cam 0), rotated by R1 for cam c > 0 - three vectors
//                                          and a cross product instead of three 3 x 3 matrices dR/dw_k
//   dp/dtvec_0_b    = I (cam 0) or R1
//   dp/drvec_c_0[k] = b_k x (p - t1)       b_k = k-th column of J_l(rvec_c_0), t1 = tvec_c_0
//   dp/dtvec_c_0    = I
constexpr int FC_RC = 0, FC_TC = 9, FC_A = 12, FC_N0 = 21, FC_N0P = 22;
constexpr int FC_R1 = 22, FC_B = 31, FC_T1 = 40, FC_SIZE = 44;

// 1/x and (sqrt x, 1/sqrt x) from the hardware seeds (v_rcp_f64 / v_rsq_f64) plus two fused
// Newton / Goldschmidt steps: <= 1-2 ulp, a third of the instructions of the IEEE division / sqrt
// expansions (no div_scale / div_fmas / div_fixup).  Arguments on this path are finite, positive and
// far from the denormal range (depths, radii, sums of squares of pixels).
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double& s, double& rs) {
    const double y0 = __builtin_amdgcn_rsq(x);
    double g = x * y0, h = 0.5 * y0;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    // one final correction of the square root: g += h * 2 * (x - g^2) / 2
    const double d = __builtin_fma(-g, g, x);
    s = __builtin_fma(d, h, g);
    rs = h + h;
}

// sin and cos for rotation angles (|x| < ~1e5): two-term Cody-Waite reduction by pi/2 and the fdlibm
// kernel polynomials (< 1 ulp on [-pi/4, pi/4]).  A fraction of the instructions of the general-purpose
// library sincos (no Payne-Hanek path, no branches).
__device__ __forceinline__ void fast_sincos(double x, double& s, double& c) {
    const double kd = __builtin_rint(x * 6.36619772367581382433e-01);
    const int k = (int)kd;
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double sr = __builtin_fma(r * z, ps, r);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const double s1 = (k & 1) ? cr : sr, c1 = (k & 1) ? sr : cr;
    s = (k & 2) ? -s1 : s1;
    c = ((k + 1) & 2) ? -c1 : c1;
}

// atan2(r, z) for r > 0 (the Kannala-Brandt angle theta of a point off the optical axis), branch-free, ~2 ulp:
//   a = min(r, |z|), b = max(r, |z|);  x = a / b, or (a - b) / (a + b) when a > tan(pi/8) b  =>  |x| <= tan(pi/8) < 7/16,
//   atan(x) by the fdlibm kernel polynomial (11 coefficients in x^2, < 1 ulp on |x| < 7/16), ONE division (hardware seed +
//   Newton + one residual correction), and the octant folded back as  theta = k pi/4 -+ atan(x)  with pi/4 in two parts.
// A third of the instructions of the general-purpose library atan2 (no special cases: r > 0 excludes them, no branches).
__device__ __forceinline__ double fast_atan2_pos(double r, double z) {
#ifdef CCAL_LIB_ATAN2          // A/B builds (tools/ab_eval.py): the general-purpose library function
    return atan2(r, z);
#endif
    const double az = __builtin_fabs(z);
    const bool big = r > az, neg = z < 0.0;
    const double a = big ? az : r, b = big ? r : az;                   // 0 <= a <= b, b > 0
    const bool sel = a > 0.41421356237309503 * b;
    const double num = sel ? a - b : a, den = sel ? a + b : b;
    const double y = fast_rcp(den);
    double x = num * y;
    x = __builtin_fma(__builtin_fma(-x, den, num), y, x);
    const double q = x * x, w = q * q;
    const double s1 = q * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                      w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                      w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    const double t = __builtin_fma(-x, s1 + s2, x);                    // atan(x)
    // theta = k pi/4 - s t:  z >= 0: r <= z: (k, s) = (0 | 1, -1), r > z: (2 | 1, +1);  z < 0: mirrored about pi/2
    const int k = sel ? (neg ? 3 : 1) : (big ? 2 : (neg ? 4 : 0));
    const double st = (big != neg) ? t : -t;
    const double kf = (double)k;
    return __builtin_fma(kf, 7.85398163397448278999e-01, __builtin_fma(kf, 3.06161699786838301793e-17, -st));
}

// R = exp([w]x) and the left Jacobian of SO(3), J_l(w) = a I + b W + e w w^T with
//   a = sin t / t,  b = (1 - cos t) / t^2,  e = (t - sin t) / t^3 = (1 - a) / t^2        (series below t^2 < 0.04).
// d(R X)/dw_k = (k-th column of J_l) x (R X): what forward-mode duals through Rodrigues' formula evaluate to, in
// closed form (three 3-vectors instead of three 3 x 3 matrices dR/dw_k).
// The reference's quaternion path (nalgebra from_scaled_axis) returns the identity as a constant at exactly
// rvec == 0, i.e. a zero rvec-Jacobian there; we use the true limit J_l = I instead ("rvec = 0").
__device__ inline void so3_exp_ljac(const double w[3], double R[9], double JL[9]) {
    const double wx = w[0], wy = w[1], wz = w[2];
    const double xx = wx * wx, yy = wy * wy, zz = wz * wz;
    const double t2 = xx + yy + zz;
    double a, b, e;
    if (t2 < 0.04) {
        a = 1.0 + t2 * (-1.0 / 6 + t2 * (1.0 / 120 + t2 * (-1.0 / 5040 + t2 * (1.0 / 362880 + t2 * (-1.0 / 39916800 + t2 * (1.0 / 6227020800.0))))));
        b = 0.5 + t2 * (-1.0 / 24 + t2 * (1.0 / 720 + t2 * (-1.0 / 40320 + t2 * (1.0 / 3628800 + t2 * (-1.0 / 479001600 + t2 * (1.0 / 87178291200.0))))));
        e = 1.0 / 6 + t2 * (-1.0 / 120 + t2 * (1.0 / 5040 + t2 * (-1.0 / 362880 + t2 * (1.0 / 39916800 + t2 * (-1.0 / 6227020800.0 + t2 * (1.0 / 1307674368000.0))))));
    } else {
        double t, it;
        fast_sqrt_rsqrt(t2, t, it);
        double s, co, sh, ch;
        fast_sincos(t, s, co);
        fast_sincos(0.5 * t, sh, ch);
        (void)ch; (void)co;
        const double it2 = it * it;
        a = s * it;
        b = 2.0 * sh * sh * it2;          // (1 - cos t) / t^2 without cancellation
        e = (1.0 - a) * it2;              // t^2 >= 0.04: 1 - a >= 6.6e-3, no harmful cancellation
    }
    const double xy = wx * wy, xz = wx * wz, yz = wy * wz;
    // R = I + a W + b W^2,  W^2 = w w^T - t2 I
    R[0] = 1.0 - b * (yy + zz); R[1] = b * xy - a * wz;     R[2] = b * xz + a * wy;
    R[3] = b * xy + a * wz;     R[4] = 1.0 - b * (xx + zz); R[5] = b * yz - a * wx;
    R[6] = b * xz - a * wy;     R[7] = b * yz + a * wx;     R[8] = 1.0 - b * (xx + yy);
    // J_l = a I + b W + e w w^T  (row-major)
    JL[0] = a + e * xx;      JL[1] = e * xy - b * wz; JL[2] = e * xz + b * wy;
    JL[3] = e * xy + b * wz; JL[4] = a + e * yy;      JL[5] = e * yz - b * wx;
    JL[6] = e * xz - b * wy; JL[7] = e * yz + b * wx; JL[8] = a + e * zz;
}

__device__ inline void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
__device__ inline void mat3_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[i * 3] * v[0] + A[i * 3 + 1] * v[1] + A[i * 3 + 2] * v[2];
}

// Fill the frame constants for one observation frame.  pose = rvec,tvec of T_0_b; extr = rvec,tvec
// of T_c_0 (OTHER only).  Every lane computes the same values; `fc` may be registers or LDS.
template <bool OTHER>
__device__ inline void frame_setup(const double* pose, const double* extr, double* fc) {
    double R0[9], J0[9];
    so3_exp_ljac(pose, R0, J0);
    if constexpr (!OTHER) {
#pragma unroll
        for (int i = 0; i < 9; ++i) fc[FC_RC + i] = R0[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) fc[FC_TC + i] = pose[3 + i];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < 3; ++i) fc[FC_A + 3 * k + i] = J0[i * 3 + k];       // a_k = k-th column of J_l
    } else {
        double R1[9], J1[9], tmp[9], v[3];
        so3_exp_ljac(extr, R1, J1);
        mat3_mul(R1, R0, tmp);
#pragma unroll
        for (int i = 0; i < 9; ++i) fc[FC_RC + i] = tmp[i];
        mat3_vec(R1, pose + 3, v);
#pragma unroll
        for (int i = 0; i < 3; ++i) fc[FC_TC + i] = v[i] + extr[3 + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double a0[3] = { J0[k], J0[3 + k], J0[6 + k] };
            mat3_vec(R1, a0, fc + FC_A + 3 * k);             // R1 (a x b) = (R1 a) x (R1 b)
#pragma unroll
            for (int i = 0; i < 3; ++i) fc[FC_B + 3 * k + i] = J1[i * 3 + k];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) fc[FC_R1 + i] = R1[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) fc[FC_T1 + i] = extr[3 + i];
        fc[FC_N0] = 0.0; fc[FC_SIZE - 1] = 0.0;              // padding entries
    }
}

// General (multi-camera) loop, register Gram kernels: every camera's blocks are evaluated at the COMPOSED pose
//   T_cb = T_c0 o T_0b  (camera 0: T_0b),   R = R_c0 R_0b,  t = R_c0 t_0b + t_c0,
// with corner_block<..., PHI>: six pose columns j [-(R X)x | I] in the split left-perturbation basis (phi_cb, delta_cb) of the
// composed pose - the same 6 + P_eff + 1 columns and the same kernel for every camera.  The block's twelve reference
// columns (rvec_0_b, tvec_0_b, rvec_c_0, tvec_c_0: src/optimization/factors.rs:212-218) are per-frame constant linear
// combinations of those six:
//   (phi_cb, delta_cb) = [ R_c0 J_l(rvec_0b)   0    |  J_l(rvec_c0)                      0 ] (d rvec_0b, d tvec_0b, d rvec_c0, d tvec_c0)
//                        [ 0                   R_c0 |  -[R_c0 t_0b]x J_l(rvec_c0)        I ]
// so the per-slot elimination (k_schur) expands a frame's 13-column Gram with that 6 x 12 matrix E once per frame instead
// of every corner carrying 19 columns.  fc12 = R | t; ept = E transposed, dense: ept[6 b + m] = E[m][b], b < 12
// (camera 0: extr = 0, i.e. R_c0 = I; its extrinsics columns b >= 6 are never read).
constexpr int GEN_EPT = 72;
__device__ inline void frame_setup_composed(const double* pose, const double* extr, double* fc12, double* ept) {
    // camera 0 passes extr = 0: R_c0 = I and J_l = I come out exactly, the products below reproduce R_0b, t_0b, J_l(rvec_0b)
    // bit for bit - one code path, no merge of two 72-value results
    double R0[9], J0[9], R1[9], J1[9], v[3], RJ[9], SJ[9];
    so3_exp_ljac(pose, R0, J0);
    so3_exp_ljac(extr, R1, J1);
    mat3_mul(R1, R0, fc12);
    mat3_vec(R1, pose + 3, v);
#pragma unroll
    for (int i = 0; i < 3; ++i) fc12[9 + i] = v[i] + extr[3 + i];
    mat3_mul(R1, J0, RJ);
    const double S[9] = { 0.0, v[2], -v[1], -v[2], 0.0, v[0], v[1], -v[0], 0.0 };      // -[v]x
    mat3_mul(S, J1, SJ);
#pragma unroll
    for (int i = 0; i < GEN_EPT; ++i) ept[i] = 0.0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            ept[6 * b + m] = RJ[m * 3 + b];                 // rvec_0_b -> phi
            ept[6 * (3 + b) + 3 + m] = R1[m * 3 + b];       // tvec_0_b -> delta
            ept[6 * (6 + b) + m] = J1[m * 3 + b];           // rvec_c_0 -> phi
            ept[6 * (6 + b) + 3 + m] = SJ[m * 3 + b];       // rvec_c_0 -> delta
        }
        ept[6 * (9 + b) + 3 + b] = 1.0;                     // tvec_c_0 -> delta
    }
}

// Intrinsics of one camera as the kernels hold them: th[0 .. P) = FULL model parameters (fy := fx with ONE_FOCAL)
// and th[P] = the run-time convention slot (KB4: the small-radius threshold of project_one, dead for the others).
template <int MODEL> __host__ __device__ constexpr int th_len() { return model_np(MODEL) + 1; }
// OPENCV5: th[4 ..] = k1, k2, p1, p2, k3 (the kernels' canonical order) gathered from wherever the caller's params() vector
// keeps them (rt.ocv5_perm: ccal_model_conventions.ocv5_order; wave-uniform, the identity by default).
template <int MODEL, bool ONE_FOCAL>
__device__ __forceinline__ void load_theta(const double* th_g, const ModelRt& rt, double* th) {
    if constexpr (MODEL == kOCV5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) th[i] = th_g[i];
#pragma unroll
        for (int i = 0; i < 5; ++i) th[4 + i] = th_g[4 + ocv5_pos(rt.ocv5_perm, i)];
    } else {
#pragma unroll
        for (int i = 0; i < model_np(MODEL); ++i) th[i] = th_g[i];
    }
    if constexpr (ONE_FOCAL) th[1] = th[0];
    th[model_np(MODEL)] = rt.kb4_eps;
}

// Normalised projection m = (mx, my) with partials w.r.t. the camera-frame point (dmx[3], dmy[3])
// and w.r.t. the distortion parameters th[4..P) (ddx[ND], ddy[ND]).
template <int MODEL>
__device__ __forceinline__ void project_partials(const double* th, double x, double y, double z,
                                                 double& mx, double& my, double* dmx, double* dmy,
                                                 double* ddx, double* ddy) {
    if constexpr (MODEL == kUCM || MODEL == kEUCM) {
        const double alpha = th[4];
        const double beta = (MODEL == kEUCM) ? th[5] : 1.0;
        const double r2 = x * x + y * y;
        double rho, irho;
        fast_sqrt_rsqrt(beta * r2 + z * z, rho, irho);
        const double n = alpha * rho + (1.0 - alpha) * z;
        const double inv = fast_rcp(n);
        mx = x * inv; my = y * inv;
        const double ab = alpha * beta * irho;
        const double nx = ab * x, ny = ab * y, nz = alpha * z * irho + (1.0 - alpha);
        const double mxi = mx * inv, myi = my * inv;
        dmx[0] = inv - mxi * nx; dmx[1] = -mxi * ny;      dmx[2] = -mxi * nz;
        dmy[0] = -myi * nx;      dmy[1] = inv - myi * ny; dmy[2] = -myi * nz;
        const double na = rho - z;
        ddx[0] = -mxi * na; ddy[0] = -myi * na;
        if constexpr (MODEL == kEUCM) {
            const double nb = 0.5 * alpha * r2 * irho;
            ddx[1] = -mxi * nb; ddy[1] = -myi * nb;
        }
    } else if constexpr (MODEL == kKB4) {
        const double r2 = x * x + y * y;
        double r, ir;
        fast_sqrt_rsqrt(r2, r, ir);          // r2 == 0 gives NaN here and falls into the pinhole branch below
        if (r > th[model_np(kKB4)]) {          // run-time convention slot (load_theta): ccal_model_conventions.kb4_small_radius
            const double t = fast_atan2_pos(r, z);
            const double t2 = t * t;
            const double k1 = th[4], k2 = th[5], k3 = th[6], k4 = th[7];
            const double td = t * (1.0 + t2 * (k1 + t2 * (k2 + t2 * (k3 + t2 * k4))));
            const double tdp = 1.0 + t2 * (3.0 * k1 + t2 * (5.0 * k2 + t2 * (7.0 * k3 + t2 * 9.0 * k4)));
            const double s = td * ir;
            const double id2 = fast_rcp(r2 + z * z);
            const double tq = z * ir * id2;                 // theta_x = tq x, theta_y = tq y
            const double tz = -r * id2;
            const double g = ir * (tdp * tq - s * ir);      // s_x = g x, s_y = g y
            const double sz = ir * tdp * tz;
            mx = x * s; my = y * s;
            dmx[0] = s + x * x * g; dmx[1] = x * y * g;     dmx[2] = x * sz;
            dmy[0] = x * y * g;     dmy[1] = s + y * y * g; dmy[2] = y * sz;
            const double xr = x * ir, yr = y * ir;
            const double t3 = t2 * t, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
            ddx[0] = xr * t3; ddx[1] = xr * t5; ddx[2] = xr * t7; ddx[3] = xr * t9;
            ddy[0] = yr * t3; ddy[1] = yr * t5; ddy[2] = yr * t7; ddy[3] = yr * t9;
        } else {
            const double iz = fast_rcp(z);
            mx = x * iz; my = y * iz;
            dmx[0] = iz; dmx[1] = 0.0; dmx[2] = -mx * iz;
            dmy[0] = 0.0; dmy[1] = iz; dmy[2] = -my * iz;
#pragma unroll
            for (int i = 0; i < 4; ++i) { ddx[i] = 0.0; ddy[i] = 0.0; }
        }
    } else {   // OPENCV5, parameter order from ccal_models.hpp
        const double k1 = th[OCV5_K1], k2 = th[OCV5_K2], p1 = th[OCV5_P1], p2 = th[OCV5_P2], k3 = th[OCV5_K3];
        const double iz = fast_rcp(z);
        const double xn = x * iz, yn = y * iz;
        const double xx = xn * xn, yy = yn * yn, xy = xn * yn;
        const double r2 = xx + yy;
        const double rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
        const double drad = k1 + r2 * (2.0 * k2 + r2 * 3.0 * k3);
        mx = xn * rad + 2.0 * p1 * xy + p2 * (r2 + 2.0 * xx);
        my = yn * rad + p1 * (r2 + 2.0 * yy) + 2.0 * p2 * xy;
        const double xd_x = rad + 2.0 * xx * drad + 2.0 * p1 * yn + 6.0 * p2 * xn;
        const double xd_y = 2.0 * xy * drad + 2.0 * p1 * xn + 2.0 * p2 * yn;
        const double yd_x = xd_y;
        const double yd_y = rad + 2.0 * yy * drad + 6.0 * p1 * yn + 2.0 * p2 * xn;
        dmx[0] = xd_x * iz; dmx[1] = xd_y * iz; dmx[2] = -(xd_x * xn + xd_y * yn) * iz;
        dmy[0] = yd_x * iz; dmy[1] = yd_y * iz; dmy[2] = -(yd_x * xn + yd_y * yn) * iz;
        const double r4 = r2 * r2, r6 = r4 * r2;
        ddx[OCV5_K1 - 4] = xn * r2; ddx[OCV5_K2 - 4] = xn * r4; ddx[OCV5_P1 - 4] = 2.0 * xy;        ddx[OCV5_P2 - 4] = r2 + 2.0 * xx; ddx[OCV5_K3 - 4] = xn * r6;
        ddy[OCV5_K1 - 4] = yn * r2; ddy[OCV5_K2 - 4] = yn * r4; ddy[OCV5_P1 - 4] = r2 + 2.0 * yy;   ddy[OCV5_P2 - 4] = 2.0 * xy;      ddy[OCV5_K3 - 4] = yn * r6;
    }
}

// Residual only (validation / cost): u, v of one corner.
template <int MODEL>
__device__ __forceinline__ void project_uv(const double* th, double x, double y, double z, double& u, double& v) {
    double mx, my;
    if constexpr (MODEL == kUCM || MODEL == kEUCM) {
        const double beta = (MODEL == kEUCM) ? th[5] : 1.0;
        double rho, irho;
        fast_sqrt_rsqrt(beta * (x * x + y * y) + z * z, rho, irho);
        (void)irho;
        const double inv = fast_rcp(th[4] * rho + (1.0 - th[4]) * z);
        mx = x * inv; my = y * inv;
    } else if constexpr (MODEL == kKB4) {
        const double r = sqrt(x * x + y * y);
        if (r > th[model_np(kKB4)]) {
            const double t = fast_atan2_pos(r, z), t2 = t * t;
            const double s = t * (1.0 + t2 * (th[4] + t2 * (th[5] + t2 * (th[6] + t2 * th[7])))) / r;
            mx = x * s; my = y * s;
        } else { mx = x / z; my = y / z; }
    } else {
        const double iz = 1.0 / z, xn = x * iz, yn = y * iz, xx = xn * xn, yy = yn * yn, xy = xn * yn, r2 = xx + yy;
        const double rad = 1.0 + r2 * (th[OCV5_K1] + r2 * (th[OCV5_K2] + r2 * th[OCV5_K3]));
        mx = xn * rad + 2.0 * th[OCV5_P1] * xy + th[OCV5_P2] * (r2 + 2.0 * xx);
        my = yn * rad + th[OCV5_P1] * (r2 + 2.0 * yy) + 2.0 * th[OCV5_P2] * xy;
    }
    u = th[0] * mx + th[2]; v = th[1] * my + th[3];
}

// One residual block: r[2] and the two Jacobian rows Ju[D], Jv[D].
//   th : FULL model parameters (fy == fx when ONE_FOCAL), fc : frame constants.
// PHI (camera-0 blocks of the single-camera Gram kernels): the three rotation columns are taken with respect to a LEFT
// perturbation phi of the rotation, d(R X) = phi x (R X), i.e. row (R X) x ju - no frame constants a_k, a third of the
// arithmetic.  d/d rvec = (d/d phi) J_l(rvec) with the frame's 3 x 3 left Jacobian, which the per-frame elimination
// applies ONCE to the frame's reduced Gram blocks (eliminate_frame, ccal_gram_common.hpp) instead of every corner applying it to its rows.
template <int MODEL, bool ONE_FOCAL, bool OTHER, bool PHI = false>
__device__ __forceinline__ void corner_block(const double* th, const double* fc,
                                             double X, double Y, double Z, double uo, double vo,
                                             double& ru, double& rv, double* Ju, double* Jv) {
    constexpr int P = model_np(MODEL);
    constexpr int ND = P - 4;
    constexpr int PE = P - (ONE_FOCAL ? 1 : 0);
    // rotated board point first: the rotation columns need it without the translation
    const double rx = fc[FC_RC + 0] * X + fc[FC_RC + 1] * Y + fc[FC_RC + 2] * Z;
    const double ry = fc[FC_RC + 3] * X + fc[FC_RC + 4] * Y + fc[FC_RC + 5] * Z;
    const double rz = fc[FC_RC + 6] * X + fc[FC_RC + 7] * Y + fc[FC_RC + 8] * Z;
    const double px = rx + fc[FC_TC + 0], py = ry + fc[FC_TC + 1], pz = rz + fc[FC_TC + 2];
    double mx, my, dmx[3], dmy[3], ddx[ND], ddy[ND];
    project_partials<MODEL>(th, px, py, pz, mx, my, dmx, dmy, ddx, ddy);
    const double fx = th[0], fy = th[1];
    ru = fx * mx + th[2] - uo;
    rv = fy * my + th[3] - vo;
    // intrinsics columns
    if constexpr (ONE_FOCAL) {       // [f, cx, cy, dist..]  (factors.rs:155-158: f feeds both fx and fy)
        Ju[0] = mx; Ju[1] = 1.0; Ju[2] = 0.0;
        Jv[0] = my; Jv[1] = 0.0; Jv[2] = 1.0;
    } else {                         // [fx, fy, cx, cy, dist..]
        Ju[0] = mx;  Ju[1] = 0.0; Ju[2] = 1.0; Ju[3] = 0.0;
        Jv[0] = 0.0; Jv[1] = my;  Jv[2] = 0.0; Jv[3] = 1.0;
    }
    constexpr int D0 = ONE_FOCAL ? 3 : 4;
#pragma unroll
    for (int i = 0; i < ND; ++i) { Ju[D0 + i] = fx * ddx[i]; Jv[D0 + i] = fy * ddy[i]; }
    // d(u,v)/dp
    const double ju[3] = { fx * dmx[0], fx * dmx[1], fx * dmx[2] };
    const double jv[3] = { fy * dmy[0], fy * dmy[1], fy * dmy[2] };
    if constexpr (PHI) {
        static_assert(!OTHER || !PHI, "the phi basis takes one (possibly composed) pose: OTHER blocks go through frame_setup_composed");
        // d(u, v) / d phi = (R X) x j
        Ju[PE + 0] = ry * ju[2] - rz * ju[1]; Ju[PE + 1] = rz * ju[0] - rx * ju[2]; Ju[PE + 2] = rx * ju[1] - ry * ju[0];
        Jv[PE + 0] = ry * jv[2] - rz * jv[1]; Jv[PE + 1] = rz * jv[0] - rx * jv[2]; Jv[PE + 2] = rx * jv[1] - ry * jv[0];
    } else {
        // rvec_0_b columns: a_k x (RC X)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double* A = fc + FC_A + 3 * k;
            const double qx = A[1] * rz - A[2] * ry;
            const double qy = A[2] * rx - A[0] * rz;
            const double qz = A[0] * ry - A[1] * rx;
            Ju[PE + k] = ju[0] * qx + ju[1] * qy + ju[2] * qz;
            Jv[PE + k] = jv[0] * qx + jv[1] * qy + jv[2] * qz;
        }
    }
    if constexpr (!OTHER) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { Ju[PE + 3 + k] = ju[k]; Jv[PE + 3 + k] = jv[k]; }
    } else {
        const double* R1 = fc + FC_R1;
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // tvec_0_b: d p / d t0 = R1
            Ju[PE + 3 + k] = ju[0] * R1[k] + ju[1] * R1[3 + k] + ju[2] * R1[6 + k];
            Jv[PE + 3 + k] = jv[0] * R1[k] + jv[1] * R1[3 + k] + jv[2] * R1[6 + k];
        }
        const double wx = px - fc[FC_T1 + 0], wy = py - fc[FC_T1 + 1], wz = pz - fc[FC_T1 + 2];     // R1 (R0 X + t0)
#pragma unroll
        for (int k = 0; k < 3; ++k) {   // rvec_c_0: b_k x (p - t1)
            const double* B = fc + FC_B + 3 * k;
            const double qx = B[1] * wz - B[2] * wy;
            const double qy = B[2] * wx - B[0] * wz;
            const double qz = B[0] * wy - B[1] * wx;
            Ju[PE + 6 + k] = ju[0] * qx + ju[1] * qy + ju[2] * qz;
            Jv[PE + 6 + k] = jv[0] * qx + jv[1] * qy + jv[2] * qz;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) { Ju[PE + 9 + k] = ju[k]; Jv[PE + 9 + k] = jv[k]; }
    }
}

// From the phi basis of a frame's reduced Gram blocks to the rvec basis (see corner_block<..., PHI>): T = diag(J_l, I).
//   C  packed lower 6 x 6 pose block (index i (i + 1) / 2 + j, pose order rvec | tvec), in place
//   jl the frame's left Jacobian as its three columns a_k: jl[3 k + m] = J_l[m][k]
__device__ __forceinline__ void phi_to_rvec_C(double* C, const double* jl) {
    // rows t = 3..5, columns phi: C[t][i] <- sum_m C[t][m] J_l[m][i]
#pragma unroll
    for (int t = 3; t < 6; ++t) {
        const double c0 = C[t * (t + 1) / 2 + 0], c1 = C[t * (t + 1) / 2 + 1], c2 = C[t * (t + 1) / 2 + 2];
#pragma unroll
        for (int i = 0; i < 3; ++i) C[t * (t + 1) / 2 + i] = c0 * jl[3 * i + 0] + c1 * jl[3 * i + 1] + c2 * jl[3 * i + 2];
    }
    // phi x phi block: J_l^T M J_l with the symmetric M
    const double M[9] = { C[0], C[1], C[3], C[1], C[2], C[4], C[3], C[4], C[5] };
    double N[9];                                   // N = M J_l
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i) N[r * 3 + i] = M[r * 3 + 0] * jl[3 * i + 0] + M[r * 3 + 1] * jl[3 * i + 1] + M[r * 3 + 2] * jl[3 * i + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) C[i * (i + 1) / 2 + j] = jl[3 * i + 0] * N[0 * 3 + j] + jl[3 * i + 1] * N[1 * 3 + j] + jl[3 * i + 2] * N[2 * 3 + j];
}
// one column of [B | g] (six pose rows): rows 0..2 <- J_l^T (rows 0..2)
__device__ __forceinline__ void phi_to_rvec_col(double* b, const double* jl) {
    const double b0 = b[0], b1 = b[1], b2 = b[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = jl[3 * i + 0] * b0 + jl[3 * i + 1] * b1 + jl[3 * i + 2] * b2;
}

// Point in the camera frame only (cost / validation kernels).
__device__ __forceinline__ void transform_point(const double* fc, double X, double Y, double Z,
                                                double& px, double& py, double& pz) {
    px = fc[FC_RC + 0] * X + fc[FC_RC + 1] * Y + fc[FC_RC + 2] * Z + fc[FC_TC + 0];
    py = fc[FC_RC + 3] * X + fc[FC_RC + 4] * Y + fc[FC_RC + 5] * Z + fc[FC_TC + 1];
    pz = fc[FC_RC + 6] * X + fc[FC_RC + 7] * Y + fc[FC_RC + 8] * Z + fc[FC_TC + 2];
}

// Huber weight as tiny-solver evaluates it: rho'(s) = 1 (s <= delta^2) else delta / sqrt(s).
__device__ __forceinline__ double huber_weight(double s, double delta) {
    return (delta > 0.0 && s > delta * delta) ? delta / sqrt(s) : 1.0;
}
// sqrt(rho') -- the factor tiny-solver's corrector applies to r and J.  1 for inliers; the outlier
// branch is skipped by the whole wavefront when no lane needs it.
// The outlier branch: (delta^2 / s)^(1/4) from the hardware's reciprocal-square-root seed and Newton steps (fast_sqrt_rsqrt, <= 2 ulp;
// s > delta^2 > 0: no scaling for tiny or huge arguments needed) - the IEEE sqrt, division and sqrt the expression compiles to are ~60
// instructions that EVERY lane of the wavefront executes whenever one of its corners is an outlier: with 5 % outliers four passes of five
// (k_gram2<EUCM>: 303 instructions per pass otherwise).  -DCCAL_HUBER_IEEE: the library expansions (A/B builds).
__device__ __forceinline__ double huber_sqrt_weight(double s, double delta) {
    double sw = 1.0;
    if (delta > 0.0 && s > delta * delta) {
#ifdef CCAL_HUBER_IEEE
        sw = sqrt(delta / sqrt(s));
#else
        double sq, rs, unused;
        fast_sqrt_rsqrt(s, sq, rs);                 // rs = 1 / sqrt(s)
        fast_sqrt_rsqrt(delta * rs, sw, unused);
#endif
    }
    return sw;
}

}  // namespace ccal
