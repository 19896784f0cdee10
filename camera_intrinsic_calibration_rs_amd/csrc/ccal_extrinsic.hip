// init_camera_extrinsic (src/util.rs:511-561): the relative pose T_i_0 of camera i w.r.t. camera 0 from
// the board poses both cameras estimated for the same frames.  One SE3Factor per common frame
// (src/optimization/factors.rs:234-272): residual = [scaled_axis(R_diff), t_diff] of
// T_diff = T_i_b^-1 * T_i_0 * T_0_b, HuberLoss(0.5), Gauss-Newton with tiny-solver's defaults, started
// from T_i_b * T_0_b^-1 of the first common frame.  SURVEY 8(f) rank 4: a 6-unknown problem with a few
// hundred residuals -- host code of the library, not a GPU kernel.  The 6 x 6 block Jacobian is analytic (what the
// reference's dual numbers evaluate to): with w = rvec, R = exp([w]x), J_l the left Jacobian of SO(3),
//   R_d = R_ib^T R R_0b                       a step d in rvec is the left perturbation psi = R_ib^T J_l(w) d of R_d
//   r_rot = log(R_d)                          d r_rot / d rvec = J_l^-1(r_rot) R_ib^T J_l(w),   d r_rot / d tvec = 0
//   r_t   = R_ib^T (R t_0b + tvec - t_ib)     d r_t / d rvec = R_ib^T [ a_k x (R t_0b) ]_k,      d r_t / d tvec = R_ib^T
#include <cmath>
#include <cstring>
#include <vector>

#include "ccal_internal.hpp"

namespace {

struct Q { double w, x, y, z; };
struct Iso { Q q; double t[3]; };

Q quat_from_rvec(const double* r) {
    const double hx = 0.5 * r[0], hy = 0.5 * r[1], hz = 0.5 * r[2];
    const double nn = hx * hx + hy * hy + hz * hz;
    if (nn <= 4.930380657631324e-32) return { 1, 0, 0, 0 };
    const double n = std::sqrt(nn), s = std::sin(n) / n;
    return { std::cos(n), hx * s, hy * s, hz * s };
}
void rot(const Q& q, const double* p, double* o) {
    const double tx = 2 * (q.y * p[2] - q.z * p[1]), ty = 2 * (q.z * p[0] - q.x * p[2]), tz = 2 * (q.x * p[1] - q.y * p[0]);
    o[0] = p[0] + q.w * tx + (q.y * tz - q.z * ty);
    o[1] = p[1] + q.w * ty + (q.z * tx - q.x * tz);
    o[2] = p[2] + q.w * tz + (q.x * ty - q.y * tx);
}
Q qmul(const Q& a, const Q& b) {
    return { a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w };
}
Iso iso_from6(const double* p) { Iso m; m.q = quat_from_rvec(p); m.t[0] = p[3]; m.t[1] = p[4]; m.t[2] = p[5]; return m; }
Iso iso_mul(const Iso& a, const Iso& b) { Iso m; m.q = qmul(a.q, b.q); rot(a.q, b.t, m.t); for (int i = 0; i < 3; ++i) m.t[i] += a.t[i]; return m; }
Iso iso_inv(const Iso& a) { Iso m; m.q = { a.q.w, -a.q.x, -a.q.y, -a.q.z }; double r[3]; rot(m.q, a.t, r); for (int i = 0; i < 3; ++i) m.t[i] = -r[i]; return m; }
void scaled_axis(const Q& q, double* o) {
    const double sg = q.w >= 0 ? 1.0 : -1.0;
    const double vx = q.x * sg, vy = q.y * sg, vz = q.z * sg, n = std::sqrt(vx * vx + vy * vy + vz * vz);
    if (n <= 2.220446049250313e-16) { o[0] = o[1] = o[2] = 0; return; }
    const double a = 2.0 * std::atan2(n, std::fabs(q.w)) / n;
    o[0] = vx * a; o[1] = vy * a; o[2] = vz * a;
}
void iso_to6(const Iso& m, double* o) { scaled_axis(m.q, o); o[3] = m.t[0]; o[4] = m.t[1]; o[5] = m.t[2]; }

// SE3Factor::residual_func
void se3_residual(const Iso& t_0_b, const Iso& t_i_b_inv, const double* x, double* r) {
    const Iso d = iso_mul(iso_mul(t_i_b_inv, iso_from6(x)), t_0_b);
    iso_to6(d, r);
}

void quat_to_mat(const Q& q, double* R) {
    const double e0[3] = { 1, 0, 0 }, e1[3] = { 0, 1, 0 }, e2[3] = { 0, 0, 1 };
    double c0[3], c1[3], c2[3];
    rot(q, e0, c0); rot(q, e1, c1); rot(q, e2, c2);
    for (int i = 0; i < 3; ++i) { R[i * 3 + 0] = c0[i]; R[i * 3 + 1] = c1[i]; R[i * 3 + 2] = c2[i]; }
}
// left Jacobian of SO(3) and its inverse at w (row-major 3 x 3); series near 0
void so3_left_jacobian(const double* w, double* J, bool inverse) {
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double b, e;                 // J = I + b W + e W^2  (inverse: b = -1/2)
    if (!inverse) {
        if (t2 < 1e-8) { b = 0.5 - t2 / 24.0; e = 1.0 / 6.0 - t2 / 120.0; }
        else { const double t = std::sqrt(t2); b = (1.0 - std::cos(t)) / t2; e = (t - std::sin(t)) / (t2 * t); }
    } else {
        b = -0.5;
        if (t2 < 1e-8) e = 1.0 / 12.0 + t2 / 720.0;
        else { const double t = std::sqrt(t2); e = 1.0 / t2 - (1.0 + std::cos(t)) / (2.0 * t * std::sin(t)); }
    }
    const double W[9] = { 0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0 };
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double w2 = 0; for (int k = 0; k < 3; ++k) w2 += W[i * 3 + k] * W[k * 3 + j];
        J[i * 3 + j] = (i == j ? 1.0 : 0.0) + b * W[i * 3 + j] + e * w2;
    }
}
// residual and its analytic 6 x 6 Jacobian (columns rvec, tvec)
void se3_residual_jac(const Iso& t_0_b, const Iso& t_i_b_inv, const double* x, double* r, double J[6][6]) {
    se3_residual(t_0_b, t_i_b_inv, x, r);
    double Rib_t[9], Jl[9], Jinv[9];
    quat_to_mat(t_i_b_inv.q, Rib_t);                       // rotation of T_i_b^-1 = R_ib^T
    so3_left_jacobian(x, Jl, false);
    so3_left_jacobian(r, Jinv, true);
    double M[9];                                           // R_ib^T J_l(w)
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double t = 0; for (int k = 0; k < 3; ++k) t += Rib_t[i * 3 + k] * Jl[k * 3 + j]; M[i * 3 + j] = t; }
    double Rt0[3];
    rot(quat_from_rvec(x), t_0_b.t, Rt0);                  // R t_0b
    // nalgebra's scaled_axis() returns the CONSTANT zero vector when the rotation has no axis (|imag q| <= eps): with
    // dual numbers the rotation rows of such a block have a zero Jacobian - which happens in the reference's first
    // iteration for the frame the start value was built from (src/util.rs:530).  Reproduced for parity.
    const Iso d = iso_mul(iso_mul(t_i_b_inv, iso_from6(x)), t_0_b);
    const bool no_axis = std::sqrt(d.q.x * d.q.x + d.q.y * d.q.y + d.q.z * d.q.z) <= 2.220446049250313e-16;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double t = 0; for (int k = 0; k < 3; ++k) t += Jinv[i * 3 + k] * M[k * 3 + j];
        J[i][j] = no_axis ? 0.0 : t; J[i][3 + j] = 0.0; J[3 + i][3 + j] = Rib_t[i * 3 + j];
    }
    for (int k = 0; k < 3; ++k) {                           // a_k x (R t_0b), a_k = k-th column of J_l
        const double a[3] = { Jl[0 * 3 + k], Jl[1 * 3 + k], Jl[2 * 3 + k] };
        const double c[3] = { a[1] * Rt0[2] - a[2] * Rt0[1], a[2] * Rt0[0] - a[0] * Rt0[2], a[0] * Rt0[1] - a[1] * Rt0[0] };
        for (int i = 0; i < 3; ++i) J[3 + i][k] = Rib_t[i * 3 + 0] * c[0] + Rib_t[i * 3 + 1] * c[1] + Rib_t[i * 3 + 2] * c[2];
    }
}

double cost(const std::vector<Iso>& a, const std::vector<Iso>& binv, const double* x, double delta) {
    double c = 0;
    for (size_t k = 0; k < a.size(); ++k) {
        double r[6]; se3_residual(a[k], binv[k], x, r);
        double s = 0; for (int i = 0; i < 6; ++i) s += r[i] * r[i];
        c += (s <= delta * delta ? 1.0 : delta / std::sqrt(s)) * s;
    }
    return c;
}

}  // namespace

// SE3Factor::residual_func (src/optimization/factors.rs:248-271) for one frame: residual r[6] and the 6 x 6 block Jacobian
// (row-major) the reference's dual numbers evaluate to
extern "C" int ccal_se3_factor(const double* pose_0_b, const double* pose_i_b, const double* x, double* r, double* J) {
    if (!pose_0_b || !pose_i_b || !x || !r || !J) return CCAL_ERR_INVALID_ARG;
    double Jm[6][6];
    se3_residual_jac(iso_from6(pose_0_b), iso_inv(iso_from6(pose_i_b)), x, r, Jm);
    std::memcpy(J, Jm, sizeof Jm);
    return CCAL_OK;
}

extern "C" int ccal_init_camera_extrinsic_opts(const double* poses_cam0, const double* poses_cami, int n_common, double* t_i_0_io,
                                               int use_initial, const ccal_solver_opts* opts, ccal_report* rep) {
    if (!poses_cam0 || !poses_cami || n_common < 1 || !t_i_0_io) return CCAL_ERR_INVALID_ARG;
    CCAL_API_TRY
    // GaussNewtonOptimizer::default() (src/util.rs:543) unless the caller says otherwise; only the stop rules are read
    const int max_it = opts ? opts->max_iterations : 100, em = opts ? opts->error_metric : 0;
    const double min_err = opts ? opts->min_error : 1e-10, min_abs = opts ? opts->min_abs_error_decrease : 1e-5,
                 min_rel = opts ? opts->min_rel_error_decrease : 1e-5;
    const double delta = 0.5;                                   // HuberLoss::new(0.5), src/util.rs:539
    std::vector<Iso> a(n_common), binv(n_common);
    for (int k = 0; k < n_common; ++k) { a[k] = iso_from6(poses_cam0 + 6 * k); binv[k] = iso_inv(iso_from6(poses_cami + 6 * k)); }
    double x[6];
    if (use_initial) std::memcpy(x, t_i_0_io, sizeof x);
    else iso_to6(iso_mul(iso_from6(poses_cami), iso_inv(a[0])), x);      // t_i_b[0] * t_0_b[0]^-1, src/util.rs:530
    ccal_report R = {};
    double cur = cost(a, binv, x, delta);
    R.initial_cost = cur;
    int status = CCAL_OK;
    for (int it = 0; it < max_it; ++it) {
        const double last = cur;
        double H[36] = { 0 }, g[6] = { 0 };
        for (int k = 0; k < n_common; ++k) {
            double r[6], J[6][6];
            se3_residual_jac(a[k], binv[k], x, r, J);
            double s = 0; for (int i = 0; i < 6; ++i) s += r[i] * r[i];
            const double w = s <= delta * delta ? 1.0 : delta / std::sqrt(s);
            for (int i = 0; i < 6; ++i) for (int c = 0; c < 6; ++c) {
                g[c] += w * J[i][c] * r[i];
                for (int e = 0; e < 6; ++e) H[c * 6 + e] += w * J[i][c] * J[i][e];
            }
        }
        // Cholesky 6x6
        double L[36]; std::memcpy(L, H, sizeof L);
        bool ok = true;
        for (int j = 0; j < 6 && ok; ++j) {
            double s = L[j * 6 + j]; for (int q = 0; q < j; ++q) s -= L[j * 6 + q] * L[j * 6 + q];
            if (!(s > 0)) { ok = false; break; }
            const double l = std::sqrt(s); L[j * 6 + j] = l;
            for (int i = j + 1; i < 6; ++i) { double t = L[i * 6 + j]; for (int q = 0; q < j; ++q) t -= L[i * 6 + q] * L[j * 6 + q]; L[i * 6 + j] = t / l; }
        }
        if (!ok) { status = CCAL_ERR_NOT_PD; break; }
        double dx[6];
        for (int i = 0; i < 6; ++i) { double t = -g[i]; for (int q = 0; q < i; ++q) t -= L[i * 6 + q] * dx[q]; dx[i] = t / L[i * 6 + i]; }
        for (int i = 5; i >= 0; --i) { double t = dx[i]; for (int q = i + 1; q < 6; ++q) t -= L[q * 6 + i] * dx[q]; dx[i] = t / L[i * 6 + i]; }
        for (int i = 0; i < 6; ++i) x[i] += dx[i];
        cur = cost(a, binv, x, delta);
        R.iterations++;
        const double le = em ? std::sqrt(std::max(last, 0.0)) : last, ce = em ? std::sqrt(std::max(cur, 0.0)) : cur;
        if (ce < min_err) break;
        if (std::isnan(cur)) { status = CCAL_ERR_NONFINITE; break; }
        if (std::fabs(le - ce) < min_abs) break;
        if (std::fabs(le - ce) / le < min_rel) break;
    }
    R.final_cost = cur; R.status = status;
    if (rep) *rep = R;
    if (status == CCAL_OK) std::memcpy(t_i_0_io, x, sizeof x);
    return status;
    CCAL_API_CATCH((ccal_ctx*)nullptr)
}
extern "C" int ccal_init_camera_extrinsic(const double* poses_cam0, const double* poses_cami, int n_common, double* t_i_0_io,
                                          int use_initial, ccal_report* rep) {
    return ccal_init_camera_extrinsic_opts(poses_cam0, poses_cami, n_common, t_i_0_io, use_initial, nullptr, rep);
}
