// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the reference's reprojection hot path, written from the
// reference's call sites (the arithmetic itself lives in crates.io
// dependencies that are absent from /root/reference, see below).  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link,
// load or execute anything under oracle/.
//
// What is restated, and from where (paths relative to /root/reference):
//   * ReprojectionFactor::residual_func          src/optimization/factors.rs:152-173
//   * OtherCamReprojectionFactor::residual_func  src/optimization/factors.rs:204-228
//   * RvecTvec / Isometry3::new(tvec, rvec)      src/types.rs:14-36, 55-64
//   * calib_camera problem layout + GN solve     src/util.rs:384-490
//   * calib_all_camera_with_extrinsics           src/util.rs:567-715
//   * bounds / disabled distortion               src/util.rs:29-71
//   * validation statistics                      src/util.rs:778-795
//
// Third-party arithmetic that is NOT in the reference tree (Cargo.toml:25,35,44;
// no Cargo.lock, caret versions): camera-intrinsic-model "0.8.0"
// (GenericModel::project_one), nalgebra "0.34.1" (Isometry3 / UnitQuaternion),
// tiny-solver "0.18.0" (forward-mode dual Jacobian, Huber corrector,
// Gauss-Newton loop).  Their published algorithms are restated here:
//   - UCM/EUCM: Khomutenko et al. 2016 / Usenko et al. 2018 ("double sphere"
//     paper's EUCM section), alpha/beta form;
//   - KB4: Kannala-Brandt 2006, 4 coefficients, theta = atan2(r, z);
//   - OPENCV5: Brown-Conrady radtan [k1,k2,p1,p2,k3];
//   - axis-angle -> unit quaternion via quaternion exp, rotate by
//     t = 2 q_v x p ; p' = p + w t + q_v x t   (nalgebra's formulation);
//   - Ceres-style loss corrector (tiny-solver restates Ceres): for Huber
//     rho'' < 0 so residual and Jacobian are both scaled by sqrt(rho').
//
// PARITY PINNING.  The reference cannot be built here (no Rust toolchain), so
// this oracle is pinned against (a) every known-answer test the reference holds
// on this path (tests/optimization_test.rs:36-80, tests/types_test.rs:5-20,
// tests/util_test.rs:77-110 + src/util.rs:230-235, tests/board_test.rs:3-40)
// and (b) an independent 50-digit mpmath evaluation + sympy Jacobians
// (oracle/gen_golden.py -> tests/golden/*.json).  Nothing in the reference pins
// EUCM/KB4/OPENCV5 projection values, any Jacobian value, Huber behaviour or
// converged intrinsics: for those items parity is UNPINNED against the Rust
// crates' rounding and is pinned only against the published math.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------
// Forward-mode dual number with N tangent slots on the stack (tiny-solver uses
// num-dual's heap-backed DualDVec64; same arithmetic, faster container).
// ---------------------------------------------------------------------------
template <int N>
struct Dual {
    double re;
    double eps[N];
    Dual() : re(0.0) { for (int i = 0; i < N; ++i) eps[i] = 0.0; }
    Dual(double v) : re(v) { for (int i = 0; i < N; ++i) eps[i] = 0.0; }
    static Dual seed(double v, int k) { Dual d(v); d.eps[k] = 1.0; return d; }
};
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.re = a.re + b.re; for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] + b.eps[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.re = a.re - b.re; for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] - b.eps[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) {
    Dual<N> r; r.re = -a.re; for (int i = 0; i < N; ++i) r.eps[i] = -a.eps[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; r.re = a.re * b.re;
    for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] * b.re + a.re * b.eps[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) {
    Dual<N> r; const double inv = 1.0 / b.re; r.re = a.re * inv;
    for (int i = 0; i < N; ++i) r.eps[i] = (a.eps[i] - r.re * b.eps[i]) * inv; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.re += b; return r; }
template <int N> inline Dual<N> operator+(double a, const Dual<N>& b) { return b + a; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.re -= b; return r; }
template <int N> inline Dual<N> operator-(double a, const Dual<N>& b) { return (-b) + a; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) {
    Dual<N> r; r.re = a.re * b; for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] * b; return r; }
template <int N> inline Dual<N> operator*(double a, const Dual<N>& b) { return b * a; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { return a * (1.0 / b); }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N> sqrt(const Dual<N>& a) {
    Dual<N> r; r.re = std::sqrt(a.re); const double d = 0.5 / r.re;
    for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] * d; return r; }
template <int N> inline Dual<N> sin(const Dual<N>& a) {
    Dual<N> r; r.re = std::sin(a.re); const double d = std::cos(a.re);
    for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] * d; return r; }
template <int N> inline Dual<N> cos(const Dual<N>& a) {
    Dual<N> r; r.re = std::cos(a.re); const double d = -std::sin(a.re);
    for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] * d; return r; }
template <int N> inline Dual<N> exp(const Dual<N>& a) {
    Dual<N> r; r.re = std::exp(a.re);
    for (int i = 0; i < N; ++i) r.eps[i] = a.eps[i] * r.re; return r; }
template <int N> inline Dual<N> atan2(const Dual<N>& y, const Dual<N>& x) {
    Dual<N> r; r.re = std::atan2(y.re, x.re); const double d = 1.0 / (x.re * x.re + y.re * y.re);
    for (int i = 0; i < N; ++i) r.eps[i] = (x.re * y.eps[i] - y.re * x.eps[i]) * d; return r; }
template <int N> inline double real(const Dual<N>& a) { return a.re; }
// ---------------------------------------------------------------------------
// The same dual number with a HEAP-backed tangent vector of run-time length: what tiny-solver actually instantiates
// its factors with (num-dual's DualDVec64: every arithmetic result allocates a fresh nalgebra DVector).  Used only by
// the cpu_baseline leg of bench.py (kind "port-heap") to bracket the reference's per-corner cost from the other side:
// Dual<N> above (stack array, loops the compiler unrolls and vectorises) is the optimistic port, this the faithful one.
// An empty tangent vector is the constant's zero tangent.
// ---------------------------------------------------------------------------
struct DualH {
    double re;
    std::vector<double> eps;
    DualH() : re(0.0) {}
    DualH(double v) : re(v) {}
    static DualH seed(double v, int k, int n) { DualH d(v); d.eps.assign(n, 0.0); d.eps[k] = 1.0; return d; }
};
namespace dualh_detail {
template <class F> inline DualH map2(double re, const DualH& a, const DualH& b, F f) {   // eps_i = f(a_i, b_i), empty = zeros
    DualH r(re);
    const size_t n = std::max(a.eps.size(), b.eps.size());
    if (n) { r.eps.resize(n); for (size_t i = 0; i < n; ++i) r.eps[i] = f(i < a.eps.size() ? a.eps[i] : 0.0, i < b.eps.size() ? b.eps[i] : 0.0); }
    return r;
}
inline DualH scale(double re, const DualH& a, double d) {
    DualH r(re);
    if (!a.eps.empty()) { r.eps.resize(a.eps.size()); for (size_t i = 0; i < a.eps.size(); ++i) r.eps[i] = a.eps[i] * d; }
    return r;
}
}  // namespace dualh_detail
inline DualH operator+(const DualH& a, const DualH& b) { return dualh_detail::map2(a.re + b.re, a, b, [](double x, double y) { return x + y; }); }
inline DualH operator-(const DualH& a, const DualH& b) { return dualh_detail::map2(a.re - b.re, a, b, [](double x, double y) { return x - y; }); }
inline DualH operator-(const DualH& a) { return dualh_detail::scale(-a.re, a, -1.0); }
inline DualH operator*(const DualH& a, const DualH& b) {
    const double ar = a.re, br = b.re;
    return dualh_detail::map2(ar * br, a, b, [ar, br](double x, double y) { return x * br + ar * y; });
}
inline DualH operator/(const DualH& a, const DualH& b) {
    const double inv = 1.0 / b.re, q = a.re * inv;
    return dualh_detail::map2(q, a, b, [inv, q](double x, double y) { return (x - q * y) * inv; });
}
inline DualH operator+(const DualH& a, double b) { DualH r = a; r.re += b; return r; }
inline DualH operator+(double a, const DualH& b) { return b + a; }
inline DualH operator-(const DualH& a, double b) { DualH r = a; r.re -= b; return r; }
inline DualH operator-(double a, const DualH& b) { return (-b) + a; }
inline DualH operator*(const DualH& a, double b) { return dualh_detail::scale(a.re * b, a, b); }
inline DualH operator*(double a, const DualH& b) { return b * a; }
inline DualH operator/(const DualH& a, double b) { return a * (1.0 / b); }
inline DualH operator/(double a, const DualH& b) { return DualH(a) / b; }
inline DualH sqrt(const DualH& a) { const double s = std::sqrt(a.re); return dualh_detail::scale(s, a, 0.5 / s); }
inline DualH sin(const DualH& a) { return dualh_detail::scale(std::sin(a.re), a, std::cos(a.re)); }
inline DualH cos(const DualH& a) { return dualh_detail::scale(std::cos(a.re), a, -std::sin(a.re)); }
inline DualH exp(const DualH& a) { const double e = std::exp(a.re); return dualh_detail::scale(e, a, e); }
inline DualH atan2(const DualH& y, const DualH& x) {
    const double d = 1.0 / (x.re * x.re + y.re * y.re), xr = x.re, yr = y.re;
    return dualh_detail::map2(std::atan2(y.re, x.re), y, x, [d, xr, yr](double ye, double xe) { return (xr * ye - yr * xe) * d; });
}
inline double real(const DualH& a) { return a.re; }

inline double real(double a) { return a; }
using std::sqrt; using std::sin; using std::cos; using std::exp; using std::atan2;

// ---------------------------------------------------------------------------
// Small fixed-size algebra generic over the scalar (double or Dual<N>), the
// way the reference is generic over T: na::RealField.
// ---------------------------------------------------------------------------
template <class T> struct V3 { T x, y, z; };
template <class T> struct Quat { T w, i, j, k; };   // scalar + imaginary part
template <class T> struct Iso3 { Quat<T> q; V3<T> t; };

template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
}

// nalgebra UnitQuaternion::from_scaled_axis == Quaternion::from_imag(aa / 2).exp()
// (Isometry3::new(tvec, rvec) calls it; src/optimization/factors.rs:162, src/types.rs:28).
// exp_eps: if |v|^2 <= eps^2 -> identity (a CONSTANT: zero tangent at rvec == 0), else
// (cos|v|, v sin|v| / |v|) with exp(scalar) == exp(0) == 1 kept as in the crate.
template <class T> inline Quat<T> quat_from_scaled_axis(const V3<T>& aa) {
    const V3<T> v = { aa.x / 2.0, aa.y / 2.0, aa.z / 2.0 };
    const T nn = v.x * v.x + v.y * v.y + v.z * v.z;
    const double eps = 2.220446049250313e-16;   // f64::EPSILON
    if (real(nn) <= eps * eps) return { T(1.0), T(0.0), T(0.0), T(0.0) };
    const T w_exp = exp(T(0.0));
    const T n = sqrt(nn);
    const T s = w_exp * sin(n) / n;
    return { w_exp * cos(n), v.x * s, v.y * s, v.z * s };
}
// UnitQuaternion * Vector3:  t = 2 (q_v x p);  p + w t + q_v x t
template <class T> inline V3<T> quat_rotate(const Quat<T>& q, const V3<T>& p) {
    const V3<T> qv = { q.i, q.j, q.k };
    V3<T> t = cross(qv, p);
    t = { t.x * 2.0, t.y * 2.0, t.z * 2.0 };
    const V3<T> c = cross(qv, t);
    return { t.x * q.w + c.x + p.x, t.y * q.w + c.y + p.y, t.z * q.w + c.z + p.z };
}
template <class T> inline Quat<T> quat_mul(const Quat<T>& a, const Quat<T>& b) {
    return { a.w * b.w - a.i * b.i - a.j * b.j - a.k * b.k,
             a.w * b.i + a.i * b.w + a.j * b.k - a.k * b.j,
             a.w * b.j - a.i * b.k + a.j * b.w + a.k * b.i,
             a.w * b.k + a.i * b.j - a.j * b.i + a.k * b.w };
}
template <class T> inline Iso3<T> iso_new(const V3<T>& tvec, const V3<T>& rvec) {
    return { quat_from_scaled_axis(rvec), tvec };
}
template <class T> inline V3<T> iso_apply(const Iso3<T>& m, const V3<T>& p) {
    const V3<T> r = quat_rotate(m.q, p);
    return { r.x + m.t.x, r.y + m.t.y, r.z + m.t.z };
}
// Isometry product a * b : rotation a.q b.q, translation a.q * b.t + a.t
template <class T> inline Iso3<T> iso_mul(const Iso3<T>& a, const Iso3<T>& b) {
    const V3<T> r = quat_rotate(a.q, b.t);
    return { quat_mul(a.q, b.q), { r.x + a.t.x, r.y + a.t.y, r.z + a.t.z } };
}
template <class T> inline Iso3<T> iso_inverse(const Iso3<T>& a) {
    const Quat<T> qi = { a.q.w, -a.q.i, -a.q.j, -a.q.k };
    const V3<T> r = quat_rotate(qi, a.t);
    return { qi, { -r.x, -r.y, -r.z } };
}
// UnitQuaternion::scaled_axis (src/types.rs:57): axis * angle, angle = 2 atan2(|v|, |w|),
// axis sign follows sign of w; zero vector when |v| is ~0.
inline void quat_scaled_axis(const Quat<double>& q, double out[3]) {
    const double sgn = q.w >= 0.0 ? 1.0 : -1.0;
    const double vx = q.i * sgn, vy = q.j * sgn, vz = q.k * sgn;
    const double n = std::sqrt(vx * vx + vy * vy + vz * vz);
    if (n <= 2.220446049250313e-16) { out[0] = out[1] = out[2] = 0.0; return; }
    const double ang = 2.0 * std::atan2(n, std::fabs(q.w));
    out[0] = vx / n * ang; out[1] = vy / n * ang; out[2] = vz / n * ang;
}

// ---------------------------------------------------------------------------
// Camera models (camera-intrinsic-model GenericModel<T>::project_one).
// Parameter order: UCM [fx,fy,cx,cy,alpha] (tests/optimization_test.rs:41),
// EUCM [..,alpha,beta] (tests/util_test.rs:83-87, data/eucm.json),
// KB4 [fx,fy,cx,cy,k1..k4], OPENCV5 [fx,fy,cx,cy,k1,k2,p1,p2,k3].
// ---------------------------------------------------------------------------
enum Model { UCM = 0, EUCM = 1, KB4 = 2, OPENCV5 = 3 };
inline int model_nparams(int m) { return m == UCM ? 5 : m == EUCM ? 6 : m == KB4 ? 8 : m == OPENCV5 ? 9 : -1; }

template <class T> inline void project_one(int model, const T* p, const V3<T>& pt, T& u, T& v) {
    const T& fx = p[0]; const T& fy = p[1]; const T& cx = p[2]; const T& cy = p[3];
    const T& x = pt.x; const T& y = pt.y; const T& z = pt.z;
    if (model == UCM || model == EUCM) {
        const T& alpha = p[4];
        const T r2 = x * x + y * y;
        const T rho2 = (model == EUCM) ? p[5] * r2 + z * z : r2 + z * z;
        const T rho = sqrt(rho2);
        const T norm = alpha * rho + (1.0 - alpha) * z;
        u = fx * (x / norm) + cx;
        v = fy * (y / norm) + cy;
    } else if (model == KB4) {
        const T r2 = x * x + y * y;
        const T r = sqrt(r2);
        if (real(r) > 1e-8) {   // KB "r -> 0" guard: pinhole limit below it
            const T th = atan2(r, z);
            const T th2 = th * th;
            T poly = p[7] * th2;
            poly = (poly + p[6]) * th2;
            poly = (poly + p[5]) * th2;
            poly = (poly + p[4]) * th2;
            const T thd = th * (poly + 1.0);
            u = fx * (thd * x / r) + cx;
            v = fy * (thd * y / r) + cy;
        } else {
            u = fx * (x / z) + cx;
            v = fy * (y / z) + cy;
        }
    } else {   // OPENCV5
        const T& k1 = p[4]; const T& k2 = p[5]; const T& p1 = p[6]; const T& p2 = p[7]; const T& k3 = p[8];
        const T xn = x / z, yn = y / z;
        const T r2 = xn * xn + yn * yn;
        const T rad = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
        const T xy = xn * yn;
        const T xd = xn * rad + 2.0 * p1 * xy + p2 * (r2 + 2.0 * xn * xn);
        const T yd = yn * rad + p1 * (r2 + 2.0 * yn * yn) + 2.0 * p2 * xy;
        u = fx * xd + cx;
        v = fy * yd + cy;
    }
}

// ---------------------------------------------------------------------------
// GenericModel::unproject / project validity (camera-intrinsic-model, not under /root/reference: restated from
// the models' published definitions - UCM/EUCM: Usenko et al. 2018, "The Double Sphere Camera Model", sec. II;
// KB4: Kannala-Brandt 2006 with Newton on theta; OPENCV5: fixed-point undistortion).  Used only by the
// ModelConvertFactor restatement (src/optimization/factors.rs:23-54).  Parity unpinned for the crate's
// exact validity thresholds and iteration counts.
// ---------------------------------------------------------------------------
inline bool unproject_one(int model, const double* p, double u, double v, double ray[3]) {
    const double mx = (u - p[2]) / p[0], my = (v - p[3]) / p[1];
    const double r2 = mx * mx + my * my;
    if (model == UCM || model == EUCM) {
        const double alpha = p[4], beta = model == EUCM ? p[5] : 1.0;
        if (alpha > 0.5 && r2 > 1.0 / (beta * (2.0 * alpha - 1.0))) return false;
        const double t1 = 1.0 - (2.0 * alpha - 1.0) * beta * r2;
        if (t1 < 0.0) return false;
        const double mz = (1.0 - beta * alpha * alpha * r2) / (alpha * std::sqrt(t1) + (1.0 - alpha));
        const double n = std::sqrt(r2 + mz * mz);
        ray[0] = mx / n; ray[1] = my / n; ray[2] = mz / n;
        return true;
    }
    if (model == KB4) {
        const double r = std::sqrt(r2);
        if (r < 1e-8) { ray[0] = mx; ray[1] = my; ray[2] = 1.0; return true; }
        double t = r;
        for (int it = 0; it < 20; ++it) {
            const double t2 = t * t;
            const double f = t * (1.0 + t2 * (p[4] + t2 * (p[5] + t2 * (p[6] + t2 * p[7])))) - r;
            const double fp = 1.0 + t2 * (3.0 * p[4] + t2 * (5.0 * p[5] + t2 * (7.0 * p[6] + t2 * 9.0 * p[7])));
            const double dt = f / fp;
            t -= dt;
            if (std::fabs(dt) < 1e-14) break;
        }
        if (!(t > 0.0) || !(t < 3.141592653589793)) return false;
        const double s = std::sin(t) / r;
        ray[0] = mx * s; ray[1] = my * s; ray[2] = std::cos(t);
        return true;
    }
    if (model == OPENCV5) {
        const double k1 = p[4], k2 = p[5], p1 = p[6], p2 = p[7], k3 = p[8];
        double x = mx, y = my;
        for (int it = 0; it < 50; ++it) {
            const double q = x * x + y * y;
            const double rad = 1.0 + q * (k1 + q * (k2 + q * k3));
            const double dx = 2.0 * p1 * x * y + p2 * (q + 2.0 * x * x);
            const double dy = p1 * (q + 2.0 * y * y) + 2.0 * p2 * x * y;
            x = (mx - dx) / rad; y = (my - dy) / rad;
        }
        const double q = x * x + y * y, rad = 1.0 + q * (k1 + q * (k2 + q * k3));
        const double ex = x * rad + 2.0 * p1 * x * y + p2 * (q + 2.0 * x * x) - mx;
        const double ey = y * rad + p1 * (q + 2.0 * y * y) + 2.0 * p2 * x * y - my;
        if (!(std::fabs(ex) + std::fabs(ey) < 1e-9)) return false;
        const double n = std::sqrt(q + 1.0);
        ray[0] = x / n; ray[1] = y / n; ray[2] = 1.0 / n;
        return true;
    }
    return false;
}
// Is `project` defined for this camera-frame point?  (The `Option` of GenericModel::project.)
inline bool project_valid(int model, const double* p, double x, double y, double z) {
    if (model == UCM || model == EUCM) {
        const double alpha = p[4], beta = model == EUCM ? p[5] : 1.0;
        const double d = std::sqrt(beta * (x * x + y * y) + z * z);
        const double w = alpha <= 0.5 ? alpha / (1.0 - alpha) : (1.0 - alpha) / alpha;
        return z > -w * d;
    }
    if (model == KB4) return x * x + y * y + z * z > 0.0;
    return z > 1e-9;
}

// ---------------------------------------------------------------------------
// The two factors, line for line.  `params` is the solver-visible intrinsic
// block (P_eff entries: fy removed when xy_same_focal), then rvec, tvec
// (and rvec_i_0, tvec_i_0 for the other-camera factor).
// ---------------------------------------------------------------------------
template <class T>
inline void reprojection_factor(int model, bool xy_same_focal, const T* params /*P_eff*/,
                                const T* rvec, const T* tvec,
                                const float p3d[3], const float p2d[2], T r[2]) {
    T full[9];                                    // factors.rs:155-158 insert_row(1, params0[0])
    const int P = model_nparams(model);
    if (xy_same_focal) { full[0] = params[0]; full[1] = params[0]; for (int i = 2; i < P; ++i) full[i] = params[i - 1]; }
    else { for (int i = 0; i < P; ++i) full[i] = params[i]; }
    const Iso3<T> tr = iso_new<T>({ tvec[0], tvec[1], tvec[2] }, { rvec[0], rvec[1], rvec[2] });   // :162
    const V3<T> pw = { T((double)p3d[0]), T((double)p3d[1]), T((double)p3d[2]) };                    // :141-143 f32 -> f64
    const V3<T> pc = iso_apply(tr, pw);                                                                // :163
    T u, v; project_one<T>(model, full, pc, u, v);                                                     // :165
    r[0] = u - (double)p2d[0];                                                                         // :167-171
    r[1] = v - (double)p2d[1];
}
template <class T>
inline void other_cam_reprojection_factor(int model, bool xy_same_focal, const T* params,
                                          const T* rvec0, const T* tvec0, const T* rvec1, const T* tvec1,
                                          const float p3d[3], const float p2d[2], T r[2]) {
    T full[9];
    const int P = model_nparams(model);
    if (xy_same_focal) { full[0] = params[0]; full[1] = params[0]; for (int i = 2; i < P; ++i) full[i] = params[i - 1]; }
    else { for (int i = 0; i < P; ++i) full[i] = params[i]; }
    const Iso3<T> t_0_b = iso_new<T>({ tvec0[0], tvec0[1], tvec0[2] }, { rvec0[0], rvec0[1], rvec0[2] });   // :214
    const Iso3<T> t_i_0 = iso_new<T>({ tvec1[0], tvec1[1], tvec1[2] }, { rvec1[0], rvec1[1], rvec1[2] });   // :217
    const V3<T> pw = { T((double)p3d[0]), T((double)p3d[1]), T((double)p3d[2]) };
    const V3<T> pc = iso_apply(iso_mul(t_i_0, t_0_b), pw);                                                   // :218
    T u, v; project_one<T>(model, full, pc, u, v);
    r[0] = u - (double)p2d[0];
    r[1] = v - (double)p2d[1];
}

// Residual + Jacobian of one block by seeding a Dual<D> per variable, exactly
// what tiny-solver does per residual block.  J is 2 x D row-major,
// column order [params, rvec, tvec(, rvec_i_0, tvec_i_0)] (src/util.rs:411, 621-627).
template <int D>
inline void factor_jacobian(int model, bool xy_same_focal, bool other_cam, int p_eff,
                            const double* params, const double* pose0 /*rvec,tvec*/,
                            const double* pose1 /*rvec_i_0,tvec_i_0 or null*/,
                            const float p3d[3], const float p2d[2], double r[2], double* J) {
    using DT = Dual<D>;
    DT pv[9], a0[3], b0[3], a1[3], b1[3];
    int k = 0;
    for (int i = 0; i < p_eff; ++i) pv[i] = DT::seed(params[i], k++);
    for (int i = 0; i < 3; ++i) a0[i] = DT::seed(pose0[i], k++);
    for (int i = 0; i < 3; ++i) b0[i] = DT::seed(pose0[3 + i], k++);
    DT rr[2];
    if (other_cam) {
        for (int i = 0; i < 3; ++i) a1[i] = DT::seed(pose1[i], k++);
        for (int i = 0; i < 3; ++i) b1[i] = DT::seed(pose1[3 + i], k++);
        other_cam_reprojection_factor<DT>(model, xy_same_focal, pv, a0, b0, a1, b1, p3d, p2d, rr);
    } else {
        reprojection_factor<DT>(model, xy_same_focal, pv, a0, b0, p3d, p2d, rr);
    }
    for (int row = 0; row < 2; ++row) {
        r[row] = rr[row].re;
        for (int c = 0; c < D; ++c) J[row * D + c] = rr[row].eps[c];
    }
}

// The same block with heap-backed duals of run-time length D (DualH): the container tiny-solver uses.
inline void factor_jacobian_heap(int D, int model, bool xy_same_focal, bool other_cam, int p_eff,
                                 const double* params, const double* pose0, const double* pose1,
                                 const float p3d[3], const float p2d[2], double r[2], double* J) {
    DualH pv[9], a0[3], b0[3], a1[3], b1[3];
    int k = 0;
    for (int i = 0; i < p_eff; ++i) pv[i] = DualH::seed(params[i], k++, D);
    for (int i = 0; i < 3; ++i) a0[i] = DualH::seed(pose0[i], k++, D);
    for (int i = 0; i < 3; ++i) b0[i] = DualH::seed(pose0[3 + i], k++, D);
    DualH rr[2];
    if (other_cam) {
        for (int i = 0; i < 3; ++i) a1[i] = DualH::seed(pose1[i], k++, D);
        for (int i = 0; i < 3; ++i) b1[i] = DualH::seed(pose1[3 + i], k++, D);
        other_cam_reprojection_factor<DualH>(model, xy_same_focal, pv, a0, b0, a1, b1, p3d, p2d, rr);
    } else {
        reprojection_factor<DualH>(model, xy_same_focal, pv, a0, b0, p3d, p2d, rr);
    }
    for (int row = 0; row < 2; ++row) {
        r[row] = rr[row].re;
        for (int c = 0; c < D; ++c) J[row * D + c] = c < (int)rr[row].eps.size() ? rr[row].eps[c] : 0.0;
    }
}

// Runtime-D dispatch (D = P_eff + 6 or P_eff + 12, P_eff in 4..9).
void factor_jacobian_dyn(int D, int model, bool xy_same_focal, bool other_cam, int p_eff,
                         const double* params, const double* pose0, const double* pose1,
                         const float p3d[3], const float p2d[2], double r[2], double* J);

// UnitQuaternion::scaled_axis generic over the scalar (needed with duals by SE3Factor).
template <class T> inline void quat_scaled_axis_t(const Quat<T>& q, T out[3]) {
    const double sgn = real(q.w) >= 0.0 ? 1.0 : -1.0;
    const T vx = q.i * sgn, vy = q.j * sgn, vz = q.k * sgn;
    const T n = sqrt(vx * vx + vy * vy + vz * vz);
    if (real(n) <= 2.220446049250313e-16) { out[0] = T(0.0); out[1] = T(0.0); out[2] = T(0.0); return; }
    const T ang = atan2(n, q.w * sgn) * 2.0;
    out[0] = vx / n * ang; out[1] = vy / n * ang; out[2] = vz / n * ang;
}

// SE3Factor::residual_func (src/optimization/factors.rs:248-271):
// t_diff = t_i_b^-1 * Isometry3::new(tvec, rvec) * t_0_b ; [scaled_axis(rotation), translation]
template <class T>
inline void se3_factor(const double* pose_0_b, const double* pose_i_b, const T* rvec, const T* tvec, T r[6]) {
    const Iso3<T> t_0_b = iso_new<T>({ T(pose_0_b[3]), T(pose_0_b[4]), T(pose_0_b[5]) }, { T(pose_0_b[0]), T(pose_0_b[1]), T(pose_0_b[2]) });
    const Iso3<T> t_i_b = iso_new<T>({ T(pose_i_b[3]), T(pose_i_b[4]), T(pose_i_b[5]) }, { T(pose_i_b[0]), T(pose_i_b[1]), T(pose_i_b[2]) });
    const Iso3<T> t_i_0 = iso_new<T>({ tvec[0], tvec[1], tvec[2] }, { rvec[0], rvec[1], rvec[2] });
    const Iso3<T> d = iso_mul(iso_mul(iso_inverse(t_i_b), t_i_0), t_0_b);
    quat_scaled_axis_t(d.q, r);
    r[3] = d.t.x; r[4] = d.t.y; r[5] = d.t.z;
}

// Huber loss as tiny-solver evaluates it (HuberLoss::new(1.0), src/util.rs:413):
// rho'(s) = 1 for s <= delta^2, delta / sqrt(s) otherwise; corrector scales r and J by sqrt(rho').
inline double huber_weight(double s, double delta) {
    return (s <= delta * delta) ? 1.0 : delta / std::sqrt(s);
}

}  // namespace oracle
