// ccal.hpp -- header-only C++17 mirror of the reference's calib-frame API on top of the C ABI (ccal.h).
//
// The reference is Rust (no toolchain in this image), so the host side a Rust maintainer would write
// is given here in C++ with the reference's names, argument meaning and error behaviour:
//
//   reference (Rust)                                               here (namespace ccal)
//   -----------------------------------------------------------   ----------------------------------------
//   detected_points::FeaturePoint / FrameFeature                   FeaturePoint / FrameFeature
//   types::RvecTvec, to_na_isometry3, ToRvecTvec                    RvecTvec (+ compose / inverse)
//   camera_intrinsic_model::GenericModel<f64>                       GenericModel
//   factors::ReprojectionFactor / OtherCamReprojectionFactor        ReprojectionFactor / OtherCamReprojectionFactor
//   util::calib_camera                  src/util.rs:384-490         calib_camera            -> std::optional
//   util::calib_all_camera_with_extrinsics   src/util.rs:567-715    calib_all_camera_with_extrinsics
//   util::init_camera_extrinsic         src/util.rs:511-561         init_camera_extrinsic
//   util::init_ucm                      src/util.rs:287-378         init_ucm                -> std::optional
//   util::validation                    src/util.rs:721-795         validation
//   util::convert_model                 src/util.rs:224-282         convert_model
//   io::object_to_json / write_report   src/io.rs, src/types.rs     model_to_json / poses_to_json / extrinsics_to_json / report_text (+ *_from_json)
//
// `None` of the reference == std::nullopt here; nothing falls back to a CPU implementation: every
// numeric call goes through libccal_hip.so.  Corner order inside a frame is by corner id (std::map), the
// reference iterates a HashMap (order is random per process there; results agree at the optimum).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <optional>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ccal.h"

namespace ccal {

struct FeaturePoint { std::array<float, 2> p2d; std::array<float, 3> p3d; };                    // src/detected_points.rs:6-9
struct FrameFeature { int64_t time_ns = 0; std::pair<uint32_t, uint32_t> img_w_h{0, 0}; std::map<uint32_t, FeaturePoint> features; };

struct RvecTvec {                                                                                // src/types.rs:13-36
    std::array<double, 3> rvec{0, 0, 0}, tvec{0, 0, 0};
    std::array<double, 6> as6() const { return {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]}; }
    static RvecTvec from6(const double* v) { RvecTvec r; for (int i = 0; i < 3; ++i) { r.rvec[i] = v[i]; r.tvec[i] = v[3 + i]; } return r; }
    // unit quaternion (w, x, y, z) of the rotation, as nalgebra's UnitQuaternion::from_scaled_axis
    std::array<double, 4> quat() const {
        const double hx = 0.5 * rvec[0], hy = 0.5 * rvec[1], hz = 0.5 * rvec[2], nn = hx * hx + hy * hy + hz * hz;
        if (nn <= 4.930380657631324e-32) return {1, 0, 0, 0};
        const double n = std::sqrt(nn), s = std::sin(n) / n;
        return {std::cos(n), hx * s, hy * s, hz * s};
    }
    static std::array<double, 3> rotate(const std::array<double, 4>& q, const std::array<double, 3>& p) {
        const double tx = 2 * (q[2] * p[2] - q[3] * p[1]), ty = 2 * (q[3] * p[0] - q[1] * p[2]), tz = 2 * (q[1] * p[1] - q[2] * p[0]);
        return {p[0] + q[0] * tx + (q[2] * tz - q[3] * ty), p[1] + q[0] * ty + (q[3] * tx - q[1] * tz), p[2] + q[0] * tz + (q[1] * ty - q[2] * tx)};
    }
    static RvecTvec from_quat(const std::array<double, 4>& q, const std::array<double, 3>& t) {   // ToRvecTvec, src/types.rs:55-64
        RvecTvec r; r.tvec = t;
        const double sg = q[0] >= 0 ? 1.0 : -1.0, vx = q[1] * sg, vy = q[2] * sg, vz = q[3] * sg, n = std::sqrt(vx * vx + vy * vy + vz * vz);
        if (n > 2.220446049250313e-16) { const double a = 2.0 * std::atan2(n, std::fabs(q[0])) / n; r.rvec = {vx * a, vy * a, vz * a}; }
        return r;
    }
    RvecTvec compose(const RvecTvec& o) const {                                                   // self * o
        const auto a = quat(), b = o.quat();
        const std::array<double, 4> q = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
        auto t = rotate(a, o.tvec); for (int i = 0; i < 3; ++i) t[i] += tvec[i];
        return from_quat(q, t);
    }
    RvecTvec inverse() const {
        const auto a = quat(); const std::array<double, 4> qi = {a[0], -a[1], -a[2], -a[3]};
        auto t = rotate(qi, tvec); for (auto& v : t) v = -v;
        return from_quat(qi, t);
    }
};

class GenericModel {                                                                             // camera_intrinsic_model::GenericModel<f64>
public:
    GenericModel(int model, std::vector<double> params, double width, double height) : model_(model), p_(std::move(params)), w_(width), h_(height) {
        if (ccal_model_num_params(model) != (int)p_.size()) throw std::invalid_argument("wrong number of parameters");
    }
    int model_id() const { return model_; }
    const std::vector<double>& params() const { return p_; }
    void set_params(const std::vector<double>& p) { p_ = p; }
    double width() const { return w_; }
    double height() const { return h_; }
private:
    int model_; std::vector<double> p_; double w_, h_;
};

// Which GPUs a call may use: one device (an int) or several ({0, 1, ..., 7}) - still ONE call of ONE process like the
// reference's (src/util.rs:384-390); with several the library shards the frame slots and sums the reduced system once per
// step (ccal_multi_*, include/ccal.h).  A device listed twice = two shards on that GPU.
// The optimizer's options every entry point below solves with: GaussNewtonOptimizer::default() (src/util.rs:443,455) as ccal_set_defaults
// restates it.  A host changes them once - e.g. `ccal::optimizer_options().error_metric = CCAL_ERROR_NORM;` if its tiny-solver's
// compute_error is the norm rather than the squared norm ("Stop rule").
inline ccal_solver_opts& optimizer_options() {
    static ccal_solver_opts o = [] { ccal_solver_opts d; ccal_set_defaults(&d); return d; }();
    return o;
}

struct Devices {
    std::vector<int> ids;
    Devices(int device = 0) : ids{device} {}
    Devices(std::vector<int> v) : ids(std::move(v)) {}
    Devices(std::initializer_list<int> l) : ids(l) {}
};

namespace detail {

struct Multi {   // RAII over ccal_multi
    ccal_multi* h = nullptr;
    explicit Multi(const Devices& d) { if (ccal_multi_create(d.ids.data(), (int)d.ids.size(), &h) != CCAL_OK) throw std::runtime_error("ccal_multi_create failed: no usable HIP device / library / transport"); }
    ~Multi() { ccal_multi_destroy(h); }
    Multi(const Multi&) = delete; Multi& operator=(const Multi&) = delete;
};
struct MProb {
    ccal_multi_problem* h = nullptr;
    ~MProb() { ccal_multi_problem_destroy(h); }
};

struct Ctx {     // RAII over ccal_ctx
    ccal_ctx* h = nullptr;
    explicit Ctx(int device = 0) { if (ccal_ctx_create(device, nullptr, &h) != CCAL_OK) throw std::runtime_error("ccal_ctx_create failed: no usable HIP device / library"); }
    ~Ctx() { ccal_ctx_destroy(h); }
    Ctx(const Ctx&) = delete; Ctx& operator=(const Ctx&) = delete;
};
struct Prob {
    ccal_problem* h = nullptr;
    ~Prob() { ccal_problem_destroy(h); }
};

// (camera, frame index) observation frames -> CSR + SoA; slots = sorted union of the frame indices
struct Flat {
    std::vector<size_t> slots; std::vector<int32_t> obs_cam, obs_slot; std::vector<int64_t> offs{0};
    std::vector<float> x, y, z, u, v;
};
inline Flat flatten(const std::vector<const std::vector<std::optional<FrameFeature>>*>& cams, const std::vector<std::set<size_t>>& use) {
    Flat f; std::set<size_t> all;
    for (auto& s : use) all.insert(s.begin(), s.end());
    f.slots.assign(all.begin(), all.end());
    for (size_t s = 0; s < f.slots.size(); ++s) {
        const size_t fi = f.slots[s];
        for (size_t c = 0; c < cams.size(); ++c) {
            if (!use[c].count(fi)) continue;
            const FrameFeature& ff = *(*cams[c])[fi];
            for (auto& kv : ff.features) {
                f.x.push_back(kv.second.p3d[0]); f.y.push_back(kv.second.p3d[1]); f.z.push_back(kv.second.p3d[2]);
                f.u.push_back(kv.second.p2d[0]); f.v.push_back(kv.second.p2d[1]);
            }
            f.obs_cam.push_back((int32_t)c); f.obs_slot.push_back((int32_t)s); f.offs.push_back((int64_t)f.x.size());
        }
    }
    return f;
}
inline int make_problem(Ctx& ctx, const Flat& f, const std::vector<GenericModel>& cams, bool xy_same_focal, Prob& out) {
    std::vector<int32_t> model; std::vector<double> w, h;
    for (auto& m : cams) { model.push_back(m.model_id()); w.push_back(m.width()); h.push_back(m.height()); }
    ccal_problem_desc d{};
    d.n_cams = (int32_t)cams.size(); d.model = model.data(); d.width = w.data(); d.height = h.data();
    d.xy_same_focal = xy_same_focal; d.n_slots = (int32_t)f.slots.size(); d.n_obs = (int32_t)f.obs_cam.size();
    d.obs_cam = f.obs_cam.data(); d.obs_slot = f.obs_slot.data(); d.obs_offsets = f.offs.data();
    d.p3d_x = f.x.data(); d.p3d_y = f.y.data(); d.p3d_z = f.z.data(); d.p2d_u = f.u.data(); d.p2d_v = f.v.data();
    d.huber_delta = 1.0;                                             // HuberLoss::new(1.0), src/util.rs:413
    return ccal_problem_create(ctx.h, &d, &out.h);
}
inline int make_problem(Multi& m, const Flat& f, const std::vector<GenericModel>& cams, bool xy_same_focal, MProb& out) {
    std::vector<int32_t> model; std::vector<double> w, h;
    for (auto& c : cams) { model.push_back(c.model_id()); w.push_back(c.width()); h.push_back(c.height()); }
    ccal_problem_desc d{};
    d.n_cams = (int32_t)cams.size(); d.model = model.data(); d.width = w.data(); d.height = h.data();
    d.xy_same_focal = xy_same_focal; d.n_slots = (int32_t)f.slots.size(); d.n_obs = (int32_t)f.obs_cam.size();
    d.obs_cam = f.obs_cam.data(); d.obs_slot = f.obs_slot.data(); d.obs_offsets = f.offs.data();
    d.p3d_x = f.x.data(); d.p3d_y = f.y.data(); d.p3d_z = f.z.data(); d.p2d_u = f.u.data(); d.p2d_v = f.v.data();
    d.huber_delta = 1.0;                                             // HuberLoss::new(1.0), src/util.rs:413
    return ccal_multi_problem_create(m.h, &d, &out.h);
}
inline std::vector<double> intr_matrix(const std::vector<GenericModel>& cams) {
    std::vector<double> intr(cams.size() * CCAL_PMAX, 0.0);
    for (size_t c = 0; c < cams.size(); ++c) for (size_t i = 0; i < cams[c].params().size(); ++i) intr[c * CCAL_PMAX + i] = cams[c].params()[i];
    return intr;
}

}  // namespace detail

// The pose initialisation inside calib_camera (src/util.rs:418-436): unproject, keep valid, planar PnP.
inline std::map<size_t, RvecTvec> init_frame_poses(const std::vector<std::optional<FrameFeature>>& frames, const GenericModel& cam, int min_points = 10, const Devices& device = 0) {
    std::set<size_t> valid;
    for (size_t i = 0; i < frames.size(); ++i) if (frames[i]) valid.insert(i);
    std::map<size_t, RvecTvec> out;
    if (valid.empty()) return out;
    detail::Multi ctx(device); detail::MProb p;
    const auto f = detail::flatten({&frames}, {valid});
    if (detail::make_problem(ctx, f, {cam}, false, p) != CCAL_OK) return out;
    std::vector<double> poses(f.slots.size() * 6); std::vector<int32_t> used(f.slots.size());
    const auto intr = detail::intr_matrix({cam});
    if (ccal_multi_init_poses(p.h, intr.data(), min_points, poses.data(), used.data()) != CCAL_OK) return out;
    for (size_t s = 0; s < f.slots.size(); ++s) if (used[s] > 0) out[f.slots[s]] = RvecTvec::from6(&poses[6 * s]);
    return out;
}

// util::calib_camera (src/util.rs:384-490).  `initial_poses == nullptr` initialises the poses inside, as the reference does.
inline std::optional<std::pair<GenericModel, std::map<size_t, RvecTvec>>>
calib_camera(const std::vector<std::optional<FrameFeature>>& frame_feature_list, const GenericModel& generic_camera,
             bool xy_same_focal, size_t disabled_distortions, bool fixed_focal,
             const std::map<size_t, RvecTvec>* initial_poses = nullptr, const Devices& device = 0) {
    std::map<size_t, RvecTvec> init_local;
    if (!initial_poses) { init_local = init_frame_poses(frame_feature_list, generic_camera, 10, device); initial_poses = &init_local; }
    std::set<size_t> valid;
    for (size_t i = 0; i < frame_feature_list.size(); ++i) if (frame_feature_list[i] && initial_poses->count(i)) valid.insert(i);
    if (valid.empty()) return std::nullopt;
    detail::Multi ctx(device); detail::MProb p;
    const auto f = detail::flatten({&frame_feature_list}, {valid});
    if (detail::make_problem(ctx, f, {generic_camera}, xy_same_focal, p) != CCAL_OK) return std::nullopt;
    auto intr = detail::intr_matrix({generic_camera});
    std::vector<double> poses;
    for (size_t fi : f.slots) { const auto v = initial_poses->at(fi).as6(); poses.insert(poses.end(), v.begin(), v.end()); }
    ccal_multi_apply_reference_bounds(p.h);                                         // src/util.rs:446
    ccal_multi_disable_distortions(p.h, (int)disabled_distortions, intr.data());    // src/util.rs:447-454
    ccal_solver_opts o = optimizer_options();                                      // GaussNewtonOptimizer::default()
    ccal_report rep{};
    int rc = ccal_multi_solve(p.h, &o, intr.data(), poses.data(), nullptr, &rep);   // :455 - one call, every listed GPU
    if (rc != CCAL_OK && rc != CCAL_ERR_NO_CONVERGENCE) return std::nullopt;        // result_option.as_ref()?
    if (fixed_focal) {                                                              // :459-464
        ccal_multi_fix_param(p.h, 0, 0);
        intr[0] = generic_camera.params()[0]; if (xy_same_focal) intr[1] = intr[0];
        rc = ccal_multi_solve(p.h, &o, intr.data(), poses.data(), nullptr, &rep);
        if (rc != CCAL_OK && rc != CCAL_ERR_NO_CONVERGENCE) throw std::runtime_error("second solve failed");   // .unwrap()
    }
    GenericModel out = generic_camera;
    out.set_params(std::vector<double>(intr.begin(), intr.begin() + generic_camera.params().size()));
    std::map<size_t, RvecTvec> rt;
    for (size_t s = 0; s < f.slots.size(); ++s) rt[f.slots[s]] = RvecTvec::from6(&poses[6 * s]);
    return std::make_pair(out, rt);
}

struct AllCameraResult { std::vector<GenericModel> intrinsics; std::vector<RvecTvec> t_i_0; std::map<size_t, RvecTvec> board_poses; };

// util::calib_all_camera_with_extrinsics (src/util.rs:567-715)
inline std::optional<AllCameraResult>
calib_all_camera_with_extrinsics(const std::vector<GenericModel>& cameras, const std::vector<RvecTvec>& t_cam_i_0,
                                 const std::vector<std::map<size_t, RvecTvec>>& cam_rtvecs,
                                 const std::vector<std::vector<std::optional<FrameFeature>>>& cams_detected_feature_frames,
                                 bool xy_same_focal, size_t disabled_distortions, bool cam0_fixed_focal, const Devices& device = 0) {
    const size_t n = cameras.size();
    std::vector<std::set<size_t>> use(n);
    std::vector<const std::vector<std::optional<FrameFeature>>*> fr;
    for (size_t c = 0; c < n; ++c) { for (auto& kv : cam_rtvecs[c]) use[c].insert(kv.first); fr.push_back(&cams_detected_feature_frames[c]); }
    const auto f = detail::flatten(fr, use);
    if (f.slots.empty()) return std::nullopt;
    detail::Multi ctx(device); detail::MProb p;
    if (detail::make_problem(ctx, f, cameras, xy_same_focal, p) != CCAL_OK) return std::nullopt;
    auto intr = detail::intr_matrix(cameras);
    std::vector<double> extr(n * 6, 0.0), poses(f.slots.size() * 6, 0.0);
    for (size_t c = 1; c < n; ++c) { const auto v = t_cam_i_0[c].as6(); for (int i = 0; i < 6; ++i) extr[c * 6 + i] = v[i]; }
    for (size_t s = 0; s < f.slots.size(); ++s)                                      // `.entry().or_insert()` in camera order, :633-651
        for (size_t c = 0; c < n; ++c) {
            auto it = cam_rtvecs[c].find(f.slots[s]);
            if (it == cam_rtvecs[c].end()) continue;
            const auto v = (c == 0 ? it->second : t_cam_i_0[c].inverse().compose(it->second)).as6();
            for (int i = 0; i < 6; ++i) poses[6 * s + i] = v[i];
            break;
        }
    ccal_multi_apply_reference_bounds(p.h);
    ccal_multi_disable_distortions(p.h, (int)disabled_distortions, intr.data());
    if (cam0_fixed_focal) ccal_multi_fix_param(p.h, 0, 0);                           // :664-667
    ccal_solver_opts o = optimizer_options();
    ccal_report rep{};
    const int rc = ccal_multi_solve(p.h, &o, intr.data(), poses.data(), extr.data(), &rep);
    if (rc != CCAL_OK && rc != CCAL_ERR_NO_CONVERGENCE) return std::nullopt;
    AllCameraResult r;
    for (size_t c = 0; c < n; ++c) {
        GenericModel m = cameras[c];
        m.set_params(std::vector<double>(intr.begin() + c * CCAL_PMAX, intr.begin() + c * CCAL_PMAX + cameras[c].params().size()));
        r.intrinsics.push_back(m);
        r.t_i_0.push_back(c == 0 ? RvecTvec{} : RvecTvec::from6(&extr[c * 6]));
    }
    for (size_t s = 0; s < f.slots.size(); ++s) r.board_poses[f.slots[s]] = RvecTvec::from6(&poses[6 * s]);
    return r;
}

// util::init_camera_extrinsic (src/util.rs:511-561)
inline std::vector<RvecTvec> init_camera_extrinsic(const std::vector<std::map<size_t, RvecTvec>>& cam_rtvecs) {
    std::vector<RvecTvec> out(1);
    for (size_t ci = 1; ci < cam_rtvecs.size(); ++ci) {
        std::vector<double> p0, pi;
        for (auto& kv : cam_rtvecs[0]) {
            auto it = cam_rtvecs[ci].find(kv.first);
            if (it == cam_rtvecs[ci].end()) continue;
            const auto a = kv.second.as6(), b = it->second.as6();
            p0.insert(p0.end(), a.begin(), a.end()); pi.insert(pi.end(), b.begin(), b.end());
        }
        if (p0.empty()) throw std::runtime_error("camera shares no frame with camera 0");
        double x[6] = {0, 0, 0, 0, 0, 0};
        if (ccal_init_camera_extrinsic(p0.data(), pi.data(), (int)(p0.size() / 6), x, 0, nullptr) != CCAL_OK) throw std::runtime_error("init_camera_extrinsic failed");
        out.push_back(RvecTvec::from6(x));
    }
    return out;
}

// util::init_ucm (src/util.rs:287-378).  UCMInitFocalAlphaFactor (src/optimization/factors.rs:82-120) is the reprojection
// factor of a UCM whose only free intrinsics are f = fx = fy and alpha, the principal point pinned at the image centre:
// the engine's UCM + xy_same_focal problem with cx, cy fixed, f in [f0 / 3, 3 f0], alpha in [1e-6, 1] (:345-346) on the two
// frames; then calib_camera on those two frames with xy_same_focal = true (:365-371).  `None` of the first solve ==
// std::nullopt; a failing calib_camera is the reference's `.expect(...)`: it throws.
inline std::optional<GenericModel>
init_ucm(const FrameFeature& frame_feature0, const FrameFeature& frame_feature1, const RvecTvec& rtvec0, const RvecTvec& rtvec1,
         double init_f, double init_alpha, bool fixed_focal, int device = 0) {
    const double w = frame_feature0.img_w_h.first, h = frame_feature0.img_w_h.second;
    const GenericModel ucm0(CCAL_MODEL_UCM, {init_f, init_f, w / 2.0, h / 2.0, init_alpha}, w, h);
    const std::vector<std::optional<FrameFeature>> frames = {frame_feature0, frame_feature1};
    std::vector<double> intr;
    {
        detail::Ctx ctx(device); detail::Prob p;
        const auto f = detail::flatten({&frames}, {{0, 1}});
        if (detail::make_problem(ctx, f, {ucm0}, true, p) != CCAL_OK) return std::nullopt;
        ccal_fix_param(p.h, 0, 1); ccal_fix_param(p.h, 0, 2);                      // cx, cy (eff indices: fy removed)
        if (fixed_focal) ccal_fix_param(p.h, 0, 0);
        ccal_set_bounds(p.h, 0, 0, init_f / 3.0, init_f * 3.0);
        ccal_set_bounds(p.h, 0, 3, 1e-6, 1.0);
        intr = detail::intr_matrix({ucm0});
        std::vector<double> poses;
        for (const RvecTvec* rt : {&rtvec0, &rtvec1}) { const auto v = rt->as6(); poses.insert(poses.end(), v.begin(), v.end()); }
        ccal_solver_opts o = optimizer_options();
        ccal_report rep{};
        const int rc = ccal_solve(p.h, &o, intr.data(), poses.data(), nullptr, &rep);
        if (rc != CCAL_OK && rc != CCAL_ERR_NO_CONVERGENCE) return std::nullopt;
    }
    const GenericModel ucm1(CCAL_MODEL_UCM, {intr[0], intr[0], w / 2.0, h / 2.0, intr[4]}, w, h);
    const auto res = calib_camera(frames, ucm1, true, 0, fixed_focal, nullptr, device);
    if (!res) throw std::runtime_error("The initial UCM model fitting failed. Might be wrong board configuration.");
    return res->first;
}

// util::convert_model (src/util.rs:224-282): fits `target_model` in place, like the reference's &mut argument
inline void convert_model(const GenericModel& source_model, GenericModel& target_model, size_t disabled_distortions, int device = 0) {
    if ((uint32_t)std::lround(source_model.width()) != (uint32_t)std::lround(target_model.width())) throw std::invalid_argument("source width and target width are not the same.");      // factors.rs:29-33: panic!
    if ((uint32_t)std::lround(source_model.height()) != (uint32_t)std::lround(target_model.height())) throw std::invalid_argument("source height and target height are not the same.");
    if (source_model.model_id() == CCAL_MODEL_UCM && (target_model.model_id() == CCAL_MODEL_EUCM || target_model.model_id() == CCAL_MODEL_EUCMT)) {
        std::vector<double> t = source_model.params();                    // closed forms, src/util.rs:229-243: no device needed
        t.push_back(1.0);
        if (target_model.model_id() == CCAL_MODEL_EUCMT) { t.push_back(0.0); t.push_back(0.0); }
        target_model.set_params(t);
        return;
    }
    detail::Ctx ctx(device);
    std::vector<double> t = target_model.params();
    const int rc = ccal_convert_model(ctx.h, source_model.model_id(), source_model.params().data(), target_model.model_id(), t.data(),
                                      source_model.width(), source_model.height(), (int)disabled_distortions, nullptr, nullptr);
    if (rc != CCAL_OK) throw std::runtime_error(std::string("convert_model failed: ") + ccal_last_error(ctx.h));       // `.unwrap()` (util.rs:276)
    target_model.set_params(t);
}

// util::validation (src/util.rs:721-795): (avg of the lowest 99 %, median)
inline std::pair<double, double> validation(size_t /*cam_idx*/, const GenericModel& final_result, const std::map<size_t, RvecTvec>& rtvec_list,
                                            const std::vector<std::optional<FrameFeature>>& detected_feature_frames, int device = 0) {
    std::set<size_t> valid;
    for (auto& kv : rtvec_list) if (kv.first < detected_feature_frames.size() && detected_feature_frames[kv.first]) valid.insert(kv.first);
    detail::Ctx ctx(device); detail::Prob p;
    const auto f = detail::flatten({&detected_feature_frames}, {valid});
    if (detail::make_problem(ctx, f, {final_result}, false, p) != CCAL_OK) throw std::runtime_error("problem creation failed");
    std::vector<double> poses;
    for (size_t fi : f.slots) { const auto v = rtvec_list.at(fi).as6(); poses.insert(poses.end(), v.begin(), v.end()); }
    const auto intr = detail::intr_matrix({final_result});
    double avg = 0, med = 0;
    if (ccal_validation(p.h, 0, intr.data(), poses.data(), nullptr, &avg, &med) != CCAL_OK) throw std::runtime_error("validation failed");
    return {avg, med};
}

// factors::ReprojectionFactor (src/optimization/factors.rs:126-173): residual_func(params) with
// params = [intrinsics (P_eff), rvec, tvec]; `J` (2 x D, row-major) optionally receives what tiny-solver gets with duals.
class ReprojectionFactor {
public:
    ReprojectionFactor(const GenericModel& target, std::array<float, 3> p3d, std::array<float, 2> p2d, bool xy_same_focal)
        : target_(target), p3d_(p3d), p2d_(p2d), xy_(xy_same_focal) {}
    std::array<double, 2> residual_func(const std::vector<std::vector<double>>& params, std::vector<double>* J = nullptr) const { return eval(params, false, J); }
protected:
    std::array<double, 2> eval(const std::vector<std::vector<double>>& params, bool other, std::vector<double>* J) const {
        const int n = other ? 2 : 1;
        std::vector<double> full = params[0];
        if (xy_) full.insert(full.begin() + 1, full[0]);                           // factors.rs:155-158
        std::vector<GenericModel> cams(n, GenericModel(target_.model_id(), full, target_.width(), target_.height()));
        detail::Ctx ctx; detail::Prob p; detail::Flat f;
        f.slots = {0}; f.obs_cam = {n - 1}; f.obs_slot = {0}; f.offs = {0, 1};
        f.x = {p3d_[0]}; f.y = {p3d_[1]}; f.z = {p3d_[2]}; f.u = {p2d_[0]}; f.v = {p2d_[1]};
        if (detail::make_problem(ctx, f, cams, xy_, p) != CCAL_OK) throw std::runtime_error("problem creation failed");
        const auto intr = detail::intr_matrix(cams);
        double pose[6], extr[12] = {0};
        for (int i = 0; i < 3; ++i) { pose[i] = params[1][i]; pose[3 + i] = params[2][i]; if (other) { extr[6 + i] = params[3][i]; extr[9 + i] = params[4][i]; } }
        std::array<double, 2> r{};
        std::vector<double> Jl((size_t)ccal_jacobian_len(p.h));
        if (ccal_eval(p.h, intr.data(), pose, extr, 0, r.data(), Jl.data()) != CCAL_OK) throw std::runtime_error(ccal_last_error(ctx.h));
        if (J) *J = Jl;
        return r;
    }
    GenericModel target_; std::array<float, 3> p3d_; std::array<float, 2> p2d_; bool xy_;
};
// factors::OtherCamReprojectionFactor (factors.rs:179-228): params = [intrinsics, rvec_0_b, tvec_0_b, rvec_i_0, tvec_i_0]
class OtherCamReprojectionFactor : public ReprojectionFactor {
public:
    using ReprojectionFactor::ReprojectionFactor;
    std::array<double, 2> residual_func(const std::vector<std::vector<double>>& params, std::vector<double>* J = nullptr) const { return eval(params, true, J); }
};


// ---------------------------------------------------------------------------------------------------------------
// Wire formats: what `ccrs` writes next to a calibration, so that results can be diffed against a
// run of the reference made elsewhere and fed back in as initial values.
//   cam{i}.json        {"EUCM": {"fx":..,"fy":..,"cx":..,"cy":..,"alpha":..,"beta":..,"width":..,"height":..}}   data/eucm.json:1-11
//   cam{i}_poses.json  {"<frame index>": {"rvec":[3],"tvec":[3]}, ...}      src/types.rs:13-17, src/bin/camera_calibration.rs:288-293
//   extrinsics.json    {"rtvecs":[{"rvec":[3],"tvec":[3]}, ...]}            src/types.rs:41-44
//   report.txt         src/io.rs:21-31
// serde tags of the model variants: UCM / EUCM / KannalaBrandt4 appear in the reference (data/eucm.json,
// examples/convert_model.rs:14,19); "OpenCVModel5" is this build's assumption about the absent crate (one string).
// ---------------------------------------------------------------------------------------------------------------
namespace json {

struct Value {                               // a JSON value: just enough for the three shapes above
    enum Kind { Null, Number, String, Array, Object } kind = Null;
    double num = 0.0;
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;       // insertion order kept
    const Value& at(const std::string& k) const {
        for (auto& kv : obj) if (kv.first == k) return kv.second;
        throw std::runtime_error("json: missing key " + k);
    }
};

class Parser {
public:
    explicit Parser(const std::string& s) : s_(s) {}
    Value parse() { Value v = value(); ws(); if (i_ != s_.size()) fail("trailing characters"); return v; }
private:
    const std::string& s_; size_t i_ = 0;
    [[noreturn]] void fail(const char* m) const { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(i_)); }
    void ws() { while (i_ < s_.size() && (s_[i_] == ' ' || s_[i_] == '\n' || s_[i_] == '\t' || s_[i_] == '\r')) ++i_; }
    Value value() {
        ws();
        if (i_ >= s_.size()) fail("unexpected end");
        const char c = s_[i_];
        Value v;
        if (c == '{') {
            v.kind = Value::Object; ++i_; ws();
            if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
            for (;;) {
                ws(); if (i_ >= s_.size() || s_[i_] != '"') fail("expected a key");
                std::string k = string();
                ws(); if (i_ >= s_.size() || s_[i_] != ':') fail("expected ':'");
                ++i_;
                v.obj.emplace_back(std::move(k), value());
                ws(); if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
                if (i_ < s_.size() && s_[i_] == '}') { ++i_; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = Value::Array; ++i_; ws();
            if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws(); if (i_ < s_.size() && s_[i_] == ',') { ++i_; continue; }
                if (i_ < s_.size() && s_[i_] == ']') { ++i_; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = Value::String; v.str = string(); return v; }
        if (s_.compare(i_, 4, "null") == 0) { i_ += 4; return v; }
        char* end = nullptr;
        v.num = std::strtod(s_.c_str() + i_, &end);
        if (end == s_.c_str() + i_) fail("expected a value");
        v.kind = Value::Number; i_ = (size_t)(end - s_.c_str());
        return v;
    }
    std::string string() {
        std::string o; ++i_;
        while (i_ < s_.size() && s_[i_] != '"') {
            if (s_[i_] == '\\' && i_ + 1 < s_.size()) { ++i_; const char e = s_[i_]; o.push_back(e == 'n' ? '\n' : e == 't' ? '\t' : e); }
            else o.push_back(s_[i_]);
            ++i_;
        }
        if (i_ >= s_.size()) fail("unterminated string");
        ++i_;
        return o;
    }
};
inline Value parse(const std::string& text) { return Parser(text).parse(); }
inline std::string number(double v) { char b[40]; std::snprintf(b, sizeof b, "%.17g", v); return b; }    // round-trips an f64
inline std::string vec3(const std::array<double, 3>& v) { return "[" + number(v[0]) + "," + number(v[1]) + "," + number(v[2]) + "]"; }
inline std::string read_file(const std::string& path) {
    std::ifstream f(path); if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss; ss << f.rdbuf(); return ss.str();
}
inline void write_file(const std::string& path, const std::string& text) {
    std::ofstream f(path); if (!f) throw std::runtime_error("cannot write " + path);
    f << text;
}

}  // namespace json

namespace detail {
struct ModelJson { int model; const char* tag; std::vector<const char*> keys; };
inline const std::vector<ModelJson>& model_json_table() {
    static const std::vector<ModelJson> t = {
        {CCAL_MODEL_UCM, "UCM", {"fx", "fy", "cx", "cy", "alpha"}},
        {CCAL_MODEL_EUCM, "EUCM", {"fx", "fy", "cx", "cy", "alpha", "beta"}},
        {CCAL_MODEL_KB4, "KannalaBrandt4", {"fx", "fy", "cx", "cy", "k1", "k2", "k3", "k4"}},
        {CCAL_MODEL_OPENCV5, "OpenCVModel5", {"fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "k3"}},
    };
    return t;
}
inline RvecTvec rtvec_from(const json::Value& v) {
    RvecTvec r;
    const auto& a = v.at("rvec").arr; const auto& b = v.at("tvec").arr;
    if (a.size() != 3 || b.size() != 3) throw std::runtime_error("json: rvec / tvec must have three entries");
    for (int i = 0; i < 3; ++i) { r.rvec[i] = a[i].num; r.tvec[i] = b[i].num; }
    return r;
}
inline std::string rtvec_to(const RvecTvec& r) { return "{\"rvec\":" + json::vec3(r.rvec) + ",\"tvec\":" + json::vec3(r.tvec) + "}"; }
}  // namespace detail

inline std::string model_to_json(const GenericModel& m) {
    for (auto& e : detail::model_json_table()) if (e.model == m.model_id()) {
        std::string s = std::string("{\n  \"") + e.tag + "\": {\n";
        for (size_t i = 0; i < e.keys.size(); ++i) s += std::string("    \"") + e.keys[i] + "\": " + json::number(m.params()[i]) + ",\n";
        s += "    \"width\": " + std::to_string((long)std::lround(m.width())) + ",\n    \"height\": " + std::to_string((long)std::lround(m.height())) + "\n  }\n}";
        return s;
    }
    throw std::invalid_argument("model_to_json: this model's JSON field names are defined only in the absent camera-intrinsic-model crate");
}
inline GenericModel model_from_json(const std::string& text) {
    const json::Value v = json::parse(text);
    if (v.kind != json::Value::Object || v.obj.size() != 1) throw std::runtime_error("json: expected {\"<Model>\": {...}}");
    for (auto& e : detail::model_json_table()) if (v.obj[0].first == e.tag) {
        std::vector<double> p;
        for (auto k : e.keys) p.push_back(v.obj[0].second.at(k).num);
        return GenericModel(e.model, p, v.obj[0].second.at("width").num, v.obj[0].second.at("height").num);
    }
    throw std::runtime_error("json: unknown camera model " + v.obj[0].first);
}
inline std::string poses_to_json(const std::map<size_t, RvecTvec>& poses) {            // BTreeMap order = std::map order
    std::string s = "{";
    bool first = true;
    for (auto& kv : poses) { s += (first ? "\n  \"" : ",\n  \"") + std::to_string(kv.first) + "\": " + detail::rtvec_to(kv.second); first = false; }
    return s + "\n}";
}
inline std::map<size_t, RvecTvec> poses_from_json(const std::string& text) {
    std::map<size_t, RvecTvec> out;
    const json::Value doc = json::parse(text);
    for (auto& kv : doc.obj) out[(size_t)std::stoull(kv.first)] = detail::rtvec_from(kv.second);
    return out;
}
inline std::string extrinsics_to_json(const std::vector<RvecTvec>& rtvecs) {
    std::string s = "{\n  \"rtvecs\": [";
    for (size_t i = 0; i < rtvecs.size(); ++i) s += (i ? ",\n    " : "\n    ") + detail::rtvec_to(rtvecs[i]);
    return s + "\n  ]\n}";
}
inline std::vector<RvecTvec> extrinsics_from_json(const std::string& text) {
    std::vector<RvecTvec> out;
    const json::Value doc = json::parse(text);
    for (auto& v : doc.at("rtvecs").arr) out.push_back(detail::rtvec_from(v));
    return out;
}
// write_report (src/io.rs:21-31), byte for byte
inline std::string report_text(bool with_extrinsic, const std::vector<std::pair<double, double>>& rep_rms) {
    std::string s = std::string("Calibrate with extrinsics: ") + (with_extrinsic ? "true" : "false") + "\n\n";
    char b[160];
    for (size_t i = 0; i < rep_rms.size(); ++i) {
        std::snprintf(b, sizeof b, "cam%zu:\n    average reprojection error: %.5f px\n    median  reprojection error: %.5f px\n\n", i, rep_rms[i].first, rep_rms[i].second);
        s += b;
    }
    return s;
}

}  // namespace ccal
