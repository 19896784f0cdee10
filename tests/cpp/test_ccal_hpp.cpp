// C++ mirror of the reference API (include/ccal.hpp) exercised like the reference's own tests.
// Built and run by tests/test_gpu_cpp_api.py on the GPU box; prints one JSON object.
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "ccal.hpp"

using namespace ccal;

// tests/optimization_test.rs:36-80, statement for statement
static void test_reprojection_factor() {
    const double w = 640, h = 480;
    const std::vector<double> cam_params = {500.0, 500.0, 320.0, 240.0, 0.5};
    const GenericModel model(CCAL_MODEL_UCM, cam_params, w, h);
    const std::array<float, 3> p3d = {1.0f, 2.0f, 10.0f};
    // project using the model: the residual against p2d = 0 is the projection
    const auto proj = ReprojectionFactor(model, p3d, {0.0f, 0.0f}, false).residual_func({cam_params, {0, 0, 0}, {0, 0, 0}});
    const std::array<float, 2> p2d = {(float)proj[0], (float)proj[1]};
    const ReprojectionFactor factor(model, p3d, p2d, false);
    const auto residual = factor.residual_func({cam_params, {0, 0, 0}, {0, 0, 0}});
    assert(std::hypot(residual[0], residual[1]) < 1e-4 && "Residual should be zero at GT");
    const auto residual_bad = factor.residual_func({cam_params, {0, 0, 0}, {0.1, 0, 0}});
    assert(std::hypot(residual_bad[0], residual_bad[1]) > 1e-3 && "Residual should be non-zero for bad params");
    std::vector<double> J;
    factor.residual_func({cam_params, {0.01, 0.02, 0.03}, {0.1, 0, 0}}, &J);
    assert(J.size() == 2 * 11);
}

// tests/types_test.rs:5-20
static void test_rvec_tvec_conversion() {
    RvecTvec rt; rt.rvec = {0.1, 0.2, 0.3}; rt.tvec = {1.0, 2.0, 3.0};
    const RvecTvec back = RvecTvec::from_quat(rt.quat(), rt.tvec);
    for (int i = 0; i < 3; ++i) { assert(std::fabs(back.rvec[i] - rt.rvec[i]) < 1e-6); assert(std::fabs(back.tvec[i] - rt.tvec[i]) < 1e-6); }
    const RvecTvec id = rt.compose(rt.inverse());
    for (int i = 0; i < 3; ++i) { assert(std::fabs(id.rvec[i]) < 1e-12); assert(std::fabs(id.tvec[i]) < 1e-12); }
}

template <class T> static T rd(std::ifstream& f) { T v; f.read(reinterpret_cast<char*>(&v), sizeof v); return v; }

static std::vector<std::optional<FrameFeature>> read_frames(std::ifstream& f) {
    const int n_frames = rd<int32_t>(f);
    std::vector<std::optional<FrameFeature>> frames(n_frames);
    for (int i = 0; i < n_frames; ++i) {
        const int present = rd<int32_t>(f), nf = rd<int32_t>(f);
        if (!present) continue;
        FrameFeature ff; ff.img_w_h = {512, 512};
        for (int k = 0; k < nf; ++k) {
            const uint32_t id = rd<uint32_t>(f);
            FeaturePoint fp; fp.p2d = {rd<float>(f), rd<float>(f)}; fp.p3d = {rd<float>(f), rd<float>(f), rd<float>(f)};
            ff.features[id] = fp;
        }
        frames[i] = ff;
    }
    return frames;
}

// The reference's two-camera flow (src/bin/camera_calibration.rs:262-320): per-camera calib_camera, init_camera_extrinsic,
// calib_all_camera_with_extrinsics.  Fixture: "RIG2", then per camera its frames, model id, parameters, width, height.
static int run_rig(std::ifstream& f) {
    std::vector<std::vector<std::optional<FrameFeature>>> frames;
    std::vector<GenericModel> cams;
    for (int c = 0; c < 2; ++c) {
        frames.push_back(read_frames(f));
        const int model_id = rd<int32_t>(f), P = rd<int32_t>(f);
        std::vector<double> params(P); for (auto& v : params) v = rd<double>(f);
        const double w = rd<double>(f), h = rd<double>(f);
        cams.emplace_back(model_id, params, w, h);
    }
    std::vector<GenericModel> solo;
    std::vector<std::map<size_t, RvecTvec>> rtvecs;
    for (int c = 0; c < 2; ++c) {
        const auto r = calib_camera(frames[c], cams[c], false, 0, false);
        if (!r) { std::printf("{\"result\": null}\n"); return 0; }
        solo.push_back(r->first); rtvecs.push_back(r->second);
    }
    const auto t_i_0 = init_camera_extrinsic(rtvecs);
    const auto all = calib_all_camera_with_extrinsics(solo, t_i_0, rtvecs, frames, false, 0, false);
    if (!all) { std::printf("{\"result\": null}\n"); return 0; }
    std::printf("{\"params\": [");
    for (int c = 0; c < 2; ++c) {
        std::printf("%s[", c ? ", " : "");
        for (size_t i = 0; i < all->intrinsics[c].params().size(); ++i) std::printf("%s%.17g", i ? ", " : "", all->intrinsics[c].params()[i]);
        std::printf("]");
    }
    const auto e = all->t_i_0[1].as6();
    std::printf("], \"t_1_0\": [");
    for (int i = 0; i < 6; ++i) std::printf("%s%.17g", i ? ", " : "", e[i]);
    std::printf("], \"n_board_poses\": %zu}\n", all->board_poses.size());
    return 0;
}

// util::init_ucm (src/util.rs:287-378) on the fixture's first two frames: "UCM0", frames, then init_f, init_alpha,
// fixed_focal and the two starting poses.
static int run_init_ucm(std::ifstream& f) {
    const auto frames = read_frames(f);
    const double init_f = rd<double>(f), init_alpha = rd<double>(f);
    const int fixed_focal = rd<int32_t>(f);
    double p[12]; for (auto& v : p) v = rd<double>(f);
    const auto m = init_ucm(*frames[0], *frames[1], RvecTvec::from6(p), RvecTvec::from6(p + 6), init_f, init_alpha, fixed_focal != 0);
    if (!m) { std::printf("{\"result\": null}\n"); return 0; }
    std::printf("{\"model\": %d, \"params\": [", m->model_id());
    for (size_t i = 0; i < m->params().size(); ++i) std::printf("%s%.17g", i ? ", " : "", m->params()[i]);
    std::printf("]}\n");
    return 0;
}

int main(int argc, char** argv) {
    test_reprojection_factor();
    test_rvec_tvec_conversion();
    if (argc < 2) { std::printf("{\"unit_tests\": \"ok\"}\n"); return 0; }
    std::ifstream f(argv[1], std::ios::binary);
    if (argc > 2 && std::strcmp(argv[2], "rig") == 0) return run_rig(f);
    if (argc > 2 && std::strcmp(argv[2], "init_ucm") == 0) return run_init_ucm(f);
    const auto frames = read_frames(f);
    const int model_id = rd<int32_t>(f), P = rd<int32_t>(f);
    std::vector<double> params(P); for (auto& v : params) v = rd<double>(f);
    const double w = rd<double>(f), h = rd<double>(f);
    const int one_focal = rd<int32_t>(f), disabled = rd<int32_t>(f), fixed_focal = rd<int32_t>(f);
    const GenericModel cam(model_id, params, w, h);
    // CCAL_TEST_DEVICES="0,0": the same ONE call over several shards (ccal::Devices -> ccal_multi_*); default: device 0
    Devices devs(0);
    if (const char* e = std::getenv("CCAL_TEST_DEVICES")) {
        devs.ids.clear();
        for (const char* q = e; *q;) { devs.ids.push_back(std::atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    }
    const auto res = calib_camera(frames, cam, one_focal != 0, (size_t)disabled, fixed_focal != 0, nullptr, devs);     // poses initialised inside
    if (!res) { std::printf("{\"result\": null}\n"); return 0; }
    const auto val = validation(0, res->first, res->second, frames);
    std::printf("{\"params\": [");
    for (size_t i = 0; i < res->first.params().size(); ++i) std::printf("%s%.17g", i ? ", " : "", res->first.params()[i]);
    std::printf("], \"n_poses\": %zu, \"avg99\": %.17g, \"median\": %.17g, \"pose0\": [", res->second.size(), val.first, val.second);
    const auto p0 = res->second.begin()->second.as6();
    for (int i = 0; i < 6; ++i) std::printf("%s%.17g", i ? ", " : "", p0[i]);
    // convert_model: the calibrated model refitted as KB4 over the reference's pixel grid
    GenericModel kb4(2, std::vector<double>(8, 0.0), w, h);
    convert_model(res->first, kb4, 0);
    std::printf("], \"kb4\": [");
    for (size_t i = 0; i < kb4.params().size(); ++i) std::printf("%s%.17g", i ? ", " : "", kb4.params()[i]);
    std::printf("]}\n");
    return 0;
}
