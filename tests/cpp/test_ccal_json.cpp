// Wire formats of include/ccal.hpp (cam{i}.json, cam{i}_poses.json, extrinsics.json, report.txt) against files written by
// the Python mirror: read each, write it back, print the report text.  Needs no GPU.  Driven by tests/test_cpp_api.py.
#include <cassert>
#include <iostream>

#include "ccal.hpp"

using namespace ccal;

static int run(int argc, char** argv);
int main(int argc, char** argv) {
    try { return run(argc, argv); } catch (const std::exception& e) { std::cerr << "exception: " << e.what() << "\n"; return 1; }
}
static int run(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: test_ccal_json <dir>\n"; return 2; }
    const std::string d = argv[1];
    // the reference's own sample model file (data/eucm.json:1-11), as data
    const GenericModel ref = model_from_json(json::read_file(d + "/eucm_reference.json"));
    assert(ref.model_id() == CCAL_MODEL_EUCM && ref.params().size() == 6);
    assert(ref.width() == 512 && ref.height() == 512);
    for (const char* name : {"ucm", "eucm", "kb4", "opencv5"}) {
        const GenericModel m = model_from_json(json::read_file(d + "/cam_" + name + ".json"));
        json::write_file(d + "/cpp_cam_" + name + ".json", model_to_json(m));
    }
    const auto poses = poses_from_json(json::read_file(d + "/cam0_poses.json"));
    json::write_file(d + "/cpp_cam0_poses.json", poses_to_json(poses));
    const auto ext = extrinsics_from_json(json::read_file(d + "/extrinsics.json"));
    json::write_file(d + "/cpp_extrinsics.json", extrinsics_to_json(ext));
    json::write_file(d + "/cpp_report.txt", report_text(true, {{0.0912345678, 0.0801}, {0.1, 0.123456789}}));
    // closed forms of convert_model (src/util.rs:229-243): no device involved
    GenericModel ucm(CCAL_MODEL_UCM, {500.0, 510.0, 320.0, 240.0, 0.55}, 640, 480);
    GenericModel eucm(CCAL_MODEL_EUCM, {0, 0, 0, 0, 0, 0}, 640, 480), eucmt(CCAL_MODEL_EUCMT, std::vector<double>(8, 0.0), 640, 480);
    convert_model(ucm, eucm, 0); convert_model(ucm, eucmt, 0);
    assert(eucm.params()[4] == 0.55 && eucm.params()[5] == 1.0);
    assert(eucmt.params()[5] == 1.0 && eucmt.params()[6] == 0.0 && eucmt.params()[7] == 0.0 && eucmt.params()[1] == 510.0);
    bool threw = false;
    try { (void)model_to_json(eucmt); } catch (const std::invalid_argument&) { threw = true; }
    assert(threw);
    try { (void)json::parse("{\"a\": [1, 2"); assert(false); } catch (const std::runtime_error&) {}
    std::cout << "JSON-OK " << poses.size() << " " << ext.size() << "\n";
    return 0;
}
