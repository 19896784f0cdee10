"""The C++ mirror of the reference API (include/ccal.hpp): syntax check on CPU, build + run on the GPU."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_ccal_hpp.cpp")
LIBDIR = os.path.join(ROOT, "camera_intrinsic_calibration_rs_amd", "lib")


def test_header_compiles_as_plain_cxx17():
    """ccal.hpp needs nothing but the C ABI header and the standard library (no HIP, no torch)."""
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), SRC])
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "ccal.h")])


def test_cpp_wire_formats_roundtrip_with_python_mirror(tmp_path):
    """cam{i}.json / cam{i}_poses.json / extrinsics.json / report.txt: files written by the Python mirror are read and
    re-written by include/ccal.hpp; both mirrors read each other's output to the last bit; the reference's sample model
    file (data/eucm.json, a fixture under tests/golden/) parses.  Host code only: runs without a GPU."""
    from camera_intrinsic_calibration_rs_amd import api, synth
    exe = str(tmp_path / "test_ccal_json")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_ccal_json.cpp"), "-o", exe,
                           "-L", LIBDIR, "-lccal_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    d = tmp_path
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")))
    with open(d / "eucm_reference.json", "w") as f:
        json.dump(golden["data_eucm_json"], f, indent=2)
    rng = np.random.default_rng(2)
    models = {}
    for name in ("ucm", "eucm", "kb4", "opencv5"):
        P = synth.MODEL_NPARAMS[synth.MODEL_NAMES[name]]
        models[name] = api.GenericModel(name, np.asarray(synth.GT_PARAMS[synth.MODEL_NAMES[name]]) * (1 + rng.normal(0, 1e-3, P)), 512, 512)
        api.model_to_json(str(d / f"cam_{name}.json"), models[name])
    poses = {int(k): api.RvecTvec.from6(rng.normal(0, 1, 6)) for k in (0, 3, 17, 250)}
    api.poses_to_json(str(d / "cam0_poses.json"), poses)
    ext = [api.RvecTvec.from6(np.zeros(6)), api.RvecTvec.from6(rng.normal(0, 0.1, 6))]
    api.extrinsics_to_json(str(d / "extrinsics.json"), ext)
    out = subprocess.check_output([exe, str(d)], env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib")).decode()
    assert "JSON-OK 4 2" in out
    for name, m in models.items():
        back = api.model_from_json(str(d / f"cpp_cam_{name}.json"))
        assert back.kind == name and back.width() == 512 and back.height() == 512
        np.testing.assert_array_equal(back.params(), m.params())
    back = api.poses_from_json(str(d / "cpp_cam0_poses.json"))
    assert sorted(back) == sorted(poses)
    for k in poses:
        np.testing.assert_array_equal(back[k].as6(), poses[k].as6())
    back = api.extrinsics_from_json(str(d / "cpp_extrinsics.json"))
    for a, b in zip(back, ext):
        np.testing.assert_array_equal(a.as6(), b.as6())
    api.write_report(str(d / "py_report.txt"), True, [(0.0912345678, 0.0801), (0.1, 0.123456789)])
    assert open(d / "py_report.txt").read() == open(d / "cpp_report.txt").read()


@pytest.mark.gpu
def test_cpp_calib_camera_matches_python_binding(tmp_path, gpu_ctx):
    from camera_intrinsic_calibration_rs_amd import api, synth
    exe = str(tmp_path / "test_ccal_hpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lccal_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    sp = synth.make_problem(18, "eucm", xy_same_focal=True)
    frames = api.frames_from_synth(sp)
    frames[4] = None
    fix = tmp_path / "frames.bin"
    with open(fix, "wb") as f:
        f.write(struct.pack("<i", len(frames)))
        for fr in frames:
            if fr is None:
                f.write(struct.pack("<ii", 0, 0)); continue
            f.write(struct.pack("<ii", 1, len(fr.features)))
            for k in sorted(fr.features):
                fp = fr.features[k]
                f.write(struct.pack("<Ifffff", k, fp.p2d[0], fp.p2d[1], fp.p3d[0], fp.p3d[1], fp.p3d[2]))
        f.write(struct.pack("<ii", 1, 6)); f.write(struct.pack("<6d", *sp.intr0[0, :6])); f.write(struct.pack("<dd", 512.0, 512.0))
        f.write(struct.pack("<iii", 1, 0, 0))
    out = subprocess.check_output([exe, str(fix)], env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib")).decode()
    got = json.loads(out.strip().splitlines()[-1])
    cam0 = api.GenericModel("eucm", sp.intr0[0, :6], 512, 512)
    model, poses = api.calib_camera(frames, cam0, True, 0, False, None, ctx=gpu_ctx)
    a, m = api.validation(0, model, poses, frames, ctx=gpu_ctx)
    np.testing.assert_allclose(got["params"], model.params(), rtol=1e-12)
    assert got["n_poses"] == len(poses) == 17
    assert abs(got["avg99"] - a) < 1e-12 and abs(got["median"] - m) < 1e-12
    np.testing.assert_allclose(got["pose0"], poses[min(poses)].as6(), atol=1e-12)
    kb4 = api.convert_model(model, api.GenericModel("kb4", [0.0] * 8, 512, 512), 0, ctx=gpu_ctx)
    np.testing.assert_allclose(got["kb4"], kb4.params(), rtol=1e-12, atol=1e-14)
    # the same ONE call over three shards of device 0 (ccal::Devices{0, 0, 0} -> ccal_multi_*: the library shards the frames,
    # one all-reduce per step): the reference's single-process form of a multi-GPU solve
    out3 = subprocess.check_output([exe, str(fix)], env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib", CCAL_TEST_DEVICES="0,0,0"),
                                   timeout=300).decode()
    got3 = json.loads(out3.strip().splitlines()[-1])
    np.testing.assert_allclose(got3["params"], got["params"], rtol=1e-9)
    assert got3["n_poses"] == got["n_poses"]
    np.testing.assert_allclose(got3["pose0"], got["pose0"], atol=1e-9)


def _write_frames(f, frames):
    f.write(struct.pack("<i", len(frames)))
    for fr in frames:
        if fr is None:
            f.write(struct.pack("<ii", 0, 0)); continue
        f.write(struct.pack("<ii", 1, len(fr.features)))
        for k in sorted(fr.features):
            fp = fr.features[k]
            f.write(struct.pack("<Ifffff", k, fp.p2d[0], fp.p2d[1], fp.p3d[0], fp.p3d[1], fp.p3d[2]))


@pytest.mark.gpu
def test_cpp_two_camera_flow_matches_python_binding(tmp_path, gpu_ctx):
    """ccal::calib_camera x 2 -> init_camera_extrinsic -> calib_all_camera_with_extrinsics from C++ against the same
    flow through the Python mirror (both drive the C ABI; frames in the same corner order)."""
    from camera_intrinsic_calibration_rs_amd import api, synth
    exe = str(tmp_path / "test_ccal_hpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lccal_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    sp = synth.make_problem(16, "eucm", n_cams=2)
    frames = [api.frames_from_synth(sp, c) for c in range(2)]
    fix = tmp_path / "rig.bin"
    with open(fix, "wb") as f:
        for c in range(2):
            _write_frames(f, frames[c])
            f.write(struct.pack("<ii", 1, 6)); f.write(struct.pack("<6d", *sp.intr0[c, :6])); f.write(struct.pack("<dd", 512.0, 512.0))
    out = subprocess.check_output([exe, str(fix), "rig"], env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib")).decode()
    got = json.loads(out.strip().splitlines()[-1])
    cams0 = [api.GenericModel("eucm", sp.intr0[c, :6], 512, 512) for c in range(2)]
    per_cam = [api.calib_camera(frames[c], cams0[c], False, 0, False, None, ctx=gpu_ctx) for c in range(2)]
    t_i_0 = api.init_camera_extrinsic([r[1] for r in per_cam])
    models, t_out, board = api.calib_all_camera_with_extrinsics([r[0] for r in per_cam], t_i_0, [r[1] for r in per_cam], frames,
                                                               False, 0, False, ctx=gpu_ctx)
    for c in range(2):
        np.testing.assert_allclose(got["params"][c], models[c].params(), rtol=1e-10)
    np.testing.assert_allclose(got["t_1_0"], t_out[1].as6(), rtol=0, atol=1e-10)
    assert got["n_board_poses"] == len(board)
    assert np.abs(np.array(got["t_1_0"][3:]) - sp.extr_gt[1, 3:]).max() < 5e-4


@pytest.mark.gpu
def test_cpp_init_ucm_matches_python_binding(tmp_path, gpu_ctx):
    """ccal::init_ucm (src/util.rs:287-378) from C++ against api.init_ucm: the same two solves through the C ABI."""
    from camera_intrinsic_calibration_rs_amd import api, synth
    exe = str(tmp_path / "test_ccal_hpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", LIBDIR, "-lccal_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    sp = synth.make_problem(2, "ucm", seed=77)
    frames = api.frames_from_synth(sp)
    gt = sp.intr_gt[0]
    fix = tmp_path / "ucm.bin"
    with open(fix, "wb") as f:
        _write_frames(f, frames)
        f.write(struct.pack("<ddi", gt[0] * 1.3, 0.4, 0))
        f.write(struct.pack("<12d", *sp.poses0[0], *sp.poses0[1]))
    out = subprocess.check_output([exe, str(fix), "init_ucm"], env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib")).decode()
    got = json.loads(out.strip().splitlines()[-1])
    m = api.init_ucm(frames[0], frames[1], api.RvecTvec.from6(sp.poses0[0]), api.RvecTvec.from6(sp.poses0[1]),
                     init_f=gt[0] * 1.3, init_alpha=0.4, fixed_focal=False, ctx=gpu_ctx)
    assert got["model"] == 0 and m is not None
    np.testing.assert_allclose(got["params"], m.params(), rtol=1e-12)
    assert got["params"][0] == got["params"][1] and abs(got["params"][0] / gt[0] - 1) < 0.03
