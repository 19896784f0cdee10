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
