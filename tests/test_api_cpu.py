"""CPU: host-side pieces of the API mirror that need no GPU (wire formats, pose algebra)."""
import json

import numpy as np

from camera_intrinsic_calibration_rs_amd import api


def test_model_json_matches_reference_sample(tmp_path):
    """data/eucm.json of the reference: {"EUCM": {fx, fy, cx, cy, alpha, beta, width, height}}."""
    ref = {"EUCM": {"fx": 190.89618687183938, "fy": 190.87022285882367, "cx": 254.9375370481962,
                    "cy": 256.86414483060787, "alpha": 0.6283550447635853, "beta": 1.0458678747533083,
                    "width": 512, "height": 512}}
    m = api.GenericModel.from_json_obj(ref)
    assert m.kind == "eucm" and m.width() == 512 and m.params()[5] == ref["EUCM"]["beta"]
    p = tmp_path / "cam0.json"
    api.model_to_json(str(p), m)
    assert json.load(open(p)) == ref
    assert np.array_equal(api.model_from_json(str(p)).params(), m.params())


def test_pose_and_extrinsics_json_roundtrip(tmp_path):
    poses = {7: api.RvecTvec((0.1, 0.2, 0.3), (1.0, 2.0, 3.0)), 2: api.RvecTvec((0.0, 0.0, 0.1), (0.0, 0.0, 1.0))}
    p = tmp_path / "cam0_poses.json"
    api.poses_to_json(str(p), poses)
    raw = json.load(open(p))
    assert list(raw.keys()) == ["2", "7"] and raw["7"] == {"rvec": [0.1, 0.2, 0.3], "tvec": [1.0, 2.0, 3.0]}
    assert api.poses_from_json(str(p)) == poses
    e = tmp_path / "extrinsics.json"
    api.extrinsics_to_json(str(e), list(poses.values()))
    assert list(json.load(open(e)).keys()) == ["rtvecs"]
    assert api.extrinsics_from_json(str(e)) == list(poses.values())


def test_write_report_format(tmp_path):
    """src/io.rs:21-31."""
    p = tmp_path / "report.txt"
    api.write_report(str(p), True, [(0.123456789, 0.1), (0.2, 0.098765)])
    assert open(p).read() == ("Calibrate with extrinsics: true\n\n"
                              "cam0:\n    average reprojection error: 0.12346 px\n    median  reprojection error: 0.10000 px\n\n"
                              "cam1:\n    average reprojection error: 0.20000 px\n    median  reprojection error: 0.09877 px\n\n")


def test_rvec_tvec_algebra_matches_oracle(oracle):
    """tests/types_test.rs:5-20 round trip + compose/inverse against the oracle's nalgebra restatement."""
    a = api.RvecTvec((0.1, 0.2, 0.3), (1.0, 2.0, 3.0)); b = api.RvecTvec((-0.4, 0.05, 0.2), (0.3, -0.1, 0.9))
    np.testing.assert_allclose(a.compose(b).as6(), oracle.pose_compose(a.as6(), b.as6()), atol=1e-13)
    np.testing.assert_allclose(a.inverse().as6(), oracle.pose_inverse(a.as6()), atol=1e-13)
    np.testing.assert_allclose(a.inverse().inverse().as6(), a.as6(), atol=1e-13)
