"""CPU: the ragged-frame plan of the Gram launch (host code of libccal_hip.so, csrc/ccal_kernels_gram2.hip: gram2_bin_plan) through a
plain C++ program (tests/cpp/test_bin_plan.cpp): every frame once, bins contiguous over the frames sorted by corner count, each
frame covered by its bin's lanes within the trip-count limit, workgroup ranges consistent; uniform frames (the headline workload)
and problems below 2 000 frames are left alone.  The launch itself is held against the oracle on the GPU
(tests/test_gpu_configs.py::test_ragged_*)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "camera_intrinsic_calibration_rs_amd", "lib")


def test_gram_bin_plan_invariants(tmp_path):
    exe = str(tmp_path / "test_bin_plan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", os.path.join(ROOT, "tests", "cpp", "test_bin_plan.cpp"), "-o", exe,
                           "-L", LIBDIR, "-lccal_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe], env=dict(os.environ, LD_LIBRARY_PATH=LIBDIR + ":/opt/rocm/lib")).decode()
    assert "PLAN-OK" in out, out
    # the plan for the bench's ragged workload shape: the solve runs single-launch groups of 12-lane wavefronts there - ONE bin, the table
    # folded so that every SIMD pairs a long wavefront with a short one; a problem beyond two wavefronts per SIMD: equalised bins
    assert "10000 frames U{24..144}, two wavefronts per SIMD: folded at 5000, 12 lanes x 10000 frames" in out, out
    assert "50000 frames U{24..144}: 3 bins, T = " in out, out
