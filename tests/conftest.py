import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def gpu_ctx():
    """A ccal context on cuda:0 / hip:0.  Fails loudly (no skip, no fallback) if the HIP library is missing."""
    from camera_intrinsic_calibration_rs_amd.engine import Context
    return Context(0)


@pytest.fixture(scope="session")
def dev_ctx():
    """A context of the SECOND build of the library (libccal_hip_legacy.so: -DCCAL_DEV_SWITCHES -DCCAL_LEGACY_KERNELS
    -DCCAL_TEST_HOOKS).  The product library reads no developer switch from the environment; tests that force a non-default
    implementation through one (CCAL_DISABLE_FUSED, CCAL_SCHURQ, CCAL_FUSE_ELIM, ...) run the same sources in this build."""
    from camera_intrinsic_calibration_rs_amd import _ffi
    from camera_intrinsic_calibration_rs_amd.engine import Context
    return Context(0, lib=_ffi.load_legacy())


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
