"""MI355X-native reprojection residual + Jacobian engine.

The compute path is the HIP library behind include/ccal.h; this package is the ctypes binding
plus the host-side mirror of the reference's calib-frame API.  Importing it never touches oracle/.
"""
from . import synth  # noqa: F401
