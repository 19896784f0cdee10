#!/usr/bin/env python3
"""Developer tool: per-phase wave cycles of k_gram1 from a CCAL_LIB=...libccal_stamp.so diagnostic build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
ctx = Context(0); sp = synth.make_problem(10000, "eucm"); prob = Problem.from_synth(ctx, sp)
prob.solve(sp.intr0, sp.poses0, opts=default_opts(0))
lib = _ffi.load(); buf = np.zeros(10000 * 40)
lib.ccal_debug_fcbuf.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64]
assert lib.ccal_debug_fcbuf(prob.handle, buf.ctypes.data_as(C.POINTER(C.c_double)), buf.size) == 0
d = buf.reshape(-1, 40)[:, :5]
names = ["setup(backsub+exp-map)", "J rows (loads+VALU)", "LDS stage+sync", "MFMA loop", "whole wave"]
for n, col in zip(names, d.T): print(f"{n:26s} mean {col.mean():9.0f}  median {np.median(col):9.0f}  max {col.max():9.0f} cycles")
