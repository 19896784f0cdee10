"""Developer tool: where a wavefront of k_schurq spends its life (100 MHz clock at the phase boundaries; library built with
-DCCAL_STAMPS: tools/build_tu_variants.sh ccal_kernels_schurq "sqstamps:-DCCAL_STAMPS"; CCAL_LIB selects it).
    python tools/stamps_sq.py [frames]"""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sp = synth.make_problem(F, "eucm", n_cams=2)
ctx = Context(0); p = Problem.from_synth(ctx, sp)
p.upload_params(sp.intr0, sp.poses0, sp.extr0)
for _ in range(30): p.build_normal_dev(0.0)
torch.cuda.synchronize()
lib = _ffi.load()
n = 8 * 4096
buf = np.zeros(n, dtype=np.float64)
lib.ccal_debug_fcbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lib.ccal_debug_fcbuf(p.handle, buf.ctypes.data_as(C.c_void_p), n)
st = buf.reshape(-1, 8)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
names = ["records -> LDS", "products with E", "cholesky", "image: zero + direct terms", "Y + slot records", "Y^T Y", "16 images -> partial row"]
print(f"{F} slots x 2 cameras: {len(st)} wavefronts, start spread {(st[:, 0].max() - t0) / 100:.2f} us, last end {(st[:, 7].max() - t0) / 100:.2f} us")
d = np.diff(st, axis=1) / 100.0
print("  " + "  ".join(f"{nm} {np.median(d[:, i]):.2f}" for i, nm in enumerate(names)) + f"  total {np.median(st[:, 7] - st[:, 0]) / 100:.2f} us")
