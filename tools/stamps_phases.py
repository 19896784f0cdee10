"""Developer tool: prologue / corner loop / epilogue times of k_gram1w's wavefronts (10 000 frames, EUCM) from a library
built with -DCCAL_STAMPS (tools/build_variants.sh "stamps:-DCCAL_STAMPS"; CCAL_LIB selects it)."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem
sp = synth.make_problem(10000, "eucm")
ctx = Context(0); p = Problem.from_synth(ctx, sp)
p.upload_params(sp.intr0, sp.poses0, sp.extr0)
for _ in range(30): p.build_normal_dev(0.0)
torch.cuda.synchronize()
lib = _ffi.load()
n = 8192 + 4096
buf = np.zeros(n, dtype=np.float64)
lib.ccal_debug_fcbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lib.ccal_debug_fcbuf(p.handle, buf.ctypes.data_as(C.c_void_p), n)
st = buf[0:4000:2]; en = buf[1:4000:2]; tl = buf[8192:8192+4000:2]; ta = buf[8193:8192+4000:2]
for name, sel in (("older (first 1000 waves)", slice(0, 1000)), ("younger (last 500 waves)", slice(1500, 2000))):
    print(name, "prologue", np.median(tl[sel] - st[sel]) / 100, "loop", np.median(ta[sel] - tl[sel]) / 100, "epilogue", np.median(en[sel] - ta[sel]) / 100, "total", np.median(en[sel] - st[sel]) / 100)
