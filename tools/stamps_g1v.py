"""Developer tool: where a wavefront of k_gram1v spends its life at session size (625 frames, EUCM, one frame per wavefront), from a
library built with -DCCAL_STAMPS (tools/build_variants.sh "stamps:-DCCAL_STAMPS"; CCAL_LIB selects it)."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 625
model = sys.argv[2] if len(sys.argv) > 2 else "eucm"
sp = synth.make_problem(frames, model)
ctx = Context(0); p = Problem.from_synth(ctx, sp)
p.upload_params(sp.intr0, sp.poses0, sp.extr0)
iter_form = os.environ.get("CCAL_ITER_ROWS", "") != "0"
if iter_form:
    from camera_intrinsic_calibration_rs_amd.engine import default_opts
    for _ in range(10): p.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(0))      # single-launch groups: the last evaluating launch's stamps
else:
    for _ in range(30): p.build_normal_dev(0.0)
torch.cuda.synchronize()
lib = _ffi.load()
n = 16 * 4096
buf = np.zeros(n, dtype=np.float64)
lib.ccal_debug_fcbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lib.ccal_debug_fcbuf(p.handle, buf.ctypes.data_as(C.c_void_p), n)
t = buf.reshape(-1, 16)
t = t[t[:, 0] > 0]
if iter_form and t[:, 7].max() > 0:
    print(f"single-launch form, front of the kernel: rows summed {np.median(t[:, 7] - t[:, 0]) / 100:.2f} us (max {(t[:, 7] - t[:, 0]).max() / 100:.2f}), decision + camera solve {np.median(t[:, 8] - t[:, 7]) / 100:.2f} (max {(t[:, 8] - t[:, 7]).max() / 100:.2f}), publish {np.median(t[:, 9] - t[:, 8]) / 100:.2f}; then:")
    t = np.concatenate([t[:, 9:10], t[:, 1:7]], axis=1)
else:
    t = t[:, :7]
names = ["state arrives", "prologue (pose, back-substitution, exponential map)", "corner loop", "lane sums through LDS", "record assembly", "fused elimination + partial row"]
d = np.diff(t, axis=1) / 100.0
print(f"{frames} frames {model}: {len(t)} wavefronts; first start to last end {(t[:, 6].max() - t[:, 0].min()) / 100:.2f} us; start spread {(t[:, 0].max() - t[:, 0].min()) / 100:.2f} us")
for i, nm in enumerate(names): print(f"  {nm:55s} median {np.median(d[:, i]):6.2f} us   max {d[:, i].max():6.2f}")
print(f"  {'wavefront total':55s} median {np.median(t[:, 6] - t[:, 0]) / 100:6.2f} us   max {(t[:, 6] - t[:, 0]).max() / 100:6.2f}")
