"""Developer tool: where a wavefront of k_gram2 spends its life (100 MHz clock at the phase boundaries; library built with
-DCCAL_STAMPS: tools/build_tu_variants.sh ccal_kernels_gram2 "stamps:-DCCAL_STAMPS"; CCAL_LIB selects it, CCAL_GRAM2=1 for
the models that do not take k_gram2 by default).   python tools/stamps_g2.py [frames] [model] [ragged]
`ragged`: 24 .. 144 corners per frame - the binned launch (k_gram2b); the table is then per quarter of the launch order (the bins
of the large frames come first)."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
model = sys.argv[2] if len(sys.argv) > 2 else "eucm"
RAGGED = "ragged" in sys.argv[3:]
sp = synth.make_problem(F, model, ragged=RAGGED, seed=0xC0FFEE + 77) if RAGGED else synth.make_problem(F, model)
ctx = Context(0); p = Problem.from_synth(ctx, sp)
p.upload_params(sp.intr0, sp.poses0, sp.extr0)
for _ in range(30): p.build_normal_dev(0.0)
torch.cuda.synchronize()
lib = _ffi.load()
n = min(F * 40, 8 * 16384) // 32 * 32
buf = np.zeros(n, dtype=np.float64)
lib.ccal_debug_fcbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
lib.ccal_debug_fcbuf(p.handle, buf.ctypes.data_as(C.c_void_p), n)
st = buf.reshape(-1, 32)
ok = st[:, 0] > 0
st = st[ok]
if len(st) == 0:
    # no wavefront left its stamps: the library is not a -DCCAL_STAMPS build, or this size / model does not take k_gram2
    print(f"{model} {F} frames: no stamps (CCAL_LIB must point at a -DCCAL_STAMPS build; CCAL_GRAM2=1 for sizes / models that take k_gram1v by default)")
    sys.exit(0)
t0 = st[:, 0].min()
nw = len(st)
names = ["state+ids", "prologue", "corner loop", "reduction", "scatter", "fused tail"]
print(f"{model} {F} frames: {nw} wavefronts, start spread {(st[:, 0].max() - t0) / 100:.2f} us, last end {(st[:, 5].max() - t0) / 100:.2f} us")
half = nw // 2
parts = [("first half of the dispatch (older on their SIMD)", slice(0, half)), ("second half (younger)", slice(half, nw))]
if RAGGED:
    q = max(nw // 8, 1)
    parts = [(f"wavefronts {i * q} .. {min((i + 1) * q, nw) - 1}", slice(i * q, min((i + 1) * q, nw))) for i in range((nw + q - 1) // q)]
for label, sel in parts:
    s = st[sel]
    d = np.diff(s[:, :6], axis=1) / 100.0
    if len(s) == 0:
        continue
    life = np.maximum(s[:, 5] - s[:, 0], 1)
    print(f"  {label}: start {np.median(s[:, 0] - t0) / 100:.2f}  " + "  ".join(f"{nm} {np.median(d[:, i]):.2f}" for i, nm in enumerate(names[1:])) +
          f"  total {np.median(s[:, 5] - s[:, 0]) / 100:.2f}  end {np.median(s[:, 5] - t0) / 100:.2f} us  (corner loop = {100 * np.median((s[:, 2] - s[:, 1]) / life):.0f} % of the wavefront's life)")
    if s[:, 24].max() > 0:      # stations of the prologue: state + intrinsics arrived | pose arrived | exponential map done | first corner rows arrived | constants in LDS
        pro = np.concatenate([s[:, 0:1], s[:, 24:28], s[:, 1:2]], axis=1)
        dp = np.diff(pro, axis=1) / 100.0
        print("      prologue: " + "  ".join(f"{n} {np.median(dp[:, i]):.2f}" for i, n in enumerate(["state + intrinsics", "pose", "exp map", "first rows", "LDS + barrier"])))
    if s[:, 16].max() > 0:      # stations of the epilogue (100 MHz): slice 0 parked | slice 0 summed | last slice summed || tail: C in rvec basis | Cholesky | Y | Y^T Y | sums parked
        ep = s[:, 16:24]
        seq = np.concatenate([s[:, 2:3], ep[:, 0:3], s[:, 3:5], ep[:, 3:8], s[:, 5:6]], axis=1)
        d2 = np.diff(seq, axis=1) / 100.0
        nm2 = ["park slice 0", "sum slice 0", "other slices", "to stamp 3", "scatter", "tail: load C + phi->rvec", "Cholesky", "Y columns", "wsync + pf + YtY", "park sums", "row sums + store"]
        print("      epilogue: " + "  ".join(f"{n} {np.median(d2[:, i]):.2f}" for i, n in enumerate(nm2)))
    if s[:, 12].max() > 0:      # -DCCAL_STAMPS=2: shader cycles per pass in the sections of the corner loop
        per = s[:, 8:12] / np.maximum(s[:, 12:13], 1)
        print("      cycles per pass: " + "  ".join(f"{nm} {np.median(per[:, i]):.0f}" for i, nm in enumerate(["back", "next rows", "rows + DPP", "Gram + chain"])) +
              f"  sum {np.median(per.sum(axis=1)):.0f}  ({np.median(s[:, 12]):.0f} passes; loop {np.median(s[:, 2] - s[:, 1]) / 100:.2f} us -> {np.median(per.sum(axis=1) * s[:, 12] / np.maximum(s[:, 2] - s[:, 1], 1)) / 10:.2f} GHz)")
