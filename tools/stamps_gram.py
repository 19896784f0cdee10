"""Developer tool: start / end time of every wavefront of k_gram1w (10 ns clock), from a library built with
-DCCAL_STAMPS (tools/build_variants.sh "stamps:-DCCAL_STAMPS", CCAL_LIB=.../libccal_stamps.so).  WPBV = wavefronts per
workgroup of that build (CCAL_GRAMV_WPB, default 2)."""
import os, sys, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from camera_intrinsic_calibration_rs_amd import synth, _ffi
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sp = synth.make_problem(F, "eucm")
ctx = Context(0); p = Problem.from_synth(ctx, sp)
p.upload_params(sp.intr0, sp.poses0, sp.extr0)
for _ in range(30): p.build_normal_dev(0.0)
torch.cuda.synchronize()
lib = _ffi.load()
n = 4096
buf = np.zeros(n, dtype=np.float64)
lib.ccal_debug_fcbuf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
rc = lib.ccal_debug_fcbuf(p.handle if hasattr(p, "handle") else p.h, buf.ctypes.data_as(C.c_void_p), n)
st = buf[0::2]; en = buf[1::2]
ok = st > 0
st = st[ok]; en = en[ok]
t0 = st.min()
print("waves", ok.sum(), "start spread us", (st.max() - t0) / 100.0, "end first/last us", (en.min() - t0) / 100.0, (en.max() - t0) / 100.0,
      "life us median", np.median(en - st) / 100.0, "p5/p95", np.percentile(en - st, 5) / 100.0, np.percentile(en - st, 95) / 100.0)
print("start percentiles us", [round((np.percentile(st, q) - t0) / 100.0, 2) for q in (10, 50, 90, 99)])
print("end percentiles us", [round((np.percentile(en, q) - t0) / 100.0, 2) for q in (1, 10, 50, 90, 99)])
life = (en - st) / 100.0
idx = np.nonzero(ok)[0]
WPBV = int(os.environ.get("WPBV", "2")); wg = idx // WPBV
slow = life > 0.5 * (np.median(life) + life.max())
print("slow fraction", slow.mean(), "slow both waves of WG?", np.mean(slow[0::2] == slow[1::2]))
for m in (8, 16, 32, 64, 256):
    frac = [round(float(slow[(wg % m) == k].mean()), 2) for k in range(min(m, 16))]
    print("mod", m, frac)
# first / last WG indices
print("slow wg idx head", wg[slow][:40].tolist())
print("life by wg decile", [round(float(np.median(life[(wg >= q * len(life) // 20) & (wg < (q + 1) * len(life) // 20)])), 1) for q in range(10)])
nw = len(life)
print("life by 50-WG bins", [round(float(np.median(life[(wg >= b) & (wg < b + 50)])), 1) for b in range(0, int(wg.max()) + 1, 50)])
print("end by 50-WG bins", [round(float(np.median((en[(wg >= b) & (wg < b + 50)] - t0) / 100.0)), 1) for b in range(0, int(wg.max()) + 1, 50)])
