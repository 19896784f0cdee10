"""Developer tool: one convert_model case on the device and on the oracle, with the fit quality of both answers (RMS pixel
distance between the source model's and the fitted model's projections of the same rays over the image)."""
import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import api, synth
from camera_intrinsic_calibration_rs_amd.engine import Context
from oracle import binding as ob
src, tgt, w, h = "kb4", "eucm", 512, 512
p = [173.02532154340673, 173.02532154340673, 246.51561060595748, 252.6531347860707, 0.009828441630769657, 0.004541123831173459, -0.001445387267250355, -0.006012307680471536]
lo, hi = [0, 0, 0, 0, 1e-6, 1e-6], [1e4, 1e4, w, h, 1.0, 100.0]
s = api.GenericModel(src, p, w, h); t = api.GenericModel(tgt, [0, 0, 0, 0, 0.5, 1.0], w, h)
po, npts, rc = ob.convert_model(s.model_id, p, t.model_id, [0, 0, 0, 0, 0.5, 1.0], w, h, 0, lo, hi)
out = api.convert_model(s, t, 0, ctx=Context(0)); pg = np.asarray(out.params())
# rays: unit directions over a cone; compare projections
th = np.linspace(0.02, 1.2, 60); ph = np.linspace(0, 2 * np.pi, 48, endpoint=False)
T, P = np.meshgrid(th, ph); d = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], -1).reshape(-1, 3)
uv_s = synth.project(synth.MODEL_NAMES[src], p, d)
inside = (uv_s[:, 0] >= 0) & (uv_s[:, 0] <= w) & (uv_s[:, 1] >= 0) & (uv_s[:, 1] <= h)
for name, q in (("oracle", po), ("device", pg)):
    uv = synth.project(synth.MODEL_NAMES[tgt], list(q), d)
    print(name, [float(x) for x in np.round(q, 6)], "rms px over the image", float(np.sqrt(((uv - uv_s)[inside] ** 2).sum(1).mean())))
print("rc oracle", rc, "points", npts)
