import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import api, synth
from camera_intrinsic_calibration_rs_amd.engine import Context, CcalError
from oracle import binding as ob
ctx = Context(0); rng = np.random.default_rng(3)
B = {"ucm": ([0,0,0,0,1e-6],[1e4,1e4,0,0,1.0]), "eucm": ([0,0,0,0,1e-6,1e-6],[1e4,1e4,0,0,1.0,100.0]),
     "kb4": ([0,0,0,0,-1,-1,-1,-1],[1e4,1e4,0,0,1,1,1,1]), "opencv5": ([0,0,0,0,-1,-1,-1,-1,-1],[1e4,1e4,0,0,1,1,1,1,1])}
def rnd(model, w, h):
    f = rng.uniform(0.3, 0.9) * w; cx = w/2 + rng.uniform(-10, 10); cy = h/2 + rng.uniform(-10, 10)
    if model == "ucm": return [f, f*rng.uniform(0.98,1.02), cx, cy, rng.uniform(0.3, 0.75)]
    if model == "eucm": return [f, f*rng.uniform(0.98,1.02), cx, cy, rng.uniform(0.3, 0.75), rng.uniform(0.8, 1.3)]
    if model == "kb4": return [f, f, cx, cy] + list(rng.uniform(-0.01, 0.01, 4))
    return [f*1.5, f*1.5, cx, cy, rng.uniform(-0.2, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-1e-3,1e-3), rng.uniform(-1e-3,1e-3), 0.0]
init = {"ucm": [0,0,0,0,0.6], "eucm": [0,0,0,0,0.5,1.0], "kb4": [0.0]*8, "opencv5": [0.0]*9}
n = 0; bad = []; worst = 0.0
for _ in range(150):
    src, tgt = rng.choice(list(B)), rng.choice(list(B))
    if src == "ucm" and tgt == "eucm": continue
    w, h = (512, 512) if rng.random() < 0.5 else (int(rng.integers(400, 1400)), int(rng.integers(300, 1000)))
    sp_ = rnd(src, w, h); dis = int(rng.integers(0, 2)) if tgt in ("kb4", "opencv5") else 0
    lo, hi = [list(x) for x in B[tgt]]; hi[2] = w; hi[3] = h
    s = api.GenericModel(src, sp_, w, h); t = api.GenericModel(tgt, init[tgt], w, h)
    po, npts, rc = ob.convert_model(s.model_id, sp_, t.model_id, init[tgt], w, h, dis, lo, hi)
    try:
        out = api.convert_model(s, t, dis, ctx=ctx); pg = np.asarray(out.params()); rg = 0
    except CcalError as e:
        pg = None; rg = e.code
    n += 1
    if rc != rg: bad.append(dict(src=str(src), tgt=str(tgt), w=w, h=h, rc_o=rc, rc_g=rg, p=sp_)); continue
    if rc == 0:
        d = float(np.max(np.abs(pg - po) / np.maximum(np.abs(po), 1e-3)))
        worst = max(worst, d)
        if d > 1e-6: bad.append(dict(src=str(src), tgt=str(tgt), w=w, h=h, d=d, p=sp_, po=list(po), pg=list(pg)))
print(json.dumps(dict(cases=n, worst=worst, n_bad=len(bad), bad=bad[:4])))
