#!/bin/bash
# Round 6, call c: pinned caller buffers (ccal_pin_buffer): parity + what the staging copies cost; stamps of the binned Gram launch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_normal.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
python - > $O/pinned_ab.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
ctx = Context(0)
for frames, cams in ((10000, 1), (2500, 1), (625, 1), (10000, 2)):
    sp = synth.make_problem(frames, "eucm", n_cams=cams)
    p = Problem.from_synth(ctx, sp)
    for method in (0, 1):
        row = []
        for pinned in (False, True):
            best = 1e9
            for _ in range(8):
                r = p.solve(sp.intr0, sp.poses0, sp.extr0, opts=default_opts(method), pinned=pinned)[3]
                best = min(best, r.solve_ms)
            row.append(best)
        p.upload_params(sp.intr0, sp.poses0, sp.extr0); rd = min(p.solve_dev(default_opts(method)).solve_ms for _ in range(1))
        bd = 1e9
        for _ in range(8):
            p.upload_params(sp.intr0, sp.poses0, sp.extr0); bd = min(bd, p.solve_dev(default_opts(method)).solve_ms)
        print(f"{frames:6d} frames x {cams} cam  {'LM' if method else 'GN'}: ccal_solve pageable {row[0]:.4f} ms  pinned {row[1]:.4f} ms  ccal_solve_dev {bd:.4f} ms")
    p.close()
PY
cat $O/pinned_ab.txt
for a in "10000 eucm" "10000 eucm ragged" "20000 eucm ragged"; do CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_stamps1.so timeout 200 python3 tools/stamps_g2.py $a 2>&1 | grep -v amdgpu.ids; done > $O/stamps_g2_bins.txt
cat $O/stamps_g2_bins.txt
