#!/bin/bash
# Round 6, call i: single-launch groups on the two-wavefronts-per-SIMD kernel (k_gram2i): parity (tests/test_gpu_iter.py holds the forms against each
# other and the oracle; normal / configs / api / batch / multi), A/B against the same library without it (nog2i), kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06i; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_iter.py tests/test_gpu_normal.py tests/test_gpu_configs.py tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
{
echo "== GN / LM (host pointers) with k_gram2i (base) and without (nog2i): full frames"
python tools/ab_build.py "base,nog2i@nog2i" eucm,ucm 10000,5000,2500,8000 3
python tools/ab_build.py "base,nog2i@nog2i" eucm 10000 3 --one-focal
echo "== ragged"
python tools/ab_build.py "base,nog2i@nog2i" eucm 10000,5000 3 --ragged
} > $O/ab_g2i.txt 2>&1
cat $O/ab_g2i.txt
bash tools/kstats.sh --what solve --reps 50 > $O/kstats_solve.txt 2>&1; cat $O/kstats_solve.txt
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
ctx = Context(0)
for frames in (10000, 5000):
    sp = synth.make_problem(frames, "eucm")
    p = Problem.from_synth(ctx, sp)
    for method in (0, 1):
        bd = 1e9
        for _ in range(8):
            p.upload_params(sp.intr0, sp.poses0, sp.extr0); r = p.solve_dev(default_opts(method)); bd = min(bd, r.solve_ms)
        print(f"{frames} frames {'LM' if method else 'GN'} ccal_solve_dev {bd:.4f} ms ({r.iterations} it)")
    p.close()
PY
