#!/bin/bash
# Round 6, call s: k_gram2i over the SORTED table for ragged frames (one bin): parity + A/B against the library before (prev)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06s; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_normal.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
python tools/ab_build.py "base,prev@prev" eucm,ucm 10000,9000 5 --ragged
python tools/ab_build.py "base,prev@prev" eucm 10000 3
} > $O/ab_g2i_sorted.txt 2>&1
cat $O/ab_g2i_sorted.txt
for L in base prev; do
  if [ $L = prev ]; then export CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_prev.so; else unset CCAL_LIB; fi
  python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from camera_intrinsic_calibration_rs_amd import synth
from camera_intrinsic_calibration_rs_amd.engine import Context, Problem, default_opts
ctx = Context(0)
sp = synth.make_problem(10000, "eucm", ragged=True, seed=0xC0FFEE + 77)
p = Problem.from_synth(ctx, sp)
for method in (0, 1):
    bd = 1e9
    for _ in range(12):
        p.upload_params(sp.intr0, sp.poses0, sp.extr0); r = p.solve_dev(default_opts(method)); bd = min(bd, r.solve_ms)
    print(os.environ.get("CCAL_LIB", "base")[-14:], f"10000 ragged frames {'LM' if method else 'GN'} ccal_solve_dev {bd:.4f} ms ({r.iterations} it, cost {r.final_cost:.6f})")
PY
done
