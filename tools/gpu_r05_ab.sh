#!/bin/bash
# Round 5, call ab: VERDICT r04 #3's table - mode-N build times of bench.py (extra.mode_N_build_ms, extra.config2) on ONE box, library of
# commit c15c8ee (lib/variants/libccal_g2swap.so) against the final one, with the bench's 0.1 s clock ramp and without it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05ab; mkdir -p $O
for lib in g2swap final; do for ramp in 0.1 0; do
  if [ $lib = g2swap ]; then export CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_g2swap.so; else unset CCAL_LIB; fi
  CCAL_BENCH_CLOCK_RAMP_S=$ramp timeout 400 python bench.py --no-traffic --no-cpu-baseline --no-rig > $O/bench_${lib}_ramp$ramp.json 2> $O/bench_${lib}_ramp$ramp.err
  python - <<PY
import json
d=json.load(open("$O/bench_${lib}_ramp$ramp.json")); ex=d["extra"]
print("$lib ramp $ramp: headline %.2f us | mode N build EUCM %.2f us  KB4 %.2f  OPENCV5 %.2f | GN %.3f ms" % (d["roofline"]["kernel_ms"]*1e3, ex["mode_N_build_ms"]*1e3, ex["config2"]["kb4"]["mode_N_build_ms"]*1e3, ex["config2"]["opencv5"]["mode_N_build_ms"]*1e3, ex["gn_solve_ms"]))
PY
done; done | tee $O/table.txt
