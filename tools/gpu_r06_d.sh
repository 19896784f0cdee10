#!/bin/bash
# Round 6, call d: does pairing a LONG older wavefront with a SHORT younger one on every SIMD pay?  (plans from the environment, experiment build);
# context block cache + packed upload: config0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06d; mkdir -p $O
{
echo "== uniform 10000 x 144, EUCM: plans lpf:frames in launch order"
python tools/ab_build.py "base,p8_32@planenv:CCAL_G2_PLAN=8:8192+32:1808,p8_16@planenv:CCAL_G2_PLAN=8:8000+16:2000,p8_12@planenv:CCAL_G2_PLAN=8:8192+12:1808,p6_32@planenv:CCAL_G2_PLAN=6:9000+32:1000,p12@planenv:CCAL_G2_PLAN=12:10000,p16_8@planenv:CCAL_G2_PLAN=16:2000+8:8000" eucm 10000 3
echo "== KB4 / OPENCV5 (one wavefront per SIMD)"
python tools/ab_build.py "base,p6_32@planenv:CCAL_G2_PLAN=6:9000+32:1000,p8_6@planenv:CCAL_G2_PLAN=8:4096+6:5904,p12_6@planenv:CCAL_G2_PLAN=12:2560+6:7440" kb4,opencv5 10000 3
echo "== ragged: the shipped plan against long-first / short-second plans"
python tools/ab_build.py "base,r12_8_6@planenv:CCAL_G2_PLAN=12:5120+8:2400+6:2480,r16_12_6@planenv:CCAL_G2_PLAN=16:1500+12:3500+6:5000,r12_6@planenv:CCAL_G2_PLAN=12:5120+6:4880" eucm 10000 3 --ragged
} > $O/ab_plans.txt 2>&1
cat $O/ab_plans.txt
timeout 600 python bench.py --no-cpu-baseline --no-traffic --no-rig --steps 200 --warmup 50 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06d/bench.json").read().strip().split("\n")[-1])
print(json.dumps(d["extra"]["config0"]["gpu_ms"])); print(json.dumps(d["summary"]))
PY
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
