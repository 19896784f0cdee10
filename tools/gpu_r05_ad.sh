#!/bin/bash
# Round 5, call ad: KB4's Hankel block (7 accumulators for the 10 distortion x distortion entries) against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05ad; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_normal.py tests/test_gpu_configs.py tests/test_gpu_multi.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== KB4 (prev = the library before)"; python tools/ab_build.py "prev@prev,hankel" kb4 10000 5; python tools/ab_build.py "prev@prev,hankel" kb4 10000 3 --one-focal; python tools/ab_build.py "prev@prev,hankel" kb4 2500,20000 3; python tools/ab_build.py "prev@prev,hankel" kb4 10000 2 --cams 2
} > $O/ab_kb4_hankel.txt 2>&1
cat $O/ab_kb4_hankel.txt
CCAL_GRAM2=1 python tools/fuzz_parity.py --seconds 60 --seed 80808 > $O/fuzz_gram2.json 2> $O/fuzz_gram2.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r05ad/fuzz_gram2.json")); print("fuzz", d["cases"], d["n_fail"], d["worst"])
PY
