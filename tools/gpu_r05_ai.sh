#!/bin/bash
# Round 5, call ai: the early requests of k_gram2's prologue for rigs too (GEN: extrinsics, per-camera intrinsics, both sets): parity, A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05ai; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_api.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== two cameras (prev = the library before)"; python tools/ab_build.py "prev@prev,early" eucm,kb4,opencv5 10000 3 --cams 2
echo "== one camera"; python tools/ab_build.py "prev@prev,early" eucm,opencv5 10000 3
} > $O/ab_early_gen.txt 2>&1
cat $O/ab_early_gen.txt
python tools/fuzz_parity.py --seconds 90 --seed 91919 --shards 3 > $O/fuzz.json 2> $O/fuzz.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r05ai/fuzz.json")); print("fuzz", d["cases"], d["n_fail"], d["worst"], d["sharded_cases"])
PY
