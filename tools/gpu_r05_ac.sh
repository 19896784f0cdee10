#!/bin/bash
# Round 5, call ac: k_eval without the two table reads in front of the pose (identity list / slot tables) against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05ac; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_eval.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== mode E eucm (GB/s): base = identity fast path, prev = the library before"; python tools/ab_eval.py base,prev 10000 5
echo "== kb4 / opencv5"; python tools/ab_eval.py base,prev 10000 3 --model kb4; python tools/ab_eval.py base,prev 10000 3 --model opencv5
echo "== 1000 / 2500 / 50000"; python tools/ab_eval.py base,prev 1000,2500,50000 3
} > $O/ab_eval_ident.txt 2>&1
cat $O/ab_eval_ident.txt
