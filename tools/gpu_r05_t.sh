#!/bin/bash
# Round 5, call t: k_gram2 prologue with the elimination record staged in LDS (back-substitution): parity, A/B against commit c15c8ee's library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05t; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (swap = the library of commit c15c8ee)"; python tools/ab_build.py "swap@g2swap,new" eucm,kb4,opencv5,ucm 10000 3
echo "== other sizes"; python tools/ab_build.py "swap@g2swap,new" eucm 2500,20000,50000 2
echo "== two cameras"; python tools/ab_build.py "swap@g2swap,new" eucm 10000 2 --cams 2
} > $O/ab_stage.txt 2>&1
cat $O/ab_stage.txt
timeout 300 python bench.py --no-traffic --no-extra > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json 2>/dev/null | head -3
