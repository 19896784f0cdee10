#!/bin/bash
# Round 5, call af: Y kept column by column in LDS (16-byte accesses) in the fused tail: parity, A/B against the previous library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05af; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (prev = the library before)"; python tools/ab_build.py "prev@prev,ycol" eucm,kb4,opencv5 10000 5
echo "== sessions"; python tools/ab_build.py "prev@prev,ycol" eucm,kb4,opencv5 625 3
} > $O/ab_ycol.txt 2>&1
cat $O/ab_ycol.txt
