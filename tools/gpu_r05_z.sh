#!/bin/bash
# Round 5, call z: fused tail with interleaved Y columns and packed parking positions: parity, A/B against commit c15c8ee's library
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05z; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (swap = the library of commit c15c8ee)"; python tools/ab_build.py "swap@g2swap,new" eucm,kb4,opencv5 10000 3
echo "== sessions"; python tools/ab_build.py "swap@g2swap,new" eucm,kb4,opencv5 625 3
} > $O/ab_tail2.txt 2>&1
cat $O/ab_tail2.txt
