#!/bin/bash
# Round 5, call s: KB4 on k_gram2, fused KB4 / OPENCV5 row formation; parity of everything that builds normal equations, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05s; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_batch.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
echo "== 10 000 frames (swap = the library of commit c15c8ee)"; python tools/ab_build.py "swap@g2swap,new" eucm,kb4,opencv5 10000 3
echo "== one focal"; python tools/ab_build.py "swap@g2swap,new" kb4,opencv5 10000 2 --one-focal
echo "== two cameras"; python tools/ab_build.py "swap@g2swap,new" kb4,opencv5 10000 2 --cams 2
echo "== other sizes"; python tools/ab_build.py "swap@g2swap,new" kb4,opencv5 2500,20000 2
echo "== ragged"; python tools/ab_build.py "swap@g2swap,new" kb4,opencv5 10000 2 --ragged
} > $O/ab_kb4_ocv5.txt 2>&1
cat $O/ab_kb4_ocv5.txt
