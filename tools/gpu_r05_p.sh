#!/bin/bash
# Round 5, call p: k_gram2 with neighbouring-lane rows + fused UCM / EUCM row formation (no software pipeline) against the library before
# the library before both changes (lib/variants/libccal_g2swap.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05p; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest_g2.log 2>&1; echo "pytest rc $?" >> $O/pytest_g2.log
tail -5 $O/pytest_g2.log
{
echo "== single camera, 10 000 frames"; python tools/ab_build.py "swap@g2swap,pairf" eucm,ucm,opencv5,kb4 10000 3
echo "== one focal"; python tools/ab_build.py "swap@g2swap,pairf" eucm,opencv5 10000 2 --one-focal
echo "== two cameras"; python tools/ab_build.py "swap@g2swap,pairf" eucm,kb4 10000 2 --cams 2
echo "== other sizes"; python tools/ab_build.py "swap@g2swap,pairf" eucm 2500,5000,20000,50000 2
echo "== ragged"; python tools/ab_build.py "swap@g2swap,pairf" eucm 10000 2 --ragged
} > $O/ab_g2_pairf.txt 2>&1
cat $O/ab_g2_pairf.txt
