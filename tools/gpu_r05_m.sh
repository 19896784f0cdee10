#!/bin/bash
# Round 5, call m: k_gram2 with the u / v rows on NEIGHBOURING lanes (mirrored v lanes, DPP trade, 32-bit corner offsets) against the
# v_permlane32_swap form (lib/variants/libccal_g2swap.so = the library before the change): parity first, then build times;
# mode E with plain instead of non-temporal stores (tools/ubench/hbm_stream: a trivial fill is faster with plain stores at 276 MB).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05m; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_normal.py tests/test_gpu_iter.py tests/test_gpu_configs.py tests/test_gpu_boundary.py tests/test_gpu_dist.py -x -q -m gpu > $O/pytest_g2.log 2>&1; echo "pytest rc $?" >> $O/pytest_g2.log
tail -5 $O/pytest_g2.log
{
echo "== single camera, 10 000 frames"; python tools/ab_build.py "swap@g2swap,pair" eucm,ucm,opencv5,kb4 10000 3
echo "== one focal"; python tools/ab_build.py "swap@g2swap,pair" eucm,opencv5 10000 2 --one-focal
echo "== two cameras"; python tools/ab_build.py "swap@g2swap,pair" eucm 10000 2 --cams 2
echo "== other sizes"; python tools/ab_build.py "swap@g2swap,pair" eucm 2500,20000,50000 2
echo "== ragged"; python tools/ab_build.py "swap@g2swap,pair" eucm 10000 2 --ragged
} > $O/ab_g2_pair.txt 2>&1
cat $O/ab_g2_pair.txt
{
echo "== mode E eucm 10000 (GB/s)"; python tools/ab_eval.py base,evplain,evplainr 10000 3
echo "== kb4"; python tools/ab_eval.py base,evplain 10000 2 --model kb4
echo "== two cameras"; python tools/ab_eval.py base,evplain 10000 2 --cams 2
echo "== 1000 / 50000"; python tools/ab_eval.py base,evplain 1000,50000 2
} > $O/ab_eval_plain.txt 2>&1
cat $O/ab_eval_plain.txt
