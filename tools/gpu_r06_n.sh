#!/bin/bash
# Round 6, call n: the plain launches after the state reads of gram2_body went back to where they were (non-ITER): against the commit before k_gram2i
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06n; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_iter.py tests/test_gpu_normal.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
{
python tools/ab_build.py "base,pre@pre_iter" eucm,kb4,opencv5,ucm 10000 5
python tools/ab_build.py "base,pre@pre_iter" kb4 20000 3
python tools/ab_build.py "base,pre@pre_iter" kb4 10000 3 --one-focal
} > $O/ab_iter_body_fixed.txt 2>&1
cat $O/ab_iter_body_fixed.txt
