#!/bin/bash
# Round 6, call o: the final library against ROUND 5's (commit edc3b4c) on the uniform workloads the round did not mean to change
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06o; mkdir -p $O
export CCAL_LIB_ALLOW_MISSING=1
{
python tools/ab_build.py "base,r05@r05" eucm,kb4,opencv5,ucm 10000 5
python tools/ab_build.py "base,r05@r05" eucm,kb4 2500,20000,50000 3
python tools/ab_build.py "base,r05@r05" eucm,opencv5 10000 3 --cams 2
python tools/ab_build.py "base,r05@r05" eucm 625,1000 3
echo "== ragged (round 5 had no bins)"
python tools/ab_build.py "base,r05@r05" eucm,kb4 10000,20000,50000 3 --ragged
python tools/ab_build.py "base,r05@r05" eucm,kb4 10000 3 --ragged --cams 2
} > $O/ab_vs_r05.txt 2>&1
cat $O/ab_vs_r05.txt
