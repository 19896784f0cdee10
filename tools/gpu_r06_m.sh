#!/bin/bash
# Round 6, call m: did giving gram2_body its single-launch form (template parameter ITER, the state in a register struct) cost the plain launches anything?
# base = the final library, pre_iter = the commit before k_gram2i
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06m; mkdir -p $O
{
python tools/ab_build.py "base,pre@pre_iter" eucm,kb4,opencv5,ucm 10000 5
python tools/ab_build.py "base,pre@pre_iter" eucm,kb4 20000 3
python tools/ab_build.py "base,pre@pre_iter" eucm,kb4 10000 3 --one-focal
python tools/ab_build.py "base,pre@pre_iter" eucm,kb4 10000 3 --cams 2
python tools/ab_build.py "base,pre@pre_iter" eucm,kb4 20000 3 --ragged
} > $O/ab_iter_body.txt 2>&1
cat $O/ab_iter_body.txt
