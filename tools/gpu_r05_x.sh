#!/bin/bash
# Round 5, call x: the two wavefronts of a SIMD alternating the issue priority every 1 / 2 / 3 passes (k_gram2, -DCCAL_G2_PRIO_ALT=n)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05x; mkdir -p $O
{
echo "== eucm"; python tools/ab_build.py "base,prio1@prio1,prio2@prio2,prio3@prio3" eucm 10000,20000,5000 3
echo "== ucm, two cameras"; python tools/ab_build.py "base,prio1@prio1,prio2@prio2" ucm 10000 2; python tools/ab_build.py "base,prio1@prio1,prio2@prio2" eucm 10000 2 --cams 2
} > $O/ab_prio.txt 2>&1
cat $O/ab_prio.txt
