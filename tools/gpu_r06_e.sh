#!/bin/bash
# Round 6, call e: asymmetric pairs with the 10- and 20-lane mappings (experiment build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06e; mkdir -p $O
{
echo "== uniform 10000 x 144, EUCM"
python tools/ab_build.py "base,p10_16@planenv:CCAL_G2_PLAN=10:6144+16:3856,p10@planenv:CCAL_G2_PLAN=10:10000,p16@planenv:CCAL_G2_PLAN=16:10000,p12_16@planenv:CCAL_G2_PLAN=12:5120+16:4880,p10_12@planenv:CCAL_G2_PLAN=10:6144+12:3856,p16_10@planenv:CCAL_G2_PLAN=16:3856+10:6144" eucm 10000 3
echo "== ragged"
python tools/ab_build.py "base,r10@planenv:CCAL_G2_PLAN=16:1400+12:2200+10:2500+8:1500+6:2400,r20@planenv:CCAL_G2_PLAN=20:1322+16:2645+12:2645+8:1322+6:2066" eucm 10000 3 --ragged
} > $O/ab_plans2.txt 2>&1
cat $O/ab_plans2.txt
