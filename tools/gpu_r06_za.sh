#!/bin/bash
# Round 6, call za: small ragged problems (2 000 .. 5 000 frames): the plain launch (the planner's answer there) against folded / equalised plans by hand;
# then the shipped planner over all sizes of the two sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06za; mkdir -p $O
SPECS="shipped"
for T in 3 4 5 6; do SPECS="$SPECS,T$T@penv:CCAL_G2_PLAN=T:$T"; done
for L in 12 16 32; do SPECS="$SPECS,fold$L@penv:CCAL_G2_PLAN=fold:$L"; done
python tools/ab_build.py "$SPECS" eucm 2000,2500,3000,4000,5000 3 --ragged > $O/ab_g2_plans_small.txt 2>&1
python tools/ab_build.py "shipped" eucm 6000,7000,8000,9000,10000,11000,12000,14000,16000,18000,20000,30000,50000 3 --ragged > $O/ab_g2_shipped_all.txt 2>&1
for f in $O/ab_g2_plans_small.txt $O/ab_g2_shipped_all.txt; do grep -v "^gram2_bin_plan" $f | awk '{print $2, $3, $6}' | sort -k1,1n -k3,3n | awk '{ if ($1 != last) { print ""; last = $1 } printf "%s %s %s | ", $1, $2, $3 }'; echo; done
