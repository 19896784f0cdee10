#!/bin/bash
# Round 6, call z: the planner's pick (the shipped library) against equalised and folded plans by hand at sizes BETWEEN the ones it was calibrated on
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06z; mkdir -p $O
SPECS="shipped"
for T in 5 6 7 8 9 10 11 12 14 16 20; do SPECS="$SPECS,T$T@penv:CCAL_G2_PLAN=T:$T"; done
for L in 8 12 16 32; do SPECS="$SPECS,fold$L@penv:CCAL_G2_PLAN=fold:$L"; done
python tools/ab_build.py "$SPECS" eucm 5000,7000,11000,14000,18000,30000 3 --ragged > $O/ab_g2_plans_between.txt 2>&1
grep -v "^gram2_bin_plan" $O/ab_g2_plans_between.txt | awk '{print $2, $3, $6}' | sort -k1,1n -k3,3n | awk '{ if ($1 != last) { print ""; last = $1 } printf "%s %s %s | ", $1, $2, $3 }'; echo
