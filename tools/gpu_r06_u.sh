#!/bin/bash
# Round 6, call u: calibrating the plan model for ragged frames: equalised plans (T:k) and folded plans (fold:lanes) by hand against the planner's choice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06u; mkdir -p $O
export CCAL_G2_PLAN_PRINT=1
SPECS="auto@penv"
for T in 6 7 8 9 10 11 12 14; do SPECS="$SPECS,T$T@penv:CCAL_G2_PLAN=T:$T"; done
for L in 6 8 12 16 32; do SPECS="$SPECS,fold$L@penv:CCAL_G2_PLAN=fold:$L"; done
python tools/ab_build.py "$SPECS" eucm 6000,8000,10000,12000,16000,20000 3 --ragged > $O/ab_g2_plans.txt 2>&1
cat $O/ab_g2_plans.txt
