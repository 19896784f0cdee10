#!/bin/bash
# Round 6, call zd: the Huber branch from the reciprocal-square-root seed (base) against the IEEE expansions (ieee): parity suites + builds with outliers
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06zd; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; grep -n "passed\|failed" $O/pytest_full.log | tail -2
{
for F in 0.0 0.01 0.05 0.2; do echo "== outliers $F"; python tools/ab_build.py "base,ieee@ieee" eucm,kb4 10000 3 --outliers $F; done
} > $O/ab_huber_fast.txt 2>&1
cat $O/ab_huber_fast.txt
timeout 600 python tools/fuzz_parity.py --seed 61010 --seconds 240 --shards 3 --batch 6 > $O/fuzz_seed61010_shards3_batch6.json 2> $O/fuzz.err; echo "fuzz rc $?"; grep -n "\"cases\"\|n_fail" $O/fuzz_seed61010_shards3_batch6.json
