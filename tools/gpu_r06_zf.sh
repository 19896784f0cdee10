#!/bin/bash
# Round 6, call zf: flakiness check of the final tree - the GPU suite three times, smoke, bench defaults, a long randomised sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06zf; mkdir -p $O
for i in 1 2 3; do timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_$i.log 2>&1; echo "pytest $i rc $? $(grep -h 'passed\|failed' $O/pytest_$i.log | tail -1)"; done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?"
( time timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err; tail -c 600 $O/bench.json
timeout 1200 python tools/fuzz_parity.py --seed 61111 --seconds 800 --shards 3 --batch 6 --big 0.12 --huge > $O/fuzz_seed61111_shards3_batch6_huge.json 2> $O/fuzz.err; echo "fuzz rc $?"; grep -n "\"cases\"\|n_fail" $O/fuzz_seed61111_shards3_batch6_huge.json
timeout 300 python tools/fuzz_convert.py --seconds 120 > $O/fuzz_convert.json 2> $O/fuzz_convert.err; echo "fuzz_convert rc $?"; tail -c 300 $O/fuzz_convert.json
