#!/bin/bash
# Round 6, call y: the shipped library (result spread up to 256 KB of poses): the suites that touch it + host-pointer costs per size
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06y; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; grep -n "passed\|failed" $O/pytest_full.log | tail -2
python tools/host_pointer_ab.py "10000,10000:ragged,5000,5000:ragged,3000,2500,625" 10 > $O/host_pointer_costs.txt 2>&1; cat $O/host_pointer_costs.txt
timeout 600 python tools/fuzz_parity.py --seed 707 --seconds 240 --big 0.3 > $O/fuzz_seed707.json 2> $O/fuzz_seed707.err; echo "fuzz rc $?"; grep -n "n_fail\|\"cases\"" $O/fuzz_seed707.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 1400 $O/bench.json
