#!/bin/bash
# Round 6, call j: k_gram2i restricted to launches that fill the chip (8 960 .. 10 240 frames): the suite + the bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06j; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_full.log 2>&1; echo "pytest rc $?" >> $O/pytest_full.log; tail -4 $O/pytest_full.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"; tail -c 1800 $O/bench_full.json
