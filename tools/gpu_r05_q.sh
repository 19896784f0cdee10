#!/bin/bash
# Round 5, call q: the whole GPU suite on the library with the neighbouring-lane k_gram2, then a bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05q; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
timeout 600 python bench.py --no-traffic > $O/bench.json 2> $O/bench.err; python tools/bench_summary.py $O/bench.json 2>/dev/null | head -40 || head -c 1500 $O/bench.json
