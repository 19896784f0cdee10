#!/bin/bash
# Round 5, call ae: bench.py once more with the recounted flops.json (KB4's Hankel block) -> profiles/r05/bench_full.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/prof_r05
timeout 600 python3 bench.py > gpurun_out/prof_r05/bench_full.json 2> gpurun_out/prof_r05/bench_full.err; tail -c 300 gpurun_out/prof_r05/bench_full.json
