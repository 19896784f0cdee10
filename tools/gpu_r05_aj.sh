#!/bin/bash
# Round 5, call aj: bench.py --gpus 2 on a one-GPU box must refuse loudly (non-zero exit, no line with n_gpus 1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05aj; mkdir -p $O
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/out.json 2> $O/err.txt; echo "rc $?"; wc -c $O/out.json; tail -2 $O/err.txt
