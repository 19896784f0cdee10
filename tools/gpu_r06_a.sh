#!/bin/bash
# Round 6, call a: parity of the round's first changes (error_metric, per-member batch failures, ragged parity cases, INPROC n = 1),
# the full bench line with `summary` and `extra.ragged`, and the tools of the multi-GPU runbook on the one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06a; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_normal.py tests/test_gpu_configs.py tests/test_gpu_batch.py tests/test_gpu_multi.py tests/test_gpu_iter.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"; tail -c 1700 $O/bench_full.json
timeout 300 tools/ubench/allreduce_latency.bin 1 > $O/allreduce_latency_n1.txt 2>&1; cat $O/allreduce_latency_n1.txt
for TR in inproc rccl; do CCAL_MULTI_TRANSPORT=$TR timeout 600 python tools/sharded_ab.py 1 > $O/sharded_$TR.json 2> $O/sharded_$TR.err; head -c 900 $O/sharded_$TR.json; echo; done
