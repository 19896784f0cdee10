#!/bin/bash
# round-4 GPU pass F: k_head's early status word - tests, bench x2 (session-size numbers), a short sharded fuzz
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04f; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" > $O/summary.txt
for i in 1 2; do timeout 500 python3 bench.py > $O/bench_$i.json 2> $O/bench_$i.err; echo "bench $i rc $?" >> $O/summary.txt; done
timeout 400 python3 tools/fuzz_parity.py --seconds 240 --seed 77 --shards 3 > $O/fuzz.json 2> $O/fuzz.err; echo "fuzz rc $?" >> $O/summary.txt
tail -3 $O/pytest.log; cat $O/summary.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04f/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['extra']
        print(f, d['value'], {k:e[k] for k in ('frames625','concurrent_sessions') if k in e})
    except Exception as x: print(f, 'ERR', x)
d=json.load(open('gpurun_out/r04f/fuzz.json')); print({k:d[k] for k in ('cases','n_fail','sharded_cases','worst','worst_sharded_vs_unsharded')})
PY
