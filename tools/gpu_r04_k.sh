#!/bin/bash
# round-4 GPU pass K: quick look - session-size solves, side-by-side sessions, quick tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r04k; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_batch.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 500 python3 bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04k/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); e=d['extra']
    print(f.split('/')[-1], round(d['value']/1e9,2), {k:round(v['solve_ms'],4) for k,v in e['frames625'].items()}, {k:round(v['solve_ms'],4) for k,v in e['frames1000'].items()})
    print('   sessions', {k:(round(v['ms_per_batch'],4), round(v['speedup_vs_1'],2)) for k,v in e['concurrent_sessions']['by_sessions'].items()})
PY
for f in 625 2000; do timeout 100 python3 tools/time_kernels.py --frames $f --what solve | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['frames'], d['gn_ms'], d['lm_ms'])"; done
timeout 100 python3 tools/time_kernels.py --frames 625 --model ucm --what solve | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ucm', d['frames'], d['gn_ms'], d['lm_ms'])"
