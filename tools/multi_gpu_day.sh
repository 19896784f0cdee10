#!/bin/bash
# Multi-GPU day one, ONE command: everything DESIGN.md section 5 lists as "never executed on more than one device", measured and
# written under profiles/<round>/multi_gpu/ (tracked).  On a box with n visible GPUs:
#   1. pytest -m gpu tests/test_gpu_multi.py tests/test_gpu_dist.py   (device sets range(n); native RCCL ranks when n >= 2)
#   2. bench.py --gpus {1,2,4,8} weak scaling (10 000 frames per GPU) and --frames-total 50000 strong scaling (configs[3]);
#      a leg that asks for more GPUs than are visible is SKIPPED with a note (bench.py itself refuses with exit code 4: checked once)
#   3. extra.single_process_sharded with CCAL_MULTI_TRANSPORT=inproc | rccl (A/B of the two transports, one process, every GPU)
#   4. tools/ubench/allreduce_latency.bin: the step's 100-double all-reduce, RCCL against the in-kernel peer sum
# Usage:  tools/multi_gpu_day.sh [round-tag (default r06)] [max GPUs (default: all visible)]
# Runs to completion on ONE GPU (the n = 1 legs) - tools/gpu_r06_*.sh call it there.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R" || exit 1
TAG=${1:-r06}
O=${CCAL_DAY_OUT:-profiles/$TAG/multi_gpu}      # (through gpurun only gpurun_out/ travels back: CCAL_DAY_OUT=gpurun_out/multi_gpu_$TAG, then copy)
mkdir -p "$O"
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
NVIS=$(python3 -c 'import torch; print(torch.cuda.device_count())' 2>/dev/null || echo 0)
NMAX=${2:-$NVIS}
if [ "$NVIS" -lt 1 ]; then echo "multi_gpu_day: no GPU visible" | tee "$O/summary.txt"; exit 4; fi
if [ "$NMAX" -gt "$NVIS" ]; then echo "multi_gpu_day: $NMAX GPUs asked for, $NVIS visible: refusing" | tee "$O/summary.txt"; exit 4; fi
{
echo "multi_gpu_day $TAG: $NVIS GPU(s) visible, using up to $NMAX; $(date -u +%FT%TZ)"
python3 -c 'import torch; [print("  device", i, torch.cuda.get_device_name(i)) for i in range(torch.cuda.device_count())]' 2>/dev/null
} | tee "$O/summary.txt"

# ---- 1. the multi-device tests ------------------------------------------------------------------------------------------------
timeout 1800 python3 -m pytest tests/test_gpu_multi.py tests/test_gpu_dist.py -m gpu -q -x > "$O/pytest_multi.txt" 2>&1
echo "pytest tests/test_gpu_multi.py tests/test_gpu_dist.py: rc $? : $(tail -1 "$O/pytest_multi.txt")" | tee -a "$O/summary.txt"

# ---- 2. bench.py, weak and strong --------------------------------------------------------------------------------------------
for N in 1 2 4 8; do
    if [ "$N" -gt "$NMAX" ]; then echo "bench --gpus $N: skipped ($NMAX GPU(s) in use)" | tee -a "$O/summary.txt"; continue; fi
    timeout 1500 python3 bench.py --gpus "$N" --no-cpu-baseline --no-traffic > "$O/bench_weak_n$N.json" 2> "$O/bench_weak_n$N.err"
    echo "bench --gpus $N (weak, 10 000 frames per GPU): rc $?" | tee -a "$O/summary.txt"
    timeout 1500 python3 bench.py --gpus "$N" --frames-total 50000 --steps 300 --warmup 50 --no-cpu-baseline --no-traffic > "$O/bench_strong50k_n$N.json" 2> "$O/bench_strong50k_n$N.err"
    echo "bench --gpus $N --frames-total 50000 (strong, configs[3]): rc $?" | tee -a "$O/summary.txt"
done
# the refusal path, once: one more GPU than there is
timeout 300 python3 bench.py --gpus $((NVIS + 1)) --steps 2 --warmup 1 > "$O/bench_refused.json" 2> "$O/bench_refused.err"
echo "bench --gpus $((NVIS + 1)) on $NVIS GPU(s): rc $? (4 = refused, no line: $(wc -c < "$O/bench_refused.json") bytes on stdout)" | tee -a "$O/summary.txt"

# ---- 3. one process, every GPU: the two transports of the step's all-reduce ------------------------------------------------------
for TR in inproc rccl; do
    CCAL_MULTI_TRANSPORT=$TR timeout 900 python3 tools/sharded_ab.py "$NMAX" > "$O/sharded_$TR.json" 2> "$O/sharded_$TR.err"
    echo "single-process sharded GN / LM, CCAL_MULTI_TRANSPORT=$TR: rc $? $(head -c 600 "$O/sharded_$TR.json")" | tee -a "$O/summary.txt"
done

# ---- 4. the collective alone ----------------------------------------------------------------------------------------------------
if [ ! -x tools/ubench/allreduce_latency.bin ]; then
    /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/ubench/allreduce_latency.hip -o tools/ubench/allreduce_latency.bin -ldl > "$O/allreduce_build.txt" 2>&1
fi
for N in 1 2 4 8; do
    [ "$N" -gt "$NMAX" ] && continue
    timeout 300 tools/ubench/allreduce_latency.bin "$N" > "$O/allreduce_latency_n$N.txt" 2>&1
    echo "allreduce_latency $N: rc $?" | tee -a "$O/summary.txt"; cat "$O/allreduce_latency_n$N.txt" | tee -a "$O/summary.txt"
done

# ---- the table ---------------------------------------------------------------------------------------------------------------------
python3 - "$O" <<'PY' | tee -a "$O/summary.txt"
import glob, json, os, sys
O = sys.argv[1]
def line(path):
    try:
        for ln in open(path):
            if ln.startswith("{"):
                return json.loads(ln)
    except Exception:
        pass
    return None
print("scaling table (bench.py lines):")
base = {}
for kind in ("weak", "strong50k"):
    for N in (1, 2, 4, 8):
        d = line(os.path.join(O, f"bench_{kind}_n{N}.json"))
        if not d:
            continue
        s = d.get("summary", {})
        if N == 1:
            base[kind] = d["value"]
        eff = d["value"] / (base.get(kind, d["value"]) * (N if kind == "weak" else 1)) if kind == "weak" else d["value"] / base.get(kind, d["value"]) / N
        sh = (d.get("extra") or {}).get("sharded_solve", {})
        print(f"  {kind:10s} n={N}: {d['value']:.4g} evals/s ({eff:.2f} of linear), frac_hbm {d['roofline']['frac']:.3f}, rccl_ranks {d.get('rccl_ranks')}, "
              f"sharded GN {((sh.get('gn') or {}).get('solve_ms'))} ms / LM {((sh.get('lm') or {}).get('solve_ms'))} ms, allreduce_us_per_step {d.get('allreduce_us_per_step')}")
PY
echo "multi_gpu_day: done -> $O" | tee -a "$O/summary.txt"
