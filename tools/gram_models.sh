# Developer tool: whole normal-equation build for KB4 / OPENCV5 against lanes per frame (and the matrix-core kernel)
for m in kb4 opencv5; do for lpf in 6 8 12 16; do
  echo -n "$m lpf $lpf: "; CCAL_GRAMV_LPF=$lpf python tools/time_kernels.py --what normal --model $m --reps 200 ${EXTRA:-} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['normal_us'],1))"
done; done
