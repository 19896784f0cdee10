for m in kb4 opencv5; do for of in "" "--one-focal"; do for g in valu mfma; do
  echo -n "$m $of CCAL_GRAM=$g: "; CCAL_GRAM=$g python tools/time_kernels.py --what normal --model $m $of --reps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['normal_us'],1))"
done; done; done
for lpf in 8 12 16; do echo -n "kb4 valu lpf $lpf: "; CCAL_GRAMV_LPF=$lpf CCAL_GRAM=valu python tools/time_kernels.py --what normal --model kb4 --reps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['normal_us'],1))"; done
for lpf in 8 12 16; do echo -n "opencv5 valu lpf $lpf: "; CCAL_GRAMV_LPF=$lpf CCAL_GRAM=valu python tools/time_kernels.py --what normal --model opencv5 --reps 200 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['normal_us'],1))"; done
