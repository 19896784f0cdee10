#!/bin/bash
# round 5, call h: is the lone launch's ~5 us overhead its tail?  persistent wavefronts striding over frames (every slot gets the
# same number of frames: no ragged last round) against the one-wavefront-per-frame launch; + mode N at 50 000 frames
mkdir -p gpurun_out
{
echo "== mode E eucm 10000 (GB/s)"; python tools/ab_eval.py base,persist5,persist5pf,persist4pf 10000 3
echo "== kb4"; python tools/ab_eval.py base,persist5,persist5pf 10000 2 --model kb4
echo "== two cameras"; python tools/ab_eval.py base,persist5,persist5pf 10000 2 --cams 2
echo "== 1000 / 50000"; python tools/ab_eval.py base,persist5,persist5pf 1000,50000 2
} > gpurun_out/r05h_ab_persist.txt 2>&1
cat gpurun_out/r05h_ab_persist.txt
for F in 2500 10000 20000 50000; do python tools/time_kernels.py --what normal --frames $F --reps 50 | tail -1; done > gpurun_out/r05h_mode_n_sizes.txt 2>&1
cat gpurun_out/r05h_mode_n_sizes.txt
