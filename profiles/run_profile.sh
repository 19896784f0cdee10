#!/bin/bash
# Produces the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun):
#   gpurun --timeout 1500 -- 'bash profiles/run_profile.sh r01'
# Counters are collected in their own passes (kernel-trace only), as MI355X_MICROARCH.md prescribes.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 50 --warmup 5 --no-cpu-baseline"
timeout 600 python3 $REPO/bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
# the headline command alone (--no-extra): k_eval's average here is the figure that must agree with roofline.kernel_ms (the full
# command below also launches k_eval on half problems - extra.multi_eval - which would drag the average down)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_headline -o stats -- $BENCH --no-extra --no-traffic > $OUT/stats_headline_bench.json 2> $OUT/stats_headline.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats_bench.json 2> $OUT/stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH --no-extra > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH --no-extra > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
# (the SQ passes leave the side-by-side leg out: its 6 000 extra dispatches take the counter file past what gpurun carries back)
export CCAL_BENCH_NO_CONCURRENT=1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
# memory side of the headline kernel (the counter set of round 2's cliff study; GRBM_GUI_ACTIVE / TCC_BUSY beside these made the
# profiler abort inside hipMalloc and hang until the outer limit in round 4 - every pass now runs under its own timeout)
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ GRBM_EA_BUSY GRBM_TC_BUSY --output-format csv -d $OUT/pmc_mem -o pmc -- $BENCH --no-extra > $OUT/pmc_mem_bench.json 2> $OUT/pmc_mem.err
# mode N: issue / wait breakdown of the Gram and elimination kernels (second SQ pass: 8 SQ slots per pass)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
unset CCAL_BENCH_NO_CONCURRENT
# the two-camera rig (BASELINE configs[4] shape: 2 x 10 000 frames, EUCM): mode E, normal-equation build, GN and LM solves
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -o stats -- python3 $REPO/tools/time_kernels.py --what eval,normal,solve --cams 2 --reps 50 > $OUT/two_camera.json 2> $OUT/stats2.err
# the other models of BASELINE configs[2] at the headline size: mode E + build + solves
for m in kb4 opencv5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$m -o stats -- python3 $REPO/tools/time_kernels.py --what eval,normal,solve --model $m --reps 50 > $OUT/model_$m.json 2> $OUT/stats_$m.err
done
ls -R $OUT | head -80
find $OUT -name "*.csv" -size +20M -delete
