#!/bin/bash
# round-4 GPU pass B: k_schurq variants (slots per wavefront, slot stride), session-size solves after the zero-copy change,
# memory-side counters of the mode-E kernels (12 / 14 / 18 columns)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
python tools/ab_build.py "s16:CCAL_SCHURQ_SLOTS=16,s8:CCAL_SCHURQ_SLOTS=8,old16@oldss:CCAL_SCHURQ_SLOTS=16,old8@oldss:CCAL_SCHURQ_SLOTS=8" eucm 10000 3 --cams 2 > $O/ab_schurq.txt 2>&1
python tools/ab_build.py "s16:CCAL_SCHURQ_SLOTS=16,s8:CCAL_SCHURQ_SLOTS=8" ucm 10000 2 --cams 2 --one-focal >> $O/ab_schurq.txt 2>&1
for v in s16 s8; do
  n=16; [ $v = s8 ] && n=8
  CCAL_SCHURQ_SLOTS=$n bash tools/kstats.sh --what normal --cams 2 --reps 50 > $O/kstats_$v.txt 2>&1
done
CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_oldss.so CCAL_SCHURQ_SLOTS=16 bash tools/kstats.sh --what normal --cams 2 --reps 50 > $O/kstats_old16.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for v in new16 new8 old16; do
  case $v in new16) E="CCAL_SCHURQ_SLOTS=16";; new8) E="CCAL_SCHURQ_SLOTS=8";; old16) E="CCAL_SCHURQ_SLOTS=16 CCAL_LIB=$R/camera_intrinsic_calibration_rs_amd/lib/variants/libccal_oldss.so";; esac
  env $E true
  export $E
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_LDS --output-format csv -d $R/$O/pmc_sq_$v -o p -- python3 $R/tools/time_kernels.py --what normal --cams 2 --reps 5 > /dev/null 2> $R/$O/pmc_sq_$v.err
  unset CCAL_LIB CCAL_SCHURQ_SLOTS
done
# mode E: which resource holds the wide blocks (verdict r03 task 6)
for cfg in "eucm 1" "kb4 1" "opencv5 1" "eucm 2"; do
  set -- $cfg
  i=0
  for cs in "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ GRBM_GUI_ACTIVE TCC_BUSY" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_HIT TCC_MISS TCC_REQ TCP_PENDING_STALL_CYCLES"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $R/$O/pmce_$1_$2_$i -o p -- python3 $R/tools/time_kernels.py --what eval --model $1 --cams $2 --reps 6 > /dev/null 2> $R/$O/pmce_$1_$2_$i.err
  done
done
cd $R
python3 - <<PY > $O/pmc_tables.txt
import csv, collections, glob, re, os
def table(pattern, want):
    d=collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    dur=collections.defaultdict(dict)
    for f in sorted(glob.glob(pattern)):
        tag=re.search(r"pmc[e_a-z]*_(.*?)/p_counter", f).group(1)
        tag=re.sub(r"_\d$","",tag)
        for row in csv.DictReader(open(f)):
            k=row['Kernel_Name'].split('(')[0].replace('void ','')
            if not any(w in k for w in want): continue
            d[(tag,k)][row['Counter_Name']][row['Dispatch_Id']+f] += float(row['Counter_Value'])
            dur[(tag,k)][row['Dispatch_Id']+f]=float(row['End_Timestamp'])-float(row['Start_Timestamp'])
    for key in sorted(d):
        ds=sorted(dur[key].values()); med=ds[len(ds)//2]
        print(key, "median ns", med, "n", len(ds))
        for c in sorted(d[key]):
            v=[x for k2,x in d[key][c].items() if dur[key][k2]>=0.5*med]
            print('   %-36s %16.0f' % (c, sum(v)/len(v)))
table("$O/pmc_sq_*/p_counter_collection.csv", ("k_schurq","k_gram2"))
table("$O/pmce_*/p_counter_collection.csv", ("k_eval",))
PY
python bench.py > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" -size +3M -delete
cat $O/pytest.log; cat $O/ab_schurq.txt; head -6 $O/kstats_s16.txt; head -6 $O/kstats_s8.txt; head -6 $O/kstats_old16.txt
