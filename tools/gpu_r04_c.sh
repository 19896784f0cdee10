#!/bin/bash
# round-4 GPU pass C (every step under its own timeout): k_schurq with column ownership (timing + LDS counters), the pre-pass
# experiment, session-size solves, then the memory-side counters of mode E with the counter sets that worked in round 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
timeout 300 python tools/ab_build.py "new,rows@rows,oldss@oldss" eucm 10000 3 --cams 2 > $O/ab_schurq.txt 2>&1
timeout 300 python tools/ab_build.py "base,pre:CCAL_PREPASS=1" eucm 10000,625 3 > $O/ab_prepass.txt 2>&1
timeout 200 python tools/ab_build.py "base,pre:CCAL_PREPASS=1" ucm 10000 2 >> $O/ab_prepass.txt 2>&1
timeout 120 bash tools/kstats.sh --what normal --cams 2 --reps 50 > $O/kstats_schurq.txt 2>&1
CCAL_PREPASS=1 timeout 120 bash tools/kstats.sh --what normal,solve --reps 50 > $O/kstats_prepass.txt 2>&1
timeout 120 bash tools/kstats.sh --what normal,solve --reps 50 > $O/kstats_base.txt 2>&1
timeout 200 python tools/concurrent_sessions.py 625 eucm > $O/concurrent.json 2> $O/concurrent.err
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_LDS --output-format csv -d $R/$O/pmc_sq_cols -o p -- python3 $R/tools/time_kernels.py --what normal --cams 2 --reps 5 > /dev/null 2> $R/$O/pmc_sq_cols.err
for cfg in "eucm 1" "kb4 1" "eucm 2"; do
  set -- $cfg
  i=0
  for cs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ GRBM_EA_BUSY GRBM_TC_BUSY"; do
    i=$((i+1))
    timeout 150 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $R/$O/pmce_$1_$2_$i -o p -- python3 $R/tools/time_kernels.py --what eval --model $1 --cams $2 --reps 6 > /dev/null 2> $R/$O/pmce_$1_$2_$i.err
    echo "pmce $1 $2 $i rc $?" >> $R/$O/pmce_rc.txt
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
table("$O/pmc_sq_*/p_counter_collection.csv", ("k_schurq",))
table("$O/pmce_*/p_counter_collection.csv", ("k_eval",))
PY
timeout 200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
find $O -name "*.csv" -size +3M -delete
cat $O/pytest.log $O/ab_schurq.txt $O/ab_prepass.txt; head -7 $O/kstats_schurq.txt; head -9 $O/kstats_prepass.txt; head -9 $O/kstats_base.txt; cat $O/pmce_rc.txt; head -c 1500 $O/concurrent.json
