# Developer tool: PMC passes of mode E below and above the ~1.35 GB cliff (gpurun --timeout 900 -- 'bash tools/pmc_eval_cliff.sh')
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for F in ${FRAMES:-44000 56000}; do
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ GRBM_EA_BUSY GRBM_TC_BUSY" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ TCC_BUSY TCC_TAG_STALL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_WRREQ_LEVEL TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmce_${F}_$i -o p -- python3 $R/tools/time_kernels.py --what eval --frames $F --reps 4 > /dev/null 2>$R/gpurun_out/pmce_${F}_$i.err
done
done
python3 - <<PY
import csv, collections, glob, re
d=collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
n=collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$R/gpurun_out/pmce_*/p_counter_collection.csv"):
    F=re.search(r"pmce_(\d+)_", f).group(1)
    for row in csv.DictReader(open(f)):
        if 'k_eval' not in row['Kernel_Name']: continue
        d[F][row['Counter_Name']][row['Dispatch_Id']] += float(row['Counter_Value'])
for F in sorted(d):
    print("frames", F)
    for c in sorted(d[F]):
        v=list(d[F][c].values())
        print('   %-44s %16.0f  (n=%d)' % (c, sum(v)/len(v), len(v)))
PY
find $R/gpurun_out -name "p_counter_collection.csv" -size +5M -delete
