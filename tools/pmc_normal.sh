cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > $R/gpurun_out/sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_MISC SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64"; do
  n=$(echo $set | cut -c1-20 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcn_$n -o p -- python3 $R/tools/time_kernels.py --what ${WHAT:-normal} --reps 5 ${EXTRA:-} > /dev/null 2>$R/gpurun_out/pmcn_$n.err
done
python3 - <<PY
import csv, collections, glob
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmcn_*/p_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'].split('(')[0]
        if any(t in k for t in ('k_gram', 'k_schur', 'k_solve', 'k_head', 'k_eval', 'k_backsub')): d[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in d.items():
    print(k)
    for c,x in sorted(v.items()): print('   %-28s %14.0f' % (c, sum(x)/len(x)))
PY
