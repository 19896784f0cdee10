#!/bin/bash
# Round 6, call r: SQ counters of the Gram launches (uniform 10 000 frames EUCM / KB4, ragged EUCM): is the kernel VALU-issue bound end to end?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r06r; mkdir -p $O
bash tools/pmc_normal.sh > $O/pmc_gram_eucm.txt 2>&1; rm -rf gpurun_out/pmcn_*
EXTRA="--model kb4" bash tools/pmc_normal.sh > $O/pmc_gram_kb4.txt 2>&1; rm -rf gpurun_out/pmcn_*
EXTRA="--ragged" bash tools/pmc_normal.sh > $O/pmc_gram_eucm_ragged.txt 2>&1; rm -rf gpurun_out/pmcn_*
cat $O/pmc_gram_eucm.txt $O/pmc_gram_kb4.txt $O/pmc_gram_eucm_ragged.txt | grep -v "^$" | head -120
