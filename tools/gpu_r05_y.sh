#!/bin/bash
# Round 5, call y: randomised device-vs-oracle sweeps on the final library: as shipped, and with k_gram2 forced for every model and size
# (CCAL_GRAM2=1, second library) so that the neighbouring-lane form meets the small, ragged, one-focal, bounded cases of the sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out/r05y; mkdir -p $O
python tools/fuzz_parity.py --seconds 300 --seed 61616 --shards 3 --batch 6 > $O/fuzz_default.json 2> $O/fuzz_default.err; tail -c 600 $O/fuzz_default.json
CCAL_GRAM2=1 python tools/fuzz_parity.py --seconds 300 --seed 71717 > $O/fuzz_gram2.json 2> $O/fuzz_gram2.err; tail -c 600 $O/fuzz_gram2.json
