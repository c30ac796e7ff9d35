#!/bin/bash
# SQ / memory counters of the two kernels of the tiered prob form (256 genomes x 5 Mbp, k = 21, s = 18000); usage: tools/prob_pmc.sh <tag>
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export GS_PROB_PARTS=${PARTS:-4}
GS_PROB_PROFILE=1 python tools/sketch_rate.py prob 256 5000000 21 18000 2>&1 | tail -4
bash tools/pmc_any.sh "k_prob_tier_filter|k_prob_tier_points<64|k_prob_part1" gpurun_out/${1:-r06}_prob_pmc.txt python tools/sketch_rate.py prob 256 5000000 21 18000 > /dev/null 2>&1
grep -E "INSTS|ACTIVE_INST_VALU|BUSY_CYC|WAIT_ANY|WAIT_INST_ANY|BANK|SIZE" gpurun_out/${1:-r06}_prob_pmc.txt
