#!/bin/bash
# tiered prob form: sweep of the slices per genome (GS_PROB_PARTS) with per-kernel times
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for parts in ${PARTS:-4 8 16 32}; do
  echo "== parts $parts"
  GS_PROB_PARTS=$parts GS_PROB_PROFILE=1 bash tools/kstats.sh gpurun_out/prob_kstats_p$parts.txt python tools/sketch_rate.py prob ${1:-256} 5000000 21 18000 2>&1 | grep -E "tiers:|k-mers/s|k_prob_tiers|k_prob_part1|k_prob_buckets" | tail -6
done
} > gpurun_out/prob_session2.log 2>&1
cat gpurun_out/prob_session2.log
