#!/bin/bash
# tiered prob form: rate and per-kernel times over GS_PROB_PT_AVG x GS_PROB_PARTS (256 x 5 Mbp)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for avg in ${AVGS:-4096 8192}; do for parts in ${PARTS:-4}; do
  echo "== PT_AVG=$avg parts=$parts"
  GS_PROB_PT_AVG=$avg GS_PROB_PARTS=$parts GS_PROB_PROFILE=1 python tools/sketch_rate.py prob ${1:-256} ${2:-5000000} 21 18000 2>&1 | tail -2
  GS_PROB_PT_AVG=$avg GS_PROB_PARTS=$parts bash tools/kstats.sh gpurun_out/prob_kstats_a${avg}_p$parts.txt python tools/sketch_rate.py prob ${1:-256} ${2:-5000000} 21 18000 2>&1 | grep -E "k_prob_tiers|k_prob_part1|k_prob_buckets"
done; done
} > gpurun_out/prob_sweep.log 2>&1
cat gpurun_out/prob_sweep.log
