#!/bin/bash
# prob session on one box: parity tests of the prob forms, then rate + per-kernel times of the tiered form (and of the bucketed one with IMPLS="tiers buckets")
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
[ -z "$NOTEST" ] && timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "prob" 2>&1 | tail -15
for impl in ${IMPLS:-tiers}; do
  for parts in ${PARTS:-4}; do
  echo "== GS_PROB_IMPL=$impl parts=$parts"
  GS_PROB_PARTS=$parts GS_PROB_IMPL=$impl GS_PROB_PROFILE=1 bash tools/kstats.sh gpurun_out/prob_kstats_${impl}_p$parts.txt python tools/sketch_rate.py prob ${1:-256} ${2:-5000000} 21 18000 2>&1 | grep -E "PROFILE|k-mers/s|k_prob|GS_PROB" | tail -12
  done
done
} > gpurun_out/prob_session.log 2>&1
tail -60 gpurun_out/prob_session.log
