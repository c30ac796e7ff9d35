#!/bin/bash
# host-side thresholds of the tiers filter: parity, rate, kernel times, a short fuzz
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "prob" 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python -u tools/sketch_rate.py prob 256 5000000 21 18000 2>&1 | tail -1; done
timeout 300 python -u tools/sketch_rate.py prob 2048 5000000 21 18000 2>&1 | tail -1
timeout 300 bash tools/kstats.sh gpurun_out/prob_kstats_l.txt python tools/sketch_rate.py prob 256 5000000 21 18000 2>&1 | grep -E "k_prob"
timeout 600 python -u tools/prob_fuzz.py 40 777 prob 2>&1 | tail -3
} > gpurun_out/session_l.log 2>&1
cat gpurun_out/session_l.log
