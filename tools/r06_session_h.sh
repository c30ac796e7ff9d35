#!/bin/bash
# round-6 session H: the profile files of the round (prob kernels + counters + rates, join counters f32 and u64, step trace), the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TAG=r06
# ---- prob: per-kernel times, SQ / memory counters, rates over shapes
NOTEST=1 PARTS=4 bash tools/prob_session.sh > /dev/null 2>&1; cp gpurun_out/prob_kstats_tiers_p4.txt gpurun_out/r06_prob_kernel_stats.txt
bash tools/prob_pmc.sh r06 > /dev/null 2>&1
{
for args in "256 5000000 21 18000" "2048 5000000 21 18000" "64 12000000 21 18000" "1024 1300000 21 18000" "16 5000000 21 18000" "512 3000000 16 4096"; do python tools/sketch_rate.py prob $args 2>&1 | tail -1; done
GS_PROB_IMPL=buckets python tools/sketch_rate.py prob 256 5000000 21 18000 2>&1 | tail -1 | sed 's/^/GS_PROB_IMPL=buckets (round 4-5 form): /'
GS_PROB_PROFILE=1 python tools/sketch_rate.py prob 256 5000000 21 18000 2>&1 | grep PROFILE | tail -1
} > gpurun_out/r06_prob_rates.txt 2>&1
# ---- the request step launch by launch; SQ counters of the request-time join
bash tools/step_trace.sh > /dev/null 2>&1
bash tools/pmc_join_headline.sh gpurun_out/r06_join_pmc.txt > /dev/null 2>&1
# ---- the u64 (8-byte key) join of the configs[4] request-at-size leg
bash tools/pmc_any.sh "k_match_join<2" gpurun_out/r06_join_u64_pmc.txt python bench.py --workload c5dist > /dev/null 2>&1
# ---- the GPU suite
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r06_gpu_suite.txt 2>&1
cat gpurun_out/r06_gpu_suite.txt; cat gpurun_out/r06_prob_rates.txt; head -8 gpurun_out/r06_prob_kernel_stats.txt; head -12 gpurun_out/r06_request_step_trace.txt; head -30 gpurun_out/r06_join_u64_pmc.txt
