#!/bin/bash
# closing profiles of round 6 at HEAD: PMC sidecar + kernel stats + 5-step bench with this session's sidecar, step trace, headline join counters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/pmc_bench.sh r06 > /dev/null 2>&1
bash tools/step_trace.sh > /dev/null 2>&1
bash tools/pmc_join_headline.sh gpurun_out/r06_join_pmc.txt > /dev/null 2>&1
python tools/bench_summary.py r06 < gpurun_out/r06_bench_request.log
head -12 gpurun_out/r06_request_step_trace.txt
grep -E "INSTS_VALU|INSTS_SALU|FETCH|WRITE" gpurun_out/r06_join_pmc.txt | head
