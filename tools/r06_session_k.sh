#!/bin/bash
# round-6 closing session: PMC sidecar + kernel stats + bench at HEAD, step trace, join counters, the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/pmc_bench.sh r06 > /dev/null 2>&1
bash tools/step_trace.sh > /dev/null 2>&1
bash tools/pmc_join_headline.sh gpurun_out/r06_join_pmc.txt > /dev/null 2>&1
bash tools/pmc_any.sh "k_match_join<2" gpurun_out/r06_join_u64_pmc.txt python bench.py --workload c5dist > /dev/null 2>&1
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r06_gpu_suite.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/r06_gpu_suite.txt
cat gpurun_out/r06_gpu_suite.txt
python tools/bench_summary.py r06 < gpurun_out/r06_bench_request.log
head -12 gpurun_out/r06_request_step_trace.txt
grep -E "INSTS_VALU|INSTS_SALU" gpurun_out/r06_join_u64_pmc.txt gpurun_out/r06_join_pmc.txt
