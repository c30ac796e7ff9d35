#!/bin/bash
# end-of-round check: the whole GPU suite, smoke, the default bench (profiles/r06_bench_default.log), the 5-step request bench (profiles/r06_bench_request.log)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/final_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/final_suite.txt
timeout 1500 python -u bench.py > gpurun_out/final_bench_default.log 2> gpurun_out/final_bench_default.err
timeout 900 python -u bench.py --steps 5 --warmup 2 --no-extra-legs > gpurun_out/final_bench_request.log 2> gpurun_out/final_bench_request.err
cat gpurun_out/final_suite.txt
python tools/skew_summary.py gpurun_out/final_bench_default.log
python tools/bench_summary.py default < gpurun_out/final_bench_default.log
python tools/bench_summary.py request < gpurun_out/final_bench_request.log
