#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_join_blocks.py tests/test_gpu_fullsize.py -x -q -k "redundant or heavy or skewed" 2>&1 | tail -4 ) > gpurun_out/r06f_tests.log 2>&1
cat gpurun_out/r06f_tests.log
for i in 1 2; do
python bench.py --steps 5 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/r06f_bench_$i.json 2> gpurun_out/r06f_bench.err
python tools/bench_summary.py r06f_$i < gpurun_out/r06f_bench_$i.json
done
