#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_join_blocks.py -q -x 2>&1 | tail -4
GS_JOIN_TIMES=1 timeout 600 python -u tools/skew_probe.py 100000 2500 2>&1 | grep -E "^rep 2|cluster-aware|block compare|^ok" | tail -4
GS_JOIN_TIMES=1 GS_JOIN_VERBOSE=1 timeout 1500 python -u bench.py --steps 2 --warmup 1 --prob-db-genomes 0 --ingest-files 0 > gpurun_out/bench_o.log 2> gpurun_out/bench_o.err; python tools/skew_summary.py; grep -E "cluster-aware|block compare" gpurun_out/bench_o.err | tail -4
} > gpurun_out/session_m.log 2>&1
cat gpurun_out/session_m.log
