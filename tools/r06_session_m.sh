#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_join_blocks.py -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "hamming or dist" 2>&1 | tail -3
GS_JOIN_TIMES=1 GS_JOIN_VERBOSE=1 timeout 600 python -u tools/skew_probe.py 100000 2500 2>&1 | grep -E "^rep|cluster-aware|block compare|^ok|clusters," | tail -6
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -k "skewed or redundant or cost_model" 2>&1 | tail -3
} > gpurun_out/session_m.log 2>&1
cat gpurun_out/session_m.log
