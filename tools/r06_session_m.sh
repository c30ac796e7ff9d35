#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_join_blocks.py -q -x 2>&1 | tail -8
timeout 600 bash tools/kstats.sh gpurun_out/skew_kstats.txt python tools/skew_probe.py 100000 2500 2>&1 | head -8
GS_BLOCKS_THIN_OFF=1 timeout 600 bash tools/kstats.sh gpurun_out/skew_kstats_off.txt python tools/skew_probe.py 100000 2500 2>&1 | head -8
} > gpurun_out/session_m.log 2>&1
cat gpurun_out/session_m.log
