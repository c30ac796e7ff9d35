#!/bin/bash
# Runs ON the GPU box: k_inflate forms side by side (kernel time from rocprofv3 --kernel-trace --stats). usage: tools/inflate_ab.sh [members] [genome_len] [forms...]
N=${1:-1536}; L=${2:-5000000}; shift; shift
for f in ${@:-global pipe}; do
  GS_INFLATE_WINDOW=$f bash tools/kstats.sh gpurun_out/_ab.txt python tools/inflate_rate.py $N $L 6 > gpurun_out/_ab.log 2>&1
  echo "GS_INFLATE_WINDOW=$f  $N members x $L bp:"; grep -E "k_inflate|k_crc" gpurun_out/_ab.txt; grep "^rep 2" gpurun_out/_ab.log
done
