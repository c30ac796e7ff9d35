#!/bin/bash
# what the match-join's memory-side atomics are made of at the headline shape: rebuilds gs_join.o with GS_JOIN_COUNT_KIND = 1..6 (gs_join.hip) on the box and reads
# `atomics_per_launch` of one bench step each.  1: sends past a proven accumulator (a second related query, or a chance match on a related node), 2: evictions of a
# run-of-one accumulator, 3 / 4: end-of-block flushes with run >= 2 / run 1, 5: all hits, 6: hits absorbed by the accumulator.  Output: gpurun_out/join_kinds.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for K in ${KINDS:-0 1 2 3 4 5 6}; do
  rm -f gsearch_amd/csrc/gs_join.o
  make -s -C gsearch_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DGS_JOIN_COUNT_KIND=$K" ../libgsearch_amd.so > /dev/null 2>&1
  timeout 600 python -u bench.py --steps 1 --warmup 1 --no-extra-legs --no-cpu-baseline > gpurun_out/jk_$K.log 2>&1
  python - "$K" <<'PY'
import json, sys
K = sys.argv[1]
for l in open("gpurun_out/jk_%s.log" % K):
    if l.startswith("{"):
        d = json.loads(l); k = d["kernels"][0]
        print("kind %s: %.4e per batch launch (%d launches, %.2f ms avg)" % (K, k["atomics_per_launch"], k["launches"], k["avg_launch_ms"]))
PY
done
} > gpurun_out/join_kinds.txt 2>&1
cat gpurun_out/join_kinds.txt
