#!/bin/bash
# the cluster-aware join inside bench.py's request_skewed leg (DNA sketches, 300 k genomes): counters of the instrumented builds (kinds 7 / 8 / 9 of gs_join.hip) and the
# in-place test switched off, one bench run each.  Output: gpurun_out/join_cl_kinds_bench.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B="python -u bench.py --steps 1 --warmup 1 --prob-db-genomes 0 --ingest-files 0 --redundant-roots 0 --c5-proteomes 64 --c5-rows 2000"
{
echo "== product, in-place test off"; GS_JOIN_INPLACE=0 GS_JOIN_TIMES=1 timeout 900 $B > gpurun_out/jb.log 2> gpurun_out/jb.err; python tools/skew_summary.py gpurun_out/jb.log; grep -E "cluster-aware" gpurun_out/jb.err | tail -2
for K in ${KINDS:-7 8 9}; do
  rm -f gsearch_amd/csrc/gs_join.o
  make -s -C gsearch_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DGS_JOIN_COUNT_KIND=$K" ../libgsearch_amd.so > /dev/null 2>&1
  echo "== kind $K"; GS_JOIN_TIMES=1 timeout 900 $B > gpurun_out/jb.log 2> gpurun_out/jb.err; python tools/skew_summary.py gpurun_out/jb.log; grep -E "cluster-aware" gpurun_out/jb.err | tail -1
done
} > gpurun_out/join_cl_kinds_bench.txt 2>&1
cat gpurun_out/join_cl_kinds_bench.txt
