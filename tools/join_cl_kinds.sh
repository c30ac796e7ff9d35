#!/bin/bash
# the cluster-aware join on a skewed database (tools/skew_probe.py, 2500 x 100 k): how many values reach the probe (kind 7), end at the in-place own-cluster test (8), are
# dropped as own-cluster hits by the probe (9); kind 0 = the product (atomics). Rebuilds gs_join.o per kind on the box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for K in ${KINDS:-0 7 8 9}; do
  rm -f gsearch_amd/csrc/gs_join.o
  make -s -C gsearch_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DGS_JOIN_COUNT_KIND=$K" ../libgsearch_amd.so > /dev/null 2>&1
  echo "kind $K: $(GS_JOIN_TIMES=1 timeout 600 python -u tools/skew_probe.py 100000 2500 2>&1 | grep -E '^rep 2|cluster-aware' | tail -2 | tr '\n' ' ')"
done
} > gpurun_out/join_cl_kinds.txt 2>&1
cat gpurun_out/join_cl_kinds.txt
