#!/bin/bash
# timing experiment: the cluster-aware join with the node-label look-up at hit time compiled out (-DGS_JOIN_FAKE_LABEL: wrong counts) against the product, on tools/skew_probe.py
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== product"; GS_JOIN_TIMES=1 timeout 600 python -u tools/skew_probe.py 100000 2500 2>&1 | grep -v "^$" | tail -12
rm -f gsearch_amd/csrc/gs_join.o
make -s -C gsearch_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -DGS_JOIN_FAKE_LABEL=1" ../libgsearch_amd.so > /dev/null 2>&1
echo "== label look-up compiled out"; GS_JOIN_TIMES=1 timeout 600 python -u tools/skew_probe.py 100000 2500 2>&1 | grep -v "^$" | tail -12
} > gpurun_out/join_fake_label.txt 2>&1
cat gpurun_out/join_fake_label.txt
