#!/bin/bash
# Runs ON the GPU box (through gpurun, from the repo root): rocprofv3 PC sampling of one genome-level 10 000-query request
# (tools/trav_ab.py --genomes), condensed to a histogram of sampled PCs of k_hnsw_search_dense (tools/pcsample_condense.py).
# The PCs are code-object offsets: tools/pcsample_bin.py joins them with the disassembly of the same libgsearch_amd.so.
# usage: pcsample_trav.sh [stochastic|host_trap] [interval]   (stochastic: cycles, power of two; host_trap: microseconds)
R=$(pwd); export TMPDIR=/tmp; cd /tmp
METHOD=${1:-stochastic}; IV=${2:-65536}
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
OUT=$R/gpurun_out/pcs_$METHOD
rm -rf $OUT; mkdir -p $OUT
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 900 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $IV \
    --output-format csv json -d $OUT/raw -o pcs -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $OUT/run.log 2>&1
echo "rocprofv3 rc=$?" >> $OUT/run.log
cd $R
find $OUT/raw -type f | xargs ls -la > $OUT/files.txt 2>&1
for f in $(find $OUT/raw -name "*.csv"); do echo "== $f"; head -5 "$f" | cut -c1-600; done > $OUT/heads.txt 2>&1
python tools/pcsample_condense.py $OUT/raw k_hnsw_search_dense $OUT/hist.csv > $OUT/condense.log 2>&1
rm -rf $OUT/raw
tail -5 $OUT/run.log; cat $OUT/condense.log | tail -20
