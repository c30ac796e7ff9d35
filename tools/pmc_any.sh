#!/bin/bash
# Runs ON the GPU box: SQ / memory counters of one kernel (regex) for any command, in separate passes, condensed by tools/pmc_kernel.py.
# usage: tools/pmc_any.sh <kernel regex> <out.txt> <command ...>
R=$(pwd); RX=$1; OUT=$2; shift 2; export TMPDIR=/tmp; cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $C --kernel-include-regex "$RX" -d $R/gpurun_out/pa_$i -- bash -c 'cd "$0" && exec "$@"' "$R" "$@" > $R/gpurun_out/pa_$i.log 2>&1
done
cd $R
python tools/pmc_kernel.py "$RX" gpurun_out/pa_1 gpurun_out/pa_2 gpurun_out/pa_3 gpurun_out/pa_4 > $OUT 2>&1
rm -rf gpurun_out/pa_*; cat $OUT
