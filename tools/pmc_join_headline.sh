#!/bin/bash
# Runs ON the GPU box: SQ counters of the request-time k_match_join on the headline data (tools/trav_ab.py --genomes: 300 k sketched genomes, 10 000 queries)
# usage: tools/pmc_join_headline.sh <out.txt>
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex k_match_join -d $R/gpurun_out/pmc_jh1 -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $R/gpurun_out/pmc_jh1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-include-regex k_match_join -d $R/gpurun_out/pmc_jh2 -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $R/gpurun_out/pmc_jh2.log 2>&1
cd $R; python tools/pmc_kernel.py "k_match_join<3, unsigned int, 1" gpurun_out/pmc_jh1 gpurun_out/pmc_jh2 > $1 2>&1; grep -E "join" gpurun_out/pmc_jh1.log | cut -c1-120 >> $1; rm -rf gpurun_out/pmc_jh1 gpurun_out/pmc_jh2; cat $1
