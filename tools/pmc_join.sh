#!/bin/bash
# Runs ON the GPU box: SQ counters of k_match_join alone (tools/join_probe.py: 100 k family-structured rows, 2500 queries, search-time joins only)
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex k_match_join -d $R/gpurun_out/pmc_j1 -- python $R/tools/join_probe.py 100000 2500 18000 100 > $R/gpurun_out/pmc_j1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC GRBM_GUI_ACTIVE --kernel-include-regex k_match_join -d $R/gpurun_out/pmc_j2 -- python $R/tools/join_probe.py 100000 2500 18000 100 > $R/gpurun_out/pmc_j2.log 2>&1
cd $R; python tools/pmc_kernel.py k_match_join gpurun_out/pmc_j1 gpurun_out/pmc_j2 > gpurun_out/r02_join_pmc.txt 2>&1; tail -3 gpurun_out/pmc_j1.log >> gpurun_out/r02_join_pmc.txt; rm -rf gpurun_out/pmc_j1 gpurun_out/pmc_j2; cat gpurun_out/r02_join_pmc.txt
