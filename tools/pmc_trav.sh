T=${1:-r05}; cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RX="k_hnsw_search_dense"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_t1 -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $R/gpurun_out/pmc_t1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_t2 -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $R/gpurun_out/pmc_t2.log 2>&1
cd $R; python tools/pmc_kernel.py k_hnsw_search_dense gpurun_out/pmc_t1 gpurun_out/pmc_t2 > gpurun_out/${T}_trav_pmc.txt 2>&1; tail -3 gpurun_out/pmc_t1.log >> gpurun_out/${T}_trav_pmc.txt; rm -rf gpurun_out/pmc_t1 gpurun_out/pmc_t2; cat gpurun_out/${T}_trav_pmc.txt
