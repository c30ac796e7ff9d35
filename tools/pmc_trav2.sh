cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
RX="k_hnsw_search_dense"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_t3 -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $R/gpurun_out/pmc_t3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS SQ_IFETCH_LEVEL SQ_BUSY_CU_CYCLES --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_t4 -- python $R/tools/trav_ab.py --genomes --reps 1 "" > $R/gpurun_out/pmc_t4.log 2>&1
cd $R; python tools/pmc_kernel.py k_hnsw_search_dense gpurun_out/pmc_t3 gpurun_out/pmc_t4 > gpurun_out/r02_trav_pmc2.txt 2>&1; rm -rf gpurun_out/pmc_t3 gpurun_out/pmc_t4; cat gpurun_out/r02_trav_pmc2.txt
