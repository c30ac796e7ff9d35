#!/bin/bash
# Runs ON the GPU box (through gpurun, from the repo root): kernel stats + HBM traffic counters of the bench command, then the bench itself with the
# sidecar so that roofline.traffic is measured by THIS session. Counters are collected in their own passes (--kernel-trace only).
# Outputs under gpurun_out/: r03_kernel_stats.csv, r03_pmc_sidecar.json, r03_pmc_raw.csv, r03_bench_request.log, r03_bench_sketch.log
R=$(pwd)
export TMPDIR=/tmp
RX='k_hnsw_search|k_match_join|k_sketch_min'
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r03 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/r03_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
cd $R
python tools/pmc_condense.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/r03_pmc_sidecar.json gpurun_out/r03_pmc_raw.csv > gpurun_out/r03_pmc_condense.log 2>&1
S=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" gpurun_out/r03_kernel_stats.csv; find gpurun_out/prof_stats gpurun_out/pmc_fetch -type f | head -8 > gpurun_out/r03_prof_files.txt
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write
GS_PMC_SIDECAR=$R/gpurun_out/r03_pmc_sidecar.json python bench.py --steps 5 --warmup 2 > gpurun_out/r03_bench_request.log 2> gpurun_out/r03_bench_request.err
python bench.py --workload sketch --steps 3 --warmup 1 > gpurun_out/r03_bench_sketch.log 2>> gpurun_out/r03_bench_request.err
tail -c 400 gpurun_out/r03_pmc_condense.log; head -12 gpurun_out/r03_kernel_stats.csv | cut -c1-160
