#!/bin/bash
# Runs ON the GPU box (through gpurun, from the repo root): kernel stats + HBM traffic counters of the bench command, then the bench itself with the
# sidecar so that roofline.traffic is measured by THIS session. Counters are collected in their own passes (--kernel-trace only, never with other
# trace domains). usage: tools/pmc_bench.sh [round tag, default r05]
# Outputs under gpurun_out/: <tag>_kernel_stats.csv, <tag>_pmc_sidecar.json, <tag>_pmc_raw.csv, <tag>_bench_request.log, <tag>_bench_sketch.log
T=${1:-r05}
R=$(pwd)
export TMPDIR=/tmp
export GS_HEAD=$(cat $R/.head 2>/dev/null)
RX='k_hnsw_search|k_match_join|k_sketch_min|k_hamming_qxc'
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o $T -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/${T}_bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE --kernel-include-regex "$RX" -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > /dev/null 2>&1
# configs[4] distance leg (u64 row gather): the same launch the bench's extra leg times
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE --kernel-include-regex 'k_hnsw_search<2>' -d $R/gpurun_out/pmc_c5f -- python $R/bench.py --workload c5dist > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE --kernel-include-regex 'k_hnsw_search<2>' -d $R/gpurun_out/pmc_c5w -- python $R/bench.py --workload c5dist > /dev/null 2>&1
cd $R
python tools/pmc_condense.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${T}_pmc_sidecar.json gpurun_out/${T}_pmc_raw.csv gpurun_out/pmc_c5f gpurun_out/pmc_c5w > gpurun_out/${T}_pmc_condense.log 2>&1
S=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && cp "$S" gpurun_out/${T}_kernel_stats.csv
rm -rf gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_c5f gpurun_out/pmc_c5w
GS_PMC_SIDECAR=$R/gpurun_out/${T}_pmc_sidecar.json python bench.py --steps 5 --warmup 2 > gpurun_out/${T}_bench_request.log 2> gpurun_out/${T}_bench_request.err
python bench.py --workload sketch --steps 3 --warmup 1 > gpurun_out/${T}_bench_sketch.log 2>> gpurun_out/${T}_bench_request.err
tail -c 400 gpurun_out/${T}_pmc_condense.log; head -12 gpurun_out/${T}_kernel_stats.csv | cut -c1-160
