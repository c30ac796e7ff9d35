#!/bin/bash
# final round-6 session: the whole GPU suite, smoke(), the default bench with the committed sidecar
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 ) > gpurun_out/r06_gpu_suite.txt 2>&1
cat gpurun_out/r06_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time python bench.py > gpurun_out/r06_bench_default.log 2> gpurun_out/r06_bench_default.err ) 2>&1 | tail -3
python tools/bench_summary.py r06_default < gpurun_out/r06_bench_default.log
python -c "
import json
j=json.loads([l for l in open('gpurun_out/r06_bench_default.log') if l.startswith('{')][-1])
e=j['extra_legs']; rp=e['request_prob']; rs=j['request_skewed']
print('prob', e['other_sketchers_k21_s18000']['prob']['kmers_per_sec'], 'request_prob ms', rp['ms_per_10000_queries'], rp['steps'][0], rp['ids_distances_evals_equal_oracle_16_queries'], rp['prob_sketch_bit_exact_vs_oracle_2_query_genomes'])
print('skewed', rs['ms_per_step'], rs['ms_per_step_over_headline'], rs['ids_distances_evals_equal_oracle_16_queries'])
print('c5', e['config4_aa_super2']['kmers_per_sec'], e['config4_aa_super2']['distance_gather_u64_m24000']['request_at_size']['call_ms'])
print('redundant', j['request_redundant']['ms_per_step'], 'ingest', e.get('ingest_gz_files',{}).get('genomes_per_sec_file_inclusive'))
print(j['roofline']['frac'], j['roofline']['traffic_source'])
"
