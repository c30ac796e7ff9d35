#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -x -q -m gpu -k "u64 or uint64 or join or count_matrix or prob_tohnsw or config5 or cost_model or redundant or index_dump" 2>&1 | tail -5 ) > gpurun_out/r06i_tests.log 2>&1
cat gpurun_out/r06i_tests.log
python bench.py --workload c5dist 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r=j.get('request_at_size') or j.get('distance_gather_u64_m24000',{}).get('request_at_size') or j
print(json.dumps({k:r[k] for k in r if k in ('call_ms','count_matrix_kernels_ms','traversal_kernel_ms','join_atomics','ids_distances_evals_equal_oracle_16_queries','queries_per_sec')},indent=0))
print(list(j.keys())[:12])"
