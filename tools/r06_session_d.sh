#!/bin/bash
# round-6 session D: new API tests, bench --shard db at one rank, the full GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "release_build or comm_allgather" 2>&1 | tail -8 ) > gpurun_out/r06d_tests.log 2>&1
cat gpurun_out/r06d_tests.log
( time python bench.py --steps 3 --warmup 1 --shard db --no-extra-legs > gpurun_out/r06d_bench_db.json 2> gpurun_out/r06d_bench_db.err ) 2>&1 | tail -4
python tools/bench_summary.py r06d_db < gpurun_out/r06d_bench_db.json; python -c "import json; j=json.loads([l for l in open(\"gpurun_out/r06d_bench_db.json\") if l.startswith(\"{\")][-1]); print(j[\"scaling\"], j.get(\"db_shard\"), j[\"multi_gpu_check\"][\"sharding\"], j.get(\"parity_checked\"))"
tail -3 gpurun_out/r06d_bench_db.err
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r06d_gpu_suite.log 2>&1
cat gpurun_out/r06d_gpu_suite.log
