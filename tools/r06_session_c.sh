#!/bin/bash
# round-6 session C: join-block tests + the skewed test, then the bench (skewed leg after the join fixes)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_gpu_join_blocks.py tests/test_gpu_fullsize.py -x -q -s -k "redundant or heavy or skewed or cost_model or config2" 2>&1 | tail -12 ) > gpurun_out/r06c_tests.log 2>&1
cat gpurun_out/r06c_tests.log
( time python bench.py --steps 3 --warmup 1 --prob-db-genomes 0 --ingest-files 0 --c5-proteomes 64 --c5-rows 2048 > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err ) 2>&1 | tail -4
python tools/bench_summary.py r06c < gpurun_out/r06c_bench.json; python -c "import json; j=json.loads([l for l in open(\"gpurun_out/r06c_bench.json\") if l.startswith(\"{\")][-1]); print(json.dumps(j.get(\"request_skewed\"),indent=1)[:3000]); print(json.dumps(j.get(\"request_redundant\"),indent=1)[:700])"
tail -3 gpurun_out/r06c_bench.err
