#!/bin/bash
# round-6 session A: the new prob tests, then the default bench with the request_prob leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "tiered_form or worker_table" 2>&1 | tail -15 ) > gpurun_out/r06a_tests.log 2>&1
( time timeout 1800 python -m pytest tests/test_gpu_fullsize.py -x -q -k "prob_tohnsw" 2>&1 | tail -25 ) >> gpurun_out/r06a_tests.log 2>&1
cat gpurun_out/r06a_tests.log
( time python bench.py --steps 3 --warmup 1 > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err ) 2>&1 | tail -4
python tools/bench_summary.py r06a < gpurun_out/r06a_bench.json; python -c "import json; j=json.loads([l for l in open(\"gpurun_out/r06a_bench.json\") if l.startswith(\"{\")][-1]); e=j.get(\"extra_legs\",{}); print(json.dumps(e.get(\"request_prob\"),indent=1)[:3000]); print(json.dumps(e.get(\"other_sketchers_k21_s18000\"),indent=1)[:1500])"
tail -5 gpurun_out/r06a_bench.err
