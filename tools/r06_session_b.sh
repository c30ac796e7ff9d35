#!/bin/bash
# round-6 session B: the configs[4] bench-shape test, the skewed-database test, then the default bench with every leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "config4_sketch or skewed" 2>&1 | tail -25 ) > gpurun_out/r06b_tests.log 2>&1
cat gpurun_out/r06b_tests.log
( time python bench.py --steps 3 --warmup 1 > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err ) 2>&1 | tail -4
python tools/bench_summary.py r06b < gpurun_out/r06b_bench.json; python -c "import json; j=json.loads([l for l in open(\"gpurun_out/r06b_bench.json\") if l.startswith(\"{\")][-1]); e=j.get(\"extra_legs\",{}); print(json.dumps(j.get(\"request_skewed\"),indent=1)[:3500]); print(json.dumps(e.get(\"request_prob\"),indent=1)[:600]); print(json.dumps(e.get(\"other_sketchers_k21_s18000\",{}).get(\"prob\"),indent=1)[:1500])"
tail -5 gpurun_out/r06b_bench.err
