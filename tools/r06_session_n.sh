#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_join_blocks.py tests/test_gpu_fullsize.py -q -k "join or skewed or redundant or cost_model or config2" 2>&1 | tail -3 > gpurun_out/session_n.log
timeout 1500 python -u bench.py --steps 3 --warmup 1 > gpurun_out/bench_n.log 2> gpurun_out/bench_n.err
python tools/bench_summary.py final < gpurun_out/bench_n.log >> gpurun_out/session_n.log 2>&1
python - >> gpurun_out/session_n.log 2>&1 <<'PY'
import json
for l in open("gpurun_out/bench_n.log"):
    if l.startswith('{"metric'):
        d = json.loads(l)
        r = d.get("request_redundant", {}); s = d.get("request_skewed", {})
        print("redundant %.1f ms (%.3fx) oracle %s" % (r.get("ms_per_step", 0), r.get("ms_per_step_over_headline", 0), r.get("ids_distances_evals_equal_oracle_32_queries")))
        print("skewed %.1f ms (%.3fx) auto %s oracle %s auto/better %.3f" % (s.get("ms_per_step", 0), s.get("ms_per_step_over_headline", 0), json.dumps(s.get("auto")), s.get("ids_distances_evals_equal_oracle_16_queries"), s.get("auto_over_better_forced_strategy", 0)))
        p = d["extra_legs"].get("request_prob", {}); print("prob request %.1f ms" % p.get("ms_per_10000_queries", 0))
PY
cat gpurun_out/session_n.log
