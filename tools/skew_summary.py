#!/usr/bin/env python3
"""one line per regime from a bench.py log: headline / request_redundant / request_skewed step times and their oracle checks.  usage: skew_summary.py [bench log]"""
import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_o.log"):
    if l.startswith('{"metric'):
        d = json.loads(l); r = d.get("request_redundant", {}); s = d.get("request_skewed", {})
        print("headline %.1f ms | redundant %.1f ms (oracle %s) | skewed %.1f ms (oracle %s, auto/better %.3f) %s" % (
            d["ms_per_step"], r.get("ms_per_step", 0), r.get("ids_distances_evals_equal_oracle_32_queries"), s.get("ms_per_step", 0),
            s.get("ids_distances_evals_equal_oracle_16_queries"), s.get("auto_over_better_forced_strategy", 0), json.dumps(s.get("auto"))))
