#!/usr/bin/env python3
"""reads bench.py's output on stdin, prints a one-line summary (value, per-step wall times, per-kernel launch averages)"""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
lines = [l for l in sys.stdin if l.startswith('{"metric')]
if not lines:
    print(tag, "no JSON line"); sys.exit(1)
j = json.loads(lines[-1])
print(tag, round(j["value"]), j.get("step_ms"), "build %.1fs" % j.get("build_seconds", 0),
      [(k["kernel"], k["launches"], round(k["total_ms"] / max(k["launches"], 1), 1)) for k in j.get("kernels", [])],
      j.get("parity_checked", {}).get("dist_evaluation_counts_equal_oracle"))
