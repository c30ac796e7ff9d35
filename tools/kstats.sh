#!/bin/bash
# Runs ON the GPU box: rocprofv3 --kernel-trace --stats of a command, condensed to "calls total_ms avg_us name" (template arguments kept, parameter lists dropped).
# usage: tools/kstats.sh <out.txt> <command ...>
R=$(pwd); OUT=$1; shift; export TMPDIR=/tmp; cd /tmp
D=$R/gpurun_out/kstats_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ks -- bash -c 'cd "$0" && exec "$@"' "$R" "$@" > $D.log 2>&1
cd $R
python - "$D" "$OUT" <<'P'
import csv, glob, sys, os
d, out = sys.argv[1], sys.argv[2]
fn = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
with open(out, "w") as f:
    if not fn:
        f.write("no kernel_stats.csv\n")
    else:
        rows = list(csv.DictReader(open(fn[0])))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        f.write("%8s %12s %12s %6s  %s\n" % ("calls", "total_ms", "avg_us", "%", "kernel"))
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
            n = r["Name"].replace("void ", "").replace("gs::", "")
            n = n.split("(")[0][:110]
            f.write("%8d %12.3f %12.1f %6.2f  %s\n" % (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot, n))
print(open(out).read())
P
tail -3 $D.log; rm -rf $D $D.log
