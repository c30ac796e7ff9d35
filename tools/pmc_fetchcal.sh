#!/bin/bash
# Runs ON the GPU box: FETCH_SIZE / WRITE_SIZE (separate passes) of tools/ubench_fetchcal, condensed per kernel next to the known byte /
# request counts -> gpurun_out/r03_fetchcal.txt (bytes the counter reports per byte / per request really moved, per access pattern).
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B" "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_EA0_ATOMIC"; do
  D=$R/gpurun_out/fc_$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $D -- $R/tools/ubench_fetchcal > $R/gpurun_out/fetchcal_known.txt 2>$D.err
done
cd $R
python - <<'P' > gpurun_out/r03_fetchcal.txt
import csv, glob, collections, re
known = {}
for l in open("gpurun_out/fetchcal_known.txt"):
    m = re.match(r"known: (\w+) (\w+) (\d+)", l)
    if m: known[m.group(1)] = (m.group(2), int(m.group(3)))
vals = collections.defaultdict(dict)
for fn in glob.glob("gpurun_out/fc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        vals[k][r["Counter_Name"]] = float(r["Counter_Value"])
print("calibration of the memory-side counters on gfx950 (tools/ubench_fetchcal, 8 GiB buffer; FETCH_SIZE / WRITE_SIZE are in KB)")
for k, (what, n) in known.items():
    v = vals.get(k, {})
    f, w = v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024
    print("%-11s known %-11s %14d | FETCH_SIZE %14.0f B = %.3f per known unit | WRITE_SIZE %14.0f B = %.3f per known unit | %s" %
          (k, what, n, f, f / n, w, w / n, " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()) if c.startswith("TCC"))))
P
rm -rf gpurun_out/fc_*; cat gpurun_out/r03_fetchcal.txt
