#!/bin/bash
# Runs ON the GPU box: SQ instruction counters of k_sketch_min over `bench.py --workload sketch` (10 k x 5 Mbp, one launch per step), condensed into
# gpurun_out/<tag>_sketch_pmc.json (usage: tools/pmc_sketch.sh [tag, default r05]; copied to profiles/ by hand). The filtered emitter's work per k-mer is data dependent, so the bench prices the kernel
# with the MEASURED VALU instruction count rather than a static count.
T=${1:-r05}; export GS_PMC_TAG=$T
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex k_sketch_min -d $R/gpurun_out/pmc_sk -- python $R/bench.py --workload sketch --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_sk.log 2>&1
cd $R
python - <<'P'
import csv, glob, json, collections
res = collections.defaultdict(list)
for fn in glob.glob("gpurun_out/pmc_sk/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "k_sketch_min" in r["Kernel_Name"]:
            res[r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
out = {"_how": "rocprofv3 --kernel-trace --pmc <SQ counters> over `python bench.py --workload sketch --steps 2 --warmup 1` (tools/pmc_sketch.sh); mean over the launches of the largest grid (10 000 genomes x 5 Mbp, k=21, s=18000 optdens, filtered emitter)", "kmers_per_launch": 10000 * (5_000_000 - 20)}
for k, v in res.items():
    g = max(x for x, _ in v); vals = [y for x, y in v if x == g]
    out[k] = sum(vals) / len(vals); out["launches"] = len(vals); out["grid_size"] = g
out["valu_wave_instr_per_64_kmers"] = out["SQ_INSTS_VALU"] / out["kmers_per_launch"] * 64
import os
out["head"] = open(".head").read().strip() if os.path.exists(".head") else None
json.dump(out, open("gpurun_out/%s_sketch_pmc.json" % os.environ.get("GS_PMC_TAG", "r05"), "w"), indent=1)
print(json.dumps(out, indent=1))
P
rm -rf gpurun_out/pmc_sk
