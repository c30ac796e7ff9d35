#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel trace of `bench.py --steps 1 --warmup 1`, condensed to the kernels of the LAST request step (everything after the
# last k_sketch_min launch starts): name, launches, total ms - what a 10 000-query request consists of, launch by launch.
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/steptrace -o st -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $R/gpurun_out/steptrace.log 2>&1
cd $R
python - <<'P'
import csv, glob, collections
fn = glob.glob("gpurun_out/steptrace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
def grid(r):
    if "Grid_Size" in r: return int(r["Grid_Size"])
    return int(r.get("Grid_Size_X", 1)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed step = from the last-but-(parity runs) ... take the LAST launch of k_sketch_min with the largest grid and go until the next k_hnsw_search_dense ends
big = [i for i, r in enumerate(rows) if "k_sketch_min" in r["Kernel_Name"] and grid(r) >= 5000000]
i0 = big[-1]
# include the prefix kernels of the sketch call (unit prefix etc.) launched just before
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 2_000_000 and "k_hnsw_search_dense" not in rows[i0 - 1]["Kernel_Name"]: i0 -= 1
i1 = next(i for i in range(big[-1], len(rows)) if "k_hnsw_search_dense" in rows[i]["Kernel_Name"])
seg = rows[i0:i1 + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
agg = collections.OrderedDict()
for r in seg:
    k = r["Kernel_Name"].split("(")[0][:70]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
busy = sum(v[1] for v in agg.values())
with open("gpurun_out/r06_request_step_trace.txt", "w") as f:
    f.write("one 10 000-query request (bench.py step), rocprofv3 --kernel-trace: %d launches, %.2f ms from first start to last end, %.2f ms inside kernels\n" % (len(seg), (t1 - t0) / 1e6, busy))
    for k, (n, ms) in sorted(agg.items(), key=lambda x: -x[1][1]):
        f.write("%8.3f ms  %4d x  %s\n" % (ms, n, k))
print(open("gpurun_out/r06_request_step_trace.txt").read())
P
rm -rf gpurun_out/steptrace
