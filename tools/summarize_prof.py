#!/usr/bin/env python3
"""Runs ON the GPU box after rocprofv3: condenses the raw CSVs under gpurun_out/ into small summaries and deletes the raw files
(gpurun merges at most 64 MiB back).  usage: summarize_prof.py <stats_dir> <fetch_dir> <write_dir> <out_prefix>"""
import collections, csv, glob, json, os, shutil, sys

def short(name):
    n = name.replace("void ", "")
    if "rocprim" in n or "hipcub" in n:
        return "rocprim/hipcub segmented sort + RLE kernels"
    return n.split("(")[0].split("<")[0].replace("gs::", "")

stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
rows = list(csv.DictReader(open(glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0])))
agg = collections.OrderedDict()
for r in rows:
    k = short(r["Name"]); a = agg.setdefault(k, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
with open(out + "_kernel_stats.csv", "w") as f:
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write('"%s",%d,%.0f,%.1f,%.3f\n' % (k, c, t, t / c, 100.0 * t / tot))
def pmc(d, cn):
    res = collections.defaultdict(list)
    fn = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fn: return res
    for r in csv.DictReader(open(fn[0])):
        if r.get("Counter_Name") == cn: res[short(r["Kernel_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return res
F, W = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
summ = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) on `python bench.py --steps 1 --warmup 1` (request, 300k-genome DB, "
                "2500 queries/step). Counters are KB; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950 (2-byte scattered lookups and atomics "
                "are uncalibrated); per launch of the largest-grid launches of each kernel (= the timed search step for the search-side kernels).", "kernels": {}}
for k in F:
    gmax = max(g for g, _ in F[k])
    fl = [v for g, v in F[k] if g == gmax]; wl = [v for g, v in W.get(k, []) if g == gmax]
    fetch = sum(fl) / len(fl) * 1024; write = (sum(wl) / len(wl) * 1024) if wl else 0.0
    summ["kernels"][k] = {"grid_size": gmax, "launches_sampled": len(fl), "FETCH_SIZE_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch, "WRITE_SIZE_bytes": write,
                          "hbm_bytes_per_launch": 2 * fetch + write}
json.dump(summ, open(out + "_pmc_traffic.json", "w"), indent=1)
for d in (stats_dir, fetch_dir, write_dir):
    shutil.rmtree(d, ignore_errors=True)
print(open(out + "_kernel_stats.csv").read()[:2500])
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in summ["kernels"].items()}, indent=0)[:1500])
