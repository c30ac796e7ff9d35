#!/usr/bin/env python3
"""Runs ON the GPU box after the two rocprofv3 --pmc passes of tools/pmc_bench.sh: condenses the counter_collection CSVs into the sidecar
bench.py reads through GS_PMC_SIDECAR and keeps the raw rows of the kernels of interest (so that `traffic` can be recomputed).
usage: pmc_condense.py <fetch_dir> <write_dir> <out_json> <out_raw_csv>"""
import collections, csv, glob, json, os, re, sys

fetch_dir, write_dir, out_json, out_raw = sys.argv[1:5]
extra_dirs = sys.argv[5:]          # further counter directories (e.g. the passes over `bench.py --workload c5dist`)
KERNELS = {"k_hnsw_search_dense": r"k_hnsw_search_dense", "k_hnsw_search_u64": r"k_hnsw_search<2>", "k_hnsw_search": r"k_hnsw_search<", "k_match_join": r"k_match_join<",
           "k_sketch_min": r"k_sketch_min<", "k_hamming_qxc": r"k_hamming_qxc<", "k_sketch_hll": r"k_sketch_hll<"}
rows = []
vals = collections.defaultdict(list)        # (kernel, counter) -> [(grid, value)]
for d in [fetch_dir, write_dir] + extra_dirs:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            for name, rx in KERNELS.items():
                if re.search(rx, r["Kernel_Name"]):
                    vals[(name, r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
                    rows.append([name, r["Kernel_Name"].split("(")[0][:80], r["Grid_Size"], r["Counter_Name"], r["Counter_Value"], r.get("Dispatch_Id", "")])
                    break
out = {"_head": os.environ.get("GS_HEAD", ""), "_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/pmc_bench.sh) over `python bench.py --steps 1 --warmup 1`; counter "
               "values are KB; per kernel the launches of the LARGEST grid (the timed request step; smaller grids are build-time launches) are averaged; FETCH_SIZE is doubled as "
               "MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950; bench.py applies the per-pattern calibration of profiles/r03_fetchcal.txt to the RAW values kept here "
               "(coalesced streams x2, scattered 2-byte look-ups 64 B each as counted, atomics 32 B of WRITE_SIZE each)",
       "kernels": {}}
for name in KERNELS:
    f, w = vals.get((name, "FETCH_SIZE")), vals.get((name, "WRITE_SIZE"))
    if not f:
        continue
    g = max(x for x, _ in f)
    fv = [y for x, y in f if x == g]
    wv = [y for x, y in (w or []) if x == g]
    fetch = sum(fv) / len(fv) * 1024.0
    write = (sum(wv) / len(wv) * 1024.0) if wv else 0.0
    out["kernels"][name] = {"grid_size": g, "launches_sampled": len(fv), "FETCH_SIZE_bytes_raw": fetch, "fetch_bytes_corrected_x2": 2 * fetch, "WRITE_SIZE_bytes": write,
                            "hbm_bytes_per_launch": 2 * fetch + write}
json.dump(out, open(out_json, "w"), indent=1)
with open(out_raw, "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["kernel", "kernel_name", "grid_size", "counter", "value_KB", "dispatch_id"])
    wr.writerows(rows)
print(json.dumps(out["kernels"], indent=1))
