#!/usr/bin/env python3
"""Runs ON the GPU box: condenses rocprofv3 --pmc counter_collection CSVs (one directory per pass) for kernels matching a regex.
usage: pmc_kernel.py <regex> <dir> [<dir> ...]   -> prints per kernel / counter: launches of the largest grid, mean value"""
import collections, csv, glob, os, re, sys
rx = re.compile(sys.argv[1])
for d in sys.argv[2:]:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        res = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if rx.search(r["Kernel_Name"]):
                res[(r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
        for (k, cn), v in sorted(res.items()):
            g = max(x for x, _ in v); vals = [y for x, y in v if x == g]
            print("%-60s %-28s grid=%d launches=%d mean=%.4g" % (k, cn, g, len(vals), sum(vals) / len(vals)))
