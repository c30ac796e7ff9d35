#!/usr/bin/env python3
"""Runs ON the GPU box: condenses rocprofv3 PC-sampling output (csv and/or json under <dir>) to a histogram of the samples that belong
to dispatches of kernels matching <regex>.
usage: pcsample_condense.py <dir> <kernel regex> <out.csv>
Output rows: key fields (whatever identifies the sampled instruction in this rocprofv3 version: code-object offset when the JSON
carries it, else the decoded instruction text + comment), optional stochastic fields (issued / type / stall reason), count."""
import collections, csv, glob, json, os, re, sys

d, rx, out = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3]
csv.field_size_limit(1 << 30)

# dispatch id -> kernel name from the kernel trace
disp = {}
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        disp[r.get("Dispatch_Id")] = r.get("Kernel_Name", "")
print("kernel dispatches:", len(disp), "matching:", sum(1 for v in disp.values() if rx.search(v)))

hist = collections.Counter()
nsamp = 0
for fn in glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True):
    rd = csv.DictReader(open(fn))
    print("csv", fn, rd.fieldnames)
    first = True
    for r in rd:
        if first:
            print("example row:", dict(r)); first = False
        k = disp.get(r.get("Dispatch_Id"), "")
        if disp and not rx.search(k):
            continue
        nsamp += 1
        key = tuple(r.get(c, "") for c in ("Instruction", "Instruction_Comment", "Wave_Issued_Instruction", "Instruction_Type", "Stall_Reason", "Wave_Count") if c in r)
        hist[key] += 1
    cols = [c for c in ("Instruction", "Instruction_Comment", "Wave_Issued_Instruction", "Instruction_Type", "Stall_Reason", "Wave_Count") if c in (rd.fieldnames or [])]

def walk(o, path=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from walk(v, path + "/" + k)
    elif isinstance(o, list):
        if o and isinstance(o[0], dict) and any("pc" in x for x in o[0].keys()):
            yield path, o
        else:
            for v in o[:4]:
                yield from walk(v, path + "[]")

jhist = collections.Counter()
for fn in glob.glob(os.path.join(d, "**", "*.json"), recursive=True):
    try:
        if os.path.getsize(fn) > (6 << 30):
            print("json too large", fn); continue
        J = json.load(open(fn))
    except Exception as e:
        print("json load failed", fn, e); continue
    for path, lst in walk(J):
        print("json list", path, len(lst), "example:", json.dumps(lst[0])[:800])
        for s in lst:
            pc = s.get("pc") or s.get("record", {}).get("pc") or {}
            did = str(s.get("dispatch_id", s.get("record", {}).get("dispatch_id", "")))
            k = disp.get(did, "")
            if disp and did and not rx.search(k):
                continue
            rec = s.get("record", s)
            snap = rec.get("snapshot", {}) if isinstance(rec, dict) else {}
            key = (pc.get("code_object_id", ""), pc.get("code_object_offset", pc.get("offset", "")), rec.get("wave_issued", ""), rec.get("inst_type", ""),
                   snap.get("reason_not_issued", rec.get("reason_not_issued", "")), s.get("inst_index", ""))
            jhist[key] += 1

with open(out, "w") as f:
    w = csv.writer(f)
    if hist:
        w.writerow(["#csv"] + cols + ["count"])
        for k, c in hist.most_common():
            w.writerow(["c"] + list(k) + [c])
    if jhist:
        w.writerow(["#json", "code_object_id", "offset", "wave_issued", "inst_type", "reason_not_issued", "inst_index", "count"])
        for k, c in jhist.most_common():
            w.writerow(["j"] + list(k) + [c])
print("samples of matching kernels (csv):", nsamp, "distinct:", len(hist), "| json distinct:", len(jhist), "total:", sum(jhist.values()))
