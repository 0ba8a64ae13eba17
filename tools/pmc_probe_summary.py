#!/usr/bin/env python3
"""Mean counter value per dispatch for every pass directory under <outdir> (req_*, fetch_*, write_*).
usage: pmc_probe_summary.py <outdir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*_*"))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: defaultdict(float))
    names = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    per_kernel = defaultdict(lambda: defaultdict(list))
    for disp, cs in acc.items():
        for c, v in cs.items():
            per_kernel[names[disp]][c].append(v)
    res[os.path.basename(d)] = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"_dispatches": max(len(v) for v in cs.values())}
                                for k, cs in per_kernel.items()}
print(json.dumps(res, indent=1))
