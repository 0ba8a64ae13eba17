#!/usr/bin/env python3
"""Aggregates the per-dispatch counter CSVs written by tools/pmc_passes.sh into one table:
per kernel name, the mean counter value per dispatch.   usage: pmc_summary.py <outdir> [name-filter]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "mrl"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(fh):
            k = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[k] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (d, c), v in per_dispatch.items():
            acc[names[d]][c].append(v)
res = {}
for k, cs in acc.items():
    if flt not in k:
        continue
    short = k.split("(")[0][-60:] + ("<" + k.split("<")[1].split(">")[0] + ">" if "<" in k else "")
    res[short] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    res[short]["_dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(res, indent=1))
