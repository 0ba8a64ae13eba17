#!/usr/bin/env python3
"""What tools/collect_round.sh wrote -> profiles/rNN_*: the four configs (tools/collect_profiles.py), the VALU counters
(tools/collect_valu.py), the RGL counters / kernel stats / rates / L2 view, the two parity soaks, the gather microbenchmark.
    python tools/collect_round.py r04 gpurun_out/<dir>"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], sys.argv[2]
P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_profiles.py"), tag, os.path.join(d, "configs")])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_valu.py"), tag, os.path.join(d, "valu")], stdout=subprocess.DEVNULL)
with open(P("rgl_pmc.json"), "w") as f:
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_rgl_summary.py"), os.path.join(d, "rgl")], stdout=f)
for shape in ("isotropic", "anisotropic"):
    shutil.copy(os.path.join(d, "rgl", f"rgl_{shape}_lds_kernel_stats.csv"), P(f"rgl_{shape}_kernel_stats.csv"))
for name in ("rgl_rates.json", "fuzz_parity_rgl.json", "fuzz_parity.json"):
    shutil.copy(os.path.join(d, name), P(name))
rows = [json.loads(l) for l in open(os.path.join(d, "gather_quad.jsonl")) if l.startswith("{")]
json.dump({"what": "tools/microbench/gather_quad.hip: a lane's four 16-B reads by how their addresses lie (scatter: four lines; lane64: one 64-B block; "
                   "quad / quadraw: the block read by the quad, with / without the DPP way back; lane32 / pair32: two reads of a 32-B block), 16 waves per CU, "
                   "by footprint. cu_cycles_per_wave_load = CU-cycles per wave-instruction. Reads of one lane that fall in one line cost one line fill, "
                   "whichever instruction issues them; a line fill costs 0.7 (1 MB: L2) to 2.4 (36 MB) CU-cycles.", "rows": rows}, open(P("gather_quad.json"), "w"), indent=1)
res = {"what": "tools/pmc_rgl_l2.sh over tools/rgl_pmc_driver.py (integrals from memory), mean per launch / per wave-unit (16M units = 262,144 waves): L2 hit rate, "
               "fabric read requests, and the L1's view of its misses (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = cycles per miss; TCP_PENDING_STALL_CYCLES "
               "summed over 256 CUs)"}
for shape in ("isotropic", "anisotropic"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "rgl_l2", shape, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float); names = {}
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, c), v in per.items():
            m = re.search(r"k_rgl(_lds)?<(\d)", names[disp])
            if m:
                acc[{"0": "eval", "1": "pdf", "2": "sample", "3": "eval_sample", "4": "eval_pdf"}[m.group(2)]][c].append(v)
    t = {}
    W = 262144
    for mode, cs in acc.items():
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        t[mode] = {"l2_requests_per_wave": round(c["TCC_REQ_sum"] / W), "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3),
                   "fabric_read_requests_per_wave": round(c["TCC_EA0_RDREQ_sum"] / W, 1), "l1_miss_requests_per_wave": round(c["TCP_TCC_READ_REQ_sum"] / W),
                   "cycles_per_l1_miss": round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]),
                   "l1_pending_stall_cycles_per_wave": round(c["TCP_PENDING_STALL_CYCLES_sum"] / W)}
    res[shape + "_memory"] = t
json.dump(res, open(P("rgl_l2.json"), "w"), indent=1)
print("collected", tag, "from", d)
