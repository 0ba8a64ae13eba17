#!/usr/bin/env python3
"""Turns what tools/pmc_valu.sh collected (SQ counters of the bench launch on random and on coherent inputs) into per-unit
numbers: profiles/<tag>_valu_pmc.json (both input kinds, every counter) and profiles/valu.json (what bench.py's
roofline.valu reads: executed VALU instructions per unit, by class, with the library sources they were counted on).
    python tools/collect_valu.py r03 gpurun_out/r03/pmc_valu_final"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, d = sys.argv[1], sys.argv[2]
UNITS = 64 << 20
out = {}
for mode in ("random", "coherent"):
    acc = defaultdict(list)
    dur = []
    for f in glob.glob(os.path.join(d, mode + "_*", "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "k_table_dma" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, c), v in per.items():
            acc[c].append(v)
    for f in glob.glob(os.path.join(d, mode + "_*", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_table_dma" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    m = {c: sum(v) / len(v) for c, v in sorted(acc.items())}
    wave_iters = UNITS / 64
    row = {c.replace("SQ_", "").lower() + "_per_wave_iteration": round(v / wave_iters, 2) for c, v in m.items() if c != "SQ_WAVES"}
    row["waves"] = m.get("SQ_WAVES")
    row["kernel_ms_under_the_profiler_median"] = round(sorted(dur)[len(dur) // 2], 3) if dur else None
    # one lane = one unit: VALU instructions per wave-iteration == per unit
    row["valu_busy_share_of_wave_time_x2_waves_per_simd"] = round(2 * m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], 3)
    out[mode] = row
library = None
for log in glob.glob(os.path.join(d, "random_sq1.log")):
    for line in open(log):
        if line.startswith("{"):
            library = json.loads(line)["config"].get("library")
out["library"] = library
out["units_per_launch"] = UNITS
out["note"] = ("SQ counters count per wave, in units of 4 cycles for the *_CYCLES / ACTIVE_* ones; per wave-iteration = per 64 units, and with one lane "
               "per unit an instruction count per wave-iteration IS the count per unit.  Collected by tools/pmc_valu.sh (counters only with "
               "--kernel-trace, one group per run); the profiler pins lower clocks than a plain run, so cycle shares are not comparable with bench times.")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_valu_pmc.json"), "w"), indent=1)
r = out["random"]
valu = {"library": library, "source": f"profiles/{tag}_valu_pmc.json (tools/pmc_valu.sh, SQ_INSTS_VALU* / wave-iterations)",
        "kernel": "k_table_dma<eval_sample>", "valu_insts_per_unit": r["insts_valu_per_wave_iteration"],
        "f64_fma_mul_add_per_unit": round(r["insts_valu_fma_f64_per_wave_iteration"] + r["insts_valu_mul_f64_per_wave_iteration"] + r["insts_valu_add_f64_per_wave_iteration"], 2),
        "f64_transcendental_per_unit": r["insts_valu_trans_f64_per_wave_iteration"], "convert_per_unit": r["insts_valu_cvt_per_wave_iteration"],
        "f32_fma_mul_per_unit": round(r["insts_valu_fma_f32_per_wave_iteration"] + r["insts_valu_mul_f32_per_wave_iteration"], 2),
        "salu_per_unit": r["insts_salu_per_wave_iteration"], "lds_insts_per_unit": r["insts_lds_per_wave_iteration"]}
json.dump(valu, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1)
print(json.dumps(valu, indent=1))
