#!/usr/bin/env python3
"""tools/pmc_rgl.sh's counter CSVs -> one table: per configuration (file shape x search mode) and entry point the mean counter
values per launch and what follows from them per unit (one lane = one unit; 16M units = 262,144 wave-units per launch).
    python3 tools/pmc_rgl_summary.py <outdir> > profiles/r04_rgl_pmc.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
UNITS = 16 << 20
MODES = {"0": "eval", "1": "pdf", "2": "sample", "3": "eval_sample", "4": "eval_pdf"}
res = {"what": "rocprofv3 --kernel-trace --pmc passes over tools/rgl_pmc_driver.py (tools/pmc_rgl.sh): mean per launch, 16M random units per launch; "
               "GRBM_GUI_ACTIVE is summed over 8 XCDs, TA / TCP counters over 256 CUs. Round 3's kernel for comparison: profiles/r03_rgl_pmc.json "
               "(vmem loads per unit 33.5 / 25.5 / 126 / 159.5 for eval / pdf / sample / eval_sample, mean of the two files). "
               "anisotropic_lds: the default for a file whose conditional integrals do not fit a CU's LDS — marginal rows in LDS for sample(), and "
               "the fused call runs as the eval_pdf kernel followed by the sample kernel (no eval_sample row)."}
for cfg in sorted(os.listdir(out)):
    d = os.path.join(out, cfg)
    if not os.path.isdir(d) or cfg.startswith("stats_"):
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float); names = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for (disp, c), v in per.items():
            m = re.search(r"k_rgl(_lds)?<(\d)", names[disp])
            if m and m.group(2) in MODES:
                acc[MODES[m.group(2)] + ("" if not m.group(1) else "")][c].append(v)
                acc[MODES[m.group(2)]]["_lds_kernel"] = [1.0 if m.group(1) else 0.0]
    table = {}
    for mode, cs in acc.items():
        c = {k: sum(v) / len(v) for k, v in cs.items()}
        wave_units = UNITS / 64
        row = {"kernel": "k_rgl_lds" if c.pop("_lds_kernel", 0) else "k_rgl", "counters": {k: round(v) for k, v in sorted(c.items())}}
        g = lambda k: c.get(k, float("nan"))
        row["per_unit"] = {
            "vmem_load_instructions": round(g("SQ_INSTS_VMEM_RD") / wave_units, 1),
            "lds_instructions": round(g("SQ_INSTS_LDS") / wave_units, 1),
            "valu_instructions": round(g("SQ_INSTS_VALU") / wave_units, 0),
            "salu_instructions": round(g("SQ_INSTS_SALU") / wave_units, 0),
            "l1_lines_per_wave_load": round(g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("TA_FLAT_READ_WAVEFRONTS_sum"), 1),
            "l1_hit_rate": round(1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"), 3),
        }
        # SQ_ACTIVE_INST_* count quad-cycles (4 clocks) summed over the chip's 1,024 SIMDs; GRBM_GUI_ACTIVE is the launch's clocks summed over 8 XCDs
        simd_cycles = g("GRBM_GUI_ACTIVE") / 8 * 1024
        row["fractions"] = {
            "ta_busy": round(g("TA_TA_BUSY_sum") / (256 * g("GRBM_GUI_ACTIVE") / 8), 3),
            "valu_busy": round(g("SQ_ACTIVE_INST_VALU") * 4 / simd_cycles, 3),
            "lds_busy": round(g("SQ_ACTIVE_INST_LDS") * 4 / (simd_cycles / 4), 3),       # one LDS pipe per CU
            "wait_of_wave_cycles": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3) if "SQ_WAVE_CYCLES" in c else None,
            "lds_bank_conflict_cycles_per_lds_instruction": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_INSTS_LDS"), 1) if g("SQ_INSTS_LDS") > 0 else None,
        }
        table[mode] = row
    res[cfg] = table
print(json.dumps(res, indent=1))
