#!/usr/bin/env python3
"""Copies what tools/profile_configs.sh wrote (bench line, kernel stats, PMC summary per config) into profiles/rNN_* and
rebuilds profiles/traffic.json from the counters.   python tools/collect_profiles.py r02 <dir> [<dir> ...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = {"merl64m": "void mrl::<3, false, true, false, false, false>", "ggx64m": "void mrl::<3, true, false, false>",
          "mixed16_256m": "void mrl::<3, true, true, false, false, false>", "resident100": "void mrl::<3, true, true, false, false, false>"}
tag, dirs = sys.argv[1], sys.argv[2:]
rows = []
for d in dirs:
    for cfg, kern in KERNEL.items():
        b = os.path.join(d, cfg + "_bench.json")
        if not os.path.exists(b):
            continue
        shutil.copy(b, os.path.join(ROOT, "profiles", f"{tag}_{cfg}_bench.json"))
        shutil.copy(os.path.join(d, cfg + "_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_{cfg}_kernel_stats.csv"))
        pm = json.load(open(os.path.join(d, cfg + "_pmc_summary.json")))
        json.dump(pm, open(os.path.join(ROOT, "profiles", f"{tag}_{cfg}_pmc_summary.json"), "w"), indent=1)
        bench = json.loads(open(b).read().strip().splitlines()[-1])
        k = pm[kern]
        rd = k["TCC_EA0_RDREQ_128B_sum"] * 128 + (k["TCC_EA0_RDREQ_sum"] - k["TCC_EA0_RDREQ_128B_sum"]) * 64
        wr = k["TCC_EA0_WRREQ_64B_sum"] * 64 + (k["TCC_EA0_WRREQ_sum"] - k["TCC_EA0_WRREQ_64B_sum"]) * 32
        rows.append({
            "config": cfg, "kernel_variant": 3, "table_layout": 1, "units": bench["config"]["units_per_gpu_per_step"],
            "library": bench["config"].get("library", "unrecorded"),        # mrl_build_info() of the library the counters were taken on
            "hbm_bytes_per_launch": int(k["FETCH_SIZE"] * 1024 * 2 + k["WRITE_SIZE"] * 1024),
            "fetch_bytes": int(k["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(k["WRITE_SIZE"] * 1024),
            "tcc_ea0_request_bytes": int(rd + wr), "tcc_ea0_rdreq": int(k["TCC_EA0_RDREQ_sum"]),
            "tcc_ea0_rdreq_128B": int(k["TCC_EA0_RDREQ_128B_sum"]), "tcc_ea0_wrreq": int(k["TCC_EA0_WRREQ_sum"]),
            "tcc_ea0_wrreq_64B": int(k["TCC_EA0_WRREQ_64B_sum"]),
            "l2_hit_rate": round(k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]), 4),
            "source": f"profiles/{tag}_{cfg}_pmc_summary.json: FETCH_SIZE x 1024 x 2 (gfx950 counts 128-B read requests at 64 B: "
                      "MI355X_MICROARCH.md HBM section; equals TCC_EA0_RDREQ_128B x 128 B, and a float4 copy of known size under the "
                      "same counters reads back exactly — profiles/r02_partial_line_probe.json) + WRITE_SIZE x 1024 "
                      "(= TCC_EA0_WRREQ_64B x 64 B), separate --pmc passes"})
        print(cfg, rows[-1]["hbm_bytes_per_launch"], rows[-1]["l2_hit_rate"], bench["value"], bench["roofline"]["kernel_ms"])
path = os.path.join(ROOT, "profiles", "traffic.json")
old = json.load(open(path))["rows"]
done = {r["config"] for r in rows}
keep = [r for r in old if not (r.get("config", "merl64m") in done and r["kernel_variant"] == 3 and r["table_layout"] == 1)]
json.dump({"rows": rows + keep}, open(path, "w"), indent=1)
