#!/usr/bin/env python3
"""VERDICT r3 item 6, one bounded attempt at the multi-table translation limit: would ordering the lookups of a workgroup by TABLE help?
The 100-table launch (BASELINE configs[4]'s per-GPU share: 125M units, 18.7 GB of bricks) is bound by address translation (UTCL1 hits
99.8 % with one table, 33-38 % with 100: profiles/r03_table_set_translation.json).  Before building a kernel that sorts a window of
units by material id, this measures what it could buy: the SAME fused mixed-material launch on material-id arrays that are already
sorted inside windows of W units (every 256-unit tile then meets one or two tables), from W = 256 (tile-local) to the whole batch —
the upper bound of any in-kernel ordering, at zero implementation cost.  Also: the rows layout (2.4 GB instead of 18.7 GB) for the
same set.      python tools/table_locality_probe.py > profiles/r04_table_locality_probe.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

n, n_tables = 125_000_000, 100
res = {"units": n, "tables": n_tables, "library": host.build_info(), "rows": []}
tables = [synth.make_table("ggx_tab", s) for s in range(16)]


def timed(g, call, reps=5):
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    g.timer_start()
    for _ in range(reps):
        call()
    ms = g.timer_stop() / reps
    torch.cuda.synchronize()
    return ms


for layout in (1, 0):
    with host.MerlHip(0) as g:
        g.use_torch_stream()
        g.set_option(host.OPT_TABLE_LAYOUT, layout)
        if layout == 1:
            g.set_option(host.OPT_TABLE_ARENA_MB, 20480)
        ids = [g.upload_merl(tables[t % 16]) for t in range(n_tables)]
        wi, wo, u = g.generate_pairs(0x5EED, 0, n)
        mat = g.generate_materials(0x5EED, 0, n, n_tables) + ids[0]
        out = None
        base = timed(g, lambda: g.eval_sample(wi, wo, u, mat=mat))
        row = {"layout": "bricks" if layout == 1 else "rows", "table_bytes": g.memory_info()["table_bytes"], "random_ids_ms": round(base, 3),
               "random_ids_G_units_per_s": round(n / base / 1e6, 2)}
        if layout == 1:
            for W in (256, 4096, 25600, 262144, n):
                if W >= n:
                    m2 = torch.sort(mat)[0]
                else:
                    k = (n // W) * W
                    m2 = mat.clone()
                    m2[:k] = torch.sort(mat[:k].view(-1, W), dim=1)[0].view(-1)
                ms = timed(g, lambda: g.eval_sample(wi, wo, u, mat=m2))
                row[f"ids_sorted_in_windows_of_{W if W < n else 'all'}_ms"] = round(ms, 3)
                del m2
        res["rows"].append(row)
        del wi, wo, u, mat
        torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
