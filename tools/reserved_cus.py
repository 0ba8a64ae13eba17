#!/usr/bin/env python3
"""What MRL_OPT_RESERVED_CUS costs on one device: the headline launch (fabric-bound) and the GGX launch (VALU-bound) with k compute
units left to communication kernels (a CU-masked stream + a grid sized for the rest).  A VALU-bound launch should slow down by
256 / (256 - k) if the mask excludes exactly k CUs; results must not change.   python tools/reserved_cus.py > profiles/r04_reserved_cus.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

n = 64 << 20
res = {"units": n, "library": host.build_info(), "rows": []}
with host.MerlHip(0) as g:
    tab = g.upload_merl(synth.make_table("ggx_tab", 0))
    ggx = g.ggx(0.1, (0.143, 0.375, 1.442), (3.983, 2.386, 1.603))
    wi, wo, u = g.generate_pairs(0x5EED, 0, n)
    # the 100-table set of BASELINE configs[4] (translation-bound): material ids per unit
    g.set_option(host.OPT_TABLE_ARENA_MB, 20480)
    many = [g.upload_merl(synth.make_table("ggx_tab", s % 16)) for s in range(100)]
    mat = g.generate_materials(0x5EED, 0, n, 100) + many[0]
    ref = {}
    for k in (0, 4, 8, 16, 0):
        # (the context's own stream carries the mask: no use_torch_stream here)
        g.set_option(host.OPT_RESERVED_CUS, k)
        row = {"reserved_cus": k, "compute_units_for_grids": g.compute_units - k}
        for name, mid in (("merl64m", tab), ("ggx64m", ggx), ("resident100_64m", None)):
            call = (lambda: g.eval_sample(wi, wo, u, mat=mat)) if mid is None else (lambda: g.eval_sample(wi, wo, u, material=mid))
            for _ in range(3):
                out = call()
            g.synchronize()
            g.timer_start()
            for _ in range(10):
                out = call()
            ms = g.timer_stop() / 10
            g.synchronize()
            same = True
            if name in ref:
                same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(out, ref[name]))
            else:
                ref[name] = [t.clone() for t in out]
            row[name] = {"ms": round(ms, 4), "G_units_per_s": round(n / ms / 1e6, 2), "same_bits_as_unreserved": same}
        res["rows"].append(row)
    base = res["rows"][0]
    for row in res["rows"]:
        k = row["reserved_cus"]
        row["ggx_slowdown"] = round(row["ggx64m"]["ms"] / base["ggx64m"]["ms"], 4)
        row["expected_if_exactly_k_cus_are_masked"] = round(256.0 / (256 - k), 4)
print(json.dumps(res, indent=1))
