#!/usr/bin/env python3
"""eval+sample rate of a 90 x 90 x 180 customized_measurement table in each parameterisation (enum mrl_param), 64M random
units resident in HBM, and of a batch that mixes the three.   python tools/param_rates.py > profiles/r02_param_rates.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

n = 64 << 20
res = {"units": n, "dims": [90, 90, 180], "steps": 10}
with host.MerlHip(0) as g:
    wi, wo, u = g.generate_pairs(0x5EED, 0, n)
    out = tuple(torch.empty(s, dtype=torch.float32, device="cuda") for s in ((n, 3), (n,), (n, 3), (n,), (n, 3)))
    ids = []
    for name, param, kind in (("half_diff", 0, "ggx_tab"), ("standard", 1, "ggx_std"), ("standard_full", 2, "ggx_std_full")):
        g.set_option(host.OPT_TABLE_PARAM, param)
        mid = g.upload_table(synth.make_table(kind, 0, (90, 90, 180)), synth.MERL_SCALE)
        ids.append(mid)
        g.eval_sample(wi, wo, u, material=mid, out=out); torch.cuda.synchronize()
        g.timer_start()
        for _ in range(10):
            g.eval_sample(wi, wo, u, material=mid, out=out)
        ms = g.timer_stop() / 10
        res[name] = {"ms": round(ms, 4), "Munits_per_s": round(n / ms / 1e3, 1)}
    mat = (torch.arange(n, device="cuda", dtype=torch.int32) * 7 + 3) % 3
    g.eval_sample(wi, wo, u, mat=mat, out=out); torch.cuda.synchronize()
    g.timer_start()
    for _ in range(10):
        g.eval_sample(wi, wo, u, mat=mat, out=out)
    ms = g.timer_stop() / 10
    res["mixed_three_parameterisations"] = {"ms": round(ms, 4), "Munits_per_s": round(n / ms / 1e3, 1)}
print(json.dumps(res, indent=1))
