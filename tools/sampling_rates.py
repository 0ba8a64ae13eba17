#!/usr/bin/env python3
"""What the table samplers cost: sample() alone and the fused eval+sample unit over 64M units, for the three
MRL_OPT_SAMPLING modes (cosine hemisphere, row marginal, conditional rows P(theta_h | theta_i)), with the variance
of the weight's luminance beside each.   python tools/sampling_rates.py > profiles/r03_sampling_rates.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mitsuba_customization_amd import host, synth

n = 64 << 20
res = {}
with host.MerlHip(0) as g:
    g.use_torch_stream()
    mid = g.upload_merl(synth.make_table("ggx_tab", 0))
    wi, wo, u = g.generate_pairs(0x5EED, 0, n)
    lum = torch.tensor([0.2126, 0.7152, 0.0722], device="cuda")
    for name, mode in (("cosine", 0), ("row_marginal", 1), ("conditional_rows", 2)):
        g.set_option(host.OPT_SAMPLING, mode)
        row = {}
        for what, call in (("sample", lambda: g.sample(wi, u, material=mid)), ("eval_sample", lambda: g.eval_sample(wi, wo, u, material=mid))):
            for _ in range(2):
                out = call()
            torch.cuda.synchronize()
            g.timer_start()
            for _ in range(5):
                out = call()
            ms = g.timer_stop() / 5
            row[what] = {"ms": round(ms, 3), "G_units_per_s": round(n / ms / 1e6, 2)}
        w = g.sample(wi[: 4 << 20], u[: 4 << 20], material=mid)[2] @ lum
        row["weight_luminance_mean"] = round(float(w.mean()), 4)
        row["weight_luminance_variance"] = round(float(w.double().var()), 4)
        res[name] = row
print(json.dumps(res, indent=1))
