#!/usr/bin/env python3
"""A/B of the per-kind wave compaction (MRL_OPT_KERNEL 3 vs 4) on batches that mix table and analytic
materials; interleaved rounds in ONE process, median and min per variant.   python tools/mixed_kinds_ab.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mitsuba_customization_amd import host, synth

n = 64 << 20
res = {}
with host.MerlHip(0) as gpu:
    t_ids = [gpu.upload_merl(synth.make_table("ggx_tab", s)) for s in range(4)]
    g_ids = [gpu.ggx(0.1, (0.143, 0.375, 1.442), (3.983, 2.386, 1.603)), gpu.ggx(0.3, (0.2, 0.9, 1.1), (3.9, 2.4, 2.2))]
    wi, wo, u = gpu.generate_pairs(0x5EED, 0, n)
    out = (torch.empty((n, 3), dtype=torch.float32, device="cuda"), torch.empty((n,), dtype=torch.float32, device="cuda"),
           torch.empty((n, 3), dtype=torch.float32, device="cuda"), torch.empty((n,), dtype=torch.float32, device="cuda"),
           torch.empty((n, 3), dtype=torch.float32, device="cuda"))
    r = gpu.generate_materials(0x5EED, 0, n, 1 << 20)                 # uniform ints in [0, 2^20)
    for share in (0.0, 0.1, 0.5, 0.9):
        pick = (r % 4).int()
        ggx_lane = (r >> 2) < int(share * (1 << 18))
        mat = torch.where(ggx_lane, torch.tensor(g_ids, device="cuda", dtype=torch.int32)[(pick % 2).long()],
                          torch.tensor(t_ids, device="cuda", dtype=torch.int32)[pick.long()]).contiguous()
        times = {v: [] for v in (0, 3, 4)}
        for rnd in range(6):
            for v in (0, 3, 4):
                if v == 0 and rnd >= 2:
                    continue
                gpu.set_option(host.OPT_KERNEL, v)
                gpu.eval_sample(wi, wo, u, mat=mat, out=out)
                torch.cuda.synchronize()
                gpu.timer_start()
                gpu.eval_sample(wi, wo, u, mat=mat, out=out)
                times[v].append(gpu.timer_stop())
        res[f"ggx_share_{share}"] = {f"variant_{v}": {"median_ms": float(np.median(t)), "min_ms": float(np.min(t)),
                                                        "G_units_per_s": n / float(np.median(t)) / 1e6} for v, t in times.items()}
print(json.dumps(res, indent=1))
