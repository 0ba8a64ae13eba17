#!/usr/bin/env python3
"""Throughput of the n-channel entry points beside the RGB path: fused eval+sample over 32M units,
MERL-sized tables, 1 / 2 / 3 (RGB path) / 4 / 8 / 16 / 32 channels.   python tools/nch_rates.py [widths...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mitsuba_customization_amd import host, synth

n = 32 << 20
rows = []
base = synth.make_table("ggx_tab", 0)
widths = [int(x) for x in sys.argv[1:]] or [1, 2, 3, 4, 8, 16, 32]
for C in widths:
    with host.MerlHip(0) as gpu:
        planes = np.stack([np.abs(base[c % 3]) * (1.0 + 0.01 * c) for c in range(C)])
        mid = gpu.upload_table_nch(planes, [1.0 / 1500.0] * C)
        table_mb = gpu.memory_info()["table_bytes"] / 1e6
        wi, wo, u = gpu.generate_pairs(0x5EED, 0, n)
        fn = lambda: gpu.eval_sample_nch(wi, wo, u, C, material=mid)
        out = fn(); torch.cuda.synchronize()
        del out
        reps = 5
        gpu.timer_start()
        for _ in range(reps):
            out = fn()
        ms = gpu.timer_stop() / reps
        lines = {1: 1, 2: 1}.get(C, (C + 3) // 4)                  # 128-B lines a lookup touches
        rows.append({"channels": C, "ms_per_32M_units": round(ms, 3), "G_units_per_s": round(n / ms / 1e6, 2),
                     "G_channel_values_per_s": round(2 * C * n / ms / 1e6, 1), "table_MB": round(table_mb, 1),
                     "lines_per_lookup": lines, "stream_bytes_per_unit": 32 + 4 + 12 + 4 + 8 * C})
        del out, wi, wo, u
        torch.cuda.empty_cache()
print(json.dumps({"what": "mrl_eval_sample_batch_nch, 32M units, MERL-sized synthetic tables", "rows": rows}, indent=1))
